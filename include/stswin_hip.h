/* libstswin_hip -- C ABI of the MI355X-native STswinCL hot path.
 *
 * The reference (YuemingJin/STswinCL) is pure Python: its "operator interface" for this path is the
 * nn.Module surface; there is no FFI table to mirror.  This header is therefore the
 * boundary a maintainer binds with ctypes: plain device pointers, sizes and a
 * hipStream_t, no torch types.  Each entry point names the reference lines it replaces
 * (paths relative to the reference repo root).
 *
 * Conventions
 *   dtype      0 = bf16 storage (fp32 accumulate), 1 = fp32 storage (exact fp32 MFMA; parity path)
 *   pointers   device memory on the current HIP device; row pitches (ld*) in ELEMENTS
 *   row maps   int32, -1 = zero row (padding); NULL = identity
 *   stream     hipStream_t (pass torch.cuda.current_stream().cuda_stream); kernels are asynchronous
 *   return     0 on success, negative on error (-hipError_t, or -1xxx for argument errors)
 *   No entry point allocates, synchronises or keeps state: all workspaces are caller-owned.
 */
#ifndef STSWIN_HIP_H
#define STSWIN_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* gemm_nt epilogue flags */
#define STSWIN_GF_GELU 1       /* out = gelu_erf(v); C2 (if given) receives v            (swin_512.py:18-19) */
#define STSWIN_GF_RESID 2      /* v += R[r_rows[m]][n]                                   (swin_512.py:234-235) */
#define STSWIN_GF_MUL_DGELU 4  /* v *= gelu'(R[..])   (backward of swin_512.py:19) */
#define STSWIN_GF_OUT_F32 8    /* C is fp32 regardless of dtype */
#define STSWIN_GF_ACCUM 16     /* C += v (needs OUT_F32) */
#define STSWIN_GF_RELU 32
#define STSWIN_GF_MUL_R 8192     /* v *= R[r_rows[m]][n]  (R = GELU' values saved by STSWIN_GF_C2_DGELU) */
#define STSWIN_GF_C2_DGELU 16384 /* with GF_GELU: C2 receives gelu'(v) instead of v - the fc1 forward then hands the backward
                                    its multiplier and the backward epilogue is one multiply */
#define STSWIN_GF_CS_PARTIAL 32768 /* `colsum` is a caller-owned fp32 table [2*ceil(M/256)][N] instead of the fp32 [N] accumulator: row b
                                    * receives (plain stores, no atomics) the column sums of output rows 128b..128b+127;
                                    * stswin_cs_reduce then adds the rows into the bias gradient.  M/128 same-address atomics cost
                                    * 38 us on a 65536 x 512 output; the table + reduce cost ~5 us (N % 4 == 0). */
#define STSWIN_GF_CS_SQ 65536   /* with STSWIN_GF_CS_PARTIAL: `colsum` holds TWO planes [2][2*ceil(M/256)][N]; the second receives the column
                                  * sums of SQUARES of the same values: the BatchNorm statistics of a convolution output come out of
                                  * the convolution (stswin_cs_group_reduce + stswin_bn_finalize with x = NULL) */
#define STSWIN_GF_TAPSKIP (1 << 29) /* tiled (non-ring) kernels, S > 1 with a_rows: a workgroup first scans its rows of the map and skips every
                                    * segment (convolution tap) that is padding for ALL of them - pays when every row tile has such a tap (dilation >=
                                    * half the map height: ASPP's dilation 18 on 32 x 32: 70 -> 53 us); the scan costs ~4 us per launch.
                                    * Bitwise neutral for FINITE weights only: a skipped all-padding tap contributes 0 * w, which the unskipped
                                    * kernel evaluates as NaN when w is Inf / NaN - a diverged run can look healthy in the skipped taps */
#define STSWIN_GF_BIG 128       /* tuning: force the 256x256 4-stage-ring kernel (bf16) */
#define STSWIN_GF_MID 512       /* tuning: 256x128x32 tile, 3-stage ring, 2 workgroups per CU (bf16) */
#define STSWIN_GF_NOPIPE 1024   /* tuning: 256x256 ring kernel without software-pipelined LDS fragment reads */
#define STSWIN_GF_HALF 2048     /* tuning: force the 256x128 ping-pong ring kernel */
#define STSWIN_GF_M32PP (1u << 31) /* tuning (STSWIN_TUNING builds): the 8-wave ping-pong ring on 32x32x16 MFMA tiles; with STSWIN_GF_W4R: the 4-wave
                                  * register-staged variant (global_load + ds_write_b128 instead of LDS-DMA).  All measured slower: profiles/r04_gemm_w4_experiment.txt */
#define STSWIN_GF_W4R (1 << 30) /* tuning (STSWIN_TUNING builds only): 256x256 ring with 4 waves of 128x128, one per SIMD, register-pipelined
                                 * 32x32x16 main loop - the vendor kernel's structure; same results, measured slower (profiles/r04_gemm_w4_experiment.txt) */
#define STSWIN_GF_ROT 4096      /* tuning (STSWIN_TUNING builds): 256x256 ring with the rotated ping-pong loop, one barrier per stage */
#define STSWIN_GF_NOBIG 256     /* tuning: forbid it (default: chosen when >= 256 big tiles fill the chip) */
#define STSWIN_GF_WAVES4 64     /* tuning: 4 waves of 64x64 per 128x128 tile instead of the default 8 waves of 64x32 */

int stswin_abi_version(void);

/* ---- a1-a4: roll + window_partition + pair regroup, and its inverse ------------------------------------
 * swin_512.py:26-38 (window_partition), :57-71 (window_reverse, T-aware), :210-218, :224-231 (roll, regroup).
 * Tokens are rows of a (B, frames_total, H*W, C) tensor; the pair is frames f0..f0+T-1 of every clip.
 * win_rowmap writes map[r] = source token of gathered row r (row order (B*nW, T, ws*ws));
 * win_move copies rows: dir 0 gather (out[r] = in[map[r]]), dir 1 scatter (out[map[r]] = in[r]). Bit-exact. */
int stswin_win_rowmap(int* map, int B, int T, int H, int W, int ws, int shift, int f0, int frames_total, void* stream);
int stswin_win_move(int dtype, const void* in, void* out, int B, int T, int H, int W, int C, int ws, int shift,
                    int f0, int frames_total, int dir, void* stream);
/* PatchMerging 2x2 gather map, 4 segments in the order (0,0),(1,0),(0,1),(1,1)  (swin_512.py:267-271). */
int stswin_merge_rowmap(int* map /*[4][frames*H/2*W/2]*/, int frames, int H, int W, void* stream);
/* 3x3 convolution tap map (padding = dilation), 9 segments ky*3+kx   (ASPP.py:13-20, base18.py:73). */
int stswin_conv3x3_rowmap(int* map /*[9][frames*H*W]*/, int frames, int H, int W, int dilation, void* stream);
/* general k x k convolution tap maps (resnet.py:31-34: stride 1/2, dilation 1/2/4; torchvision stem layers):
 * inverse = 0: map[t][m_out] = input pixel read by output pixel m_out through tap t (or -1 = padding);
 * inverse = 1: map[t][n_in]  = the output pixel that reads input pixel n_in through tap t (or -1)  -> dgrad gather. */
int stswin_conv_rowmap(int* map, int frames, int Hin, int Win, int Hout, int Wout, int k, int stride, int pad, int dilation,
                       int inverse, void* stream);
/* stem im2col (torchvision conv1 7x7/2 pad 3, Cin = 3; resnet.py:102): NCHW fp32 images [F][3][H][W] ->
 * patches [F*Ho*Wo][ld] with column (ky*7+kx)*3 + c, zero padded to ld. */
int stswin_stem_im2col(int dtype, const float* img, void* patches, long ld, int frames, int H, int W, int Ho, int Wo,
                       void* stream);
/* stem input as a 2 x 2 space-to-depth image (torchvision conv1 7x7/2 pad 3 = a 4 x 4 / stride 1 convolution over 12-channel
 * superpixels; resnet.py:98-102): NCHW fp32 images [F][3][H][W] -> records [F][Ho + 3][Wo + 3][16] (Ho = (H-1)/2 + 1; 2 records of
 * zero padding before, 1 after, in both directions), record (sy, sx)[(dy*2 + dx)*3 + c] = img[c][2(sy-2) + dy][2(sx-2) + dx],
 * positions 12..15 zero.  Output pixel (oy, ox) contracts, per tap row s = 0..3, the 64 contiguous values of records
 * (oy + s, ox .. ox + 3) with W[co][c][2s + dy - 1][2t + dx - 1] (t = record in the segment; taps outside 0..6 are zero):
 * stswin_gemm_nt / stswin_gemm_tn with a row map [4][M], Kseg = bseg = 64 and lda = 16.  The buffer must extend 48 values past
 * the last record (the last segment reads on). */
int stswin_stem_s2d(int dtype, const float* img, void* out, int frames, int H, int W, void* stream);
/* Weight gradient of the stem convolution from dy (bf16 [F*Ho*Wo][64]) and the space-to-depth image `rec` of stswin_stem_s2d (bf16):
 * dw fp32 [64][4][4][16] = [cout][tap row][record][position], the order stswin_gemm_tn produces over the row map (accumulate != 0 adds).
 * Record rows pass through an LDS ring once; per-workgroup partials go to `scratch` (>= stswin_stem_wgrad_scratch floats) and are added
 * in a fixed order.  Wo = (W-1)/2 + 1 must be a multiple of 128, else -1722 and nothing is launched.  resnet.py:98-102 backward. */
/* The stem convolution over the space-to-depth image `rec` of stswin_stem_s2d (bf16): y bf16 [F*Ho*Wo][64] = conv(7, 2, 3) with
 * wmat bf16 [64][4][4][16] (= [cout][tap row][record][position], zeros where the 7 x 7 window has no tap); stats (or NULL): the
 * STSWIN_GF_CS_SQ table of y.  Record rows pass through an LDS ring once, the weights live in registers: replaces stswin_gemm_nt
 * over the row map for this M = F*Ho*Wo, N = 64, K = 256 shape.  Wo % 128 != 0: -1732, nothing launched.  resnet.py:98-102. */
int stswin_stem_conv(const void* rec, const void* wmat, void* y, float* stats, int frames, int H, int W, void* stream);
long stswin_stem_wgrad_scratch(int frames, int Ho, int Wo);
int stswin_stem_wgrad(const void* dy, const void* rec, float* dw, int accumulate, float* scratch, long scratch_floats, int frames, int H, int W,
                      void* stream);
/* nn.MaxPool2d(3, 2, 1) on tokens [F][H][W][C] -> [F][Ho][Wo][C]; arg (uint8 [F*Ho*Wo][C]) = winning tap (first max in
 * (ky,kx) scan order, like torch); backward gathers dy through arg (no atomics). */
/* (Cout, Cin, k, k) fp32 nn.Conv2d weight -> the bf16 / fp32 GEMM operand matrices of the token convolutions in one launch:
 * fwd [cop][S][cip] (tap-major K of stswin_gemm_nt) and, if not NULL, dgrad [cip][S][cop].  omap[cop] / imap[cip] = source
 * channel of each padded channel position or -1 (zero).  (ASPP.py:37-50, base18.py:60-77, resnet.py convolutions) */
int stswin_conv_pack(int dtype, const float* w, void* fwd, void* dgrad, const int* omap, const int* imap, int co, int ci, int S,
                     int cop, int cip, void* stream);
/* nn.Linear weight [n][k] fp32 -> fwd [n][k] and (if not NULL) tr [k][n] in the compute dtype, one launch (n, k multiples of 4):
 * the B operands of y = x W^T and of dx = dy W (swin_512.py:18-21,115,139,275), re-made after every optimizer step. */
int stswin_linear_pack(int dtype, const float* w, void* fwd, void* tr, int n, int k, void* stream);
/* The same two packings for up to 64 Linear / 32 convolution weights in ONE launch each (host arrays of device pointers and
 * sizes): the re-cast of every GEMM operand after an optimizer step (train_swin.py:170-173 steps ~125 M parameters) was ~90
 * launches of 6-15 us; see stswincl_amd.ops.repack. */
int stswin_linear_pack_multi(int dtype, int count, const float* const* w, void* const* fwd, void* const* tr, const int* n, const int* k,
                             void* stream);
int stswin_conv_pack_multi(int dtype, int count, const float* const* w, void* const* fwd, void* const* dgrad, const int* const* omap,
                           const int* const* imap, const int* ci, const int* S, const int* cop, const int* cip, void* stream);
int stswin_maxpool3x3s2(int dtype, const void* in, long ldi, void* out, long ldo, unsigned char* arg, int frames, int H,
                        int W, int Ho, int Wo, int C, int backward, void* stream);
/* 3x3 / stride 1 / pad 1 convolution of 64 -> 64 channels over bf16 NHWC tokens [frames*H*W][64] (torchvision resnet18.layer1 as
 * used by resnet.py:104-105, basic blocks resnet.py:31-51) from an LDS-resident halo with the weights in registers: replaces the
 * gather form of stswin_gemm_nt for these shapes.  wmat = the [64][9][64] matrices stswin_conv_pack makes; sign = +1 with the fwd
 * matrix: the convolution; sign = -1 with the dgrad matrix: its input gradient (x = dy).  resid (or NULL): [frames*H*W][64] added
 * before the store (a gradient another consumer of the input produced); stats (or NULL): the STSWIN_GF_CS_SQ table of the stored
 * values ([2][2*ceil(M/256)][64]: per-128-row-block column sums | sums of squares).  W in {16, 32, 64, 128}, H*W % 256 == 0; other
 * geometries return -1702 / -1701 and launch nothing (the caller keeps the gather GEMM). */
int stswin_conv3x3_c64(const void* x, const void* wmat, void* y, const void* resid, float* stats, int frames, int H, int W, int sign,
                       void* stream);
/* Weight gradient of the same convolution from dy and x (both bf16 [frames*H*W][64]): dw fp32 [64][9][64] (the order of
 * stswin_gemm_tn with bseg = 64: [cout][tap][cin]) or, tapminor != 0, [64][64][3][3] (nn.Conv2d.weight's own layout); accumulate != 0
 * adds to dw instead of overwriting it.  Image rows of x pass through an LDS ring once, per-workgroup partials go to `scratch`
 * (>= stswin_conv3x3_c64_wgrad_scratch(frames, H, W) floats) and are added in a fixed order: bitwise reproducible.  W in {32, 64,
 * 128}, H*W % 128 == 0; else -1712 / -1711 and nothing is launched (the caller keeps stswin_gemm_tn).  resnet.py:31-51 backward. */
long stswin_conv3x3_c64_wgrad_scratch(int frames, int H, int W);
int stswin_conv3x3_c64_wgrad(const void* dy, const void* x, float* dw, int tapminor, int accumulate, float* scratch, long scratch_floats,
                             int frames, int H, int W, void* stream);

/* ---- segmented gather GEMM: C[c_rows[m]][n] = epi( sum_s A[a_rows[s][m]][0:Kseg] . B[n][s*Kseg:(s+1)*Kseg] )
 * nn.Linear of swin_512.py:115 (qkv, with the window gather fused via a_rows and the q scaling via
 * scale/scale_cols, :118), :139 (proj, with window_reverse + un-roll + shortcut fused via c_rows/r_rows, :224-234),
 * :18-21 (Mlp fc1+GELU, fc2 + residual), :275 (PatchMerging reduction); 1x1 / 3x3 / dilated nn.Conv2d of
 * ASPP.py:37-50 and base18.py:60-77 as implicit GEMM over NHWC tokens.  Kseg must be a multiple of 64 (bf16) / 32 (f32). */
int stswin_gemm_nt(int dtype, const void* A, long lda, const int* a_rows, const void* B, long ldb, void* C, long ldc,
                   const int* c_rows, void* C2, long ldc2, const float* bias, const void* R, long ldr,
                   const int* r_rows, int M, int N, int Kseg, int S, float scale, int scale_cols, int flags,
                   float* colsum /* optional fp32 [N]: += column sums of the stored values (bias gradient) */, void* stream);
/* out[n] += sum_b partials[b][n], b < 2*ceil(M/256): second half of a STSWIN_GF_CS_PARTIAL stswin_gemm_nt (same M, N). */
int stswin_cs_reduce(const float* partials, int M, int N, float* out, void* stream);
/* weight gradients: C[i][j] += sum_m At[at_rows[m]][i] * Bt[bt_rows[m]][j]   (fp32 atomics; splits<=0: auto).
 * bseg > 0 (convolution wgrad in one launch): column j of the B operand is column j % bseg of row
 * bt_rows[(j / bseg) * Mk + m], i.e. tap t = j / bseg uses its own row map.
 * With bf16 operands and a workspace the split-K partials are kept as bf16 (each an fp32 sum of Mk / splits products, rounded
 * once; the combine pass adds them in fp32): half the slab traffic.  Environment STSWIN_TN_F32_SLABS=1 keeps fp32 partials.
 * Bits of a positive `splits`: STSWIN_TN_OVERWRITE (1<<27) stores C = result instead of accumulating (C may be
 * uninitialised; saves the caller's zero fill); bits 28-30 are tuning overrides (forbid / force the 256x256 ring kernel,
 * 4-wave 128x128 variant) used by tools/tn_sweep.py; the low bits are the split count, 0 = automatic. */
#define STSWIN_TN_OVERWRITE (1 << 27)
#define STSWIN_TN_NO_COMBINE (1 << 26)   /* with a workspace: leave the partials there; the caller runs stswin_tn_combine */
/* gemm_nt for few output tiles and a long K (ASPP.py:13-20,37-40: dilated 3x3 convolutions on 32x32 maps; M = 4096, N = 512,
 * K = 9 x 1024): the 256x256 ring kernel over tiles x K-splits with fp32 partial slabs in `workspace`, then a fixed-order combine
 * (+ bias, ReLU) into C (bf16, pitch ldc).  bf16 only, plain epilogue only.  stswin_gemm_nt_splitk_scratch returns the floats of
 * workspace needed, or 0 when the shape is not a candidate (<= 32 tiles of 256x256, >= 64 stages of 32: call stswin_gemm_nt).
 * Environment STSWIN_SPLITK_BF16=1 (tuning, read per call): bf16 partial slabs - 20 % faster, the partials rounded once more. */
long stswin_gemm_nt_splitk_scratch(int M, int N, int Kseg, int S);
int stswin_gemm_nt_splitk(const void* A, long lda, const int* a_rows, const void* B, long ldb, void* C, long ldc, const float* bias, int M, int N,
                          int Kseg, int S, int relu, float* workspace, long workspace_floats, void* stream);
/* Fused split-K combine (the default where it applies; stswin_last_variant(1) then carries STSWIN_VAR_TN_FUSED): with bf16 operands, a
 * workspace, the 256x256 ring kernel and a grid of at most one workgroup per compute unit, the partial tiles are combined INSIDE the
 * launch - every workgroup of a tile signals an arrival counter; once the tile's partials are complete its row slices are handed out by
 * ticket to the workgroups of the tile that are present and added in split order (bit for bit the separate pass's result;
 * tests/test_hip_gemm.py).  No workgroup depends on another one being resident: an early workgroup polls for a bounded time
 * (200 us) and then leaves, the last arriver takes whatever slices are left - a second stream, a second process or an RCCL kernel
 * holding compute units costs time, never correctness, and nothing traps.  Counter regions are per (device, stream).
 * stswin_tn_fused_hold(+1 / -1 / 0) adds / releases / reads a process-wide hold: while any hold is outstanding the separate
 * tn_reduce pass is used (GradBucketReducer holds one while its collectives overlap backward); environment STSWIN_TN_FUSED=0 / 1
 * overrides the holds (read per call). */
int stswin_tn_fused_hold(int delta);
/* Compute units the one-workgroup-per-CU launches plan for (split-K factor of the gemm_tn ring kernel, persistent grids of the attention
 * backward): the whole device (256) by default; lower it while a communication kernel is known to hold k units (256 - k), 0 restores
 * the default.  Returns the previous setting.  A planning hint only: any value gives correct results. */
int stswin_set_cu_budget(int cus);
int stswin_gemm_tn(int dtype, const void* At, long lda, const int* at_rows, const void* Bt, long ldb, const int* bt_rows,
                   float* C, long ldc, int Mk, int Ni, int Nj, int splits, int bseg,
                   float* workspace /* optional caller-owned scratch: split-K partials are stored there and combined by a
                                       second kernel instead of fp32 atomics */, long workspace_floats, void* stream);
/* Up to 4 bf16 weight-gradient problems of ONE backward step in one launch (round 5).  The reference's autograd computes the weight
 * gradients of a Swin block one by one (swin_512.py:115-141,18-21 through torch.autograd: qkv, proj, fc1, fc2); each of those GEMMs has
 * few 256x256 output tiles (4-16 at stage 1), so alone it needs 16-64 contraction splits to fill the device and pays its own ramp,
 * partial-tile round trip and combine.  Launched together, the three that are ready at the end of a block's backward (fc1, proj, qkv)
 * are 32 tiles x 8 splits = 256 workgroups.  Every problem is one stswin_gemm_tn would give to the 256x256 ring kernel with the fused
 * combine (both output dims >= 256, at most one row map, bf16 partials, ldc >= 0); split counts are chosen so that every workgroup
 * gets about the same number of 32-row stages and the grid is at most one workgroup per compute unit (stswin_set_cu_budget).
 * Returns 0 when launched; STSWIN_TN_GROUP_DECLINED when the set is not eligible (a hold of stswin_tn_fused_hold is outstanding,
 * STSWIN_TN_FUSED=0 / STSWIN_TN_F32_SLABS=1 / STSWIN_TN_GROUP=0 in the environment, a shape outside the ring kernel's range, less than
 * 7/8 of the device filled, workspace too small) - nothing has been written and the caller launches the problems one by one with
 * stswin_gemm_tn; other negative codes: malformed arguments.  splits_out (optional, [count]): the split counts used.  The result of a
 * problem is a deterministic function of its operands and its split count (partials rounded to bf16 once, added in split order). */
#define STSWIN_TN_GROUP_DECLINED (-1050)
typedef struct stswin_tn_problem {
  const void* At; long lda; const int* at_rows;     /* [Mk (gathered by at_rows)][Ni] */
  const void* Bt; long ldb; const int* bt_rows;     /* [Mk (gathered by bt_rows; with bseg > 0: per-tap maps [Nj / bseg][Mk])][Nj or bseg] */
  float* C; long ldc;                               /* [Ni][Nj] fp32 */
  int Mk, Ni, Nj, bseg;
  int overwrite;                                    /* 1: C = result, 0: C += result */
  int tapminor;                                     /* with bseg > 0: store [row][channel][tap] (STSWIN_TN_OUT_TAPMINOR) */
} stswin_tn_problem;
int stswin_gemm_tn_group(int dtype, int count, const stswin_tn_problem* problems, float* workspace, long workspace_floats, int* splits_out,
                         void* stream);
/* The combine pass of a stswin_gemm_tn launched with STSWIN_TN_NO_COMBINE, for callers that run it on a second stream (it then
 * overlaps the input-gradient GEMM that follows every weight-gradient GEMM of a backward pass instead of standing between two
 * launches; ordering between the two streams is the caller's: events).  splits / slab_bf16 as stswin_last_variant(1) reported
 * them for that launch (no partials were written if it reported no slab bit: nothing to combine, C is final). */
int stswin_tn_combine(const float* workspace, float* C, long ldc, int Ni, int Nj, int splits, int overwrite, int slab_bf16,
                      void* stream);
/* Which kernel the launcher chose for the calling thread's most recent stswin_gemm_nt (family 0) / stswin_gemm_tn
 * (family 1) call: one of the codes below (family 1: | slab bits | split count << 16).  Test instrumentation only - the
 * parity suite asserts that the production shapes of the training step run the production kernels
 * (swin_512.py:109-141 qkv/proj, :7-23 Mlp, resnet.py:31-34 3x3 convolutions and their weight gradients). */
#define STSWIN_VAR_F32 100                 /* added to a 128x128-family code when the exact-fp32 instantiation ran */
#define STSWIN_VAR_NT_RING256_REGEPI 1     /* 256x256 ping-pong ring, operand-swapped MFMA + register epilogue (production) */
#define STSWIN_VAR_NT_RING256_LDSEPI 2     /* 256x256 ping-pong ring, fp32 LDS epilogue (fp32 outputs, unaligned operands) */
#define STSWIN_VAR_NT_RING256_NOPIPE 3
#define STSWIN_VAR_NT_RING256_W4 14        /* 256x256 ring, 4 waves of 128x128, register-pipelined main loop (GF_W4R / STSWIN_NT_W4=1) */
#define STSWIN_VAR_NT_STREAM 4
#define STSWIN_VAR_NT_DUO 5
#define STSWIN_VAR_NT_RING256x128_PP 6
#define STSWIN_VAR_NT_MID 7
#define STSWIN_VAR_NT_256x64 8
#define STSWIN_VAR_NT_128x64 9
#define STSWIN_VAR_NT_128x128 10
#define STSWIN_VAR_NT_128x128_W4 11
#define STSWIN_VAR_NT_SPLITK 13        /* 256x256 ring on tiles x splits + fp32 slab combine (stswin_gemm_nt_splitk); splits in bits 16.. */
#define STSWIN_VAR_NT_ROWS 12          /* M <= 8 rows: one wave per output column (ASPP.py:43-46 image-pool branch) */
#define STSWIN_VAR_TN_RING_PLAIN 20        /* gemm_tn_ring_kernel<0> */
#define STSWIN_VAR_TN_RING_ATROWS 21       /* <1> */
#define STSWIN_VAR_TN_RING_BTROWS 22       /* <2> */
#define STSWIN_VAR_TN_RING_BSEG 23         /* <3>: tap-segmented convolution weight gradient */
#define STSWIN_VAR_TN_128x128 30
#define STSWIN_VAR_TN_128x128_W4 31
#define STSWIN_VAR_TN_ROWS 32          /* Mk <= 8: a sum of outer products, no split-K slabs */
#define STSWIN_VAR_TN_SLABS_F32 0x1000     /* split-K partial slabs + tn_reduce, fp32 partials */
#define STSWIN_VAR_TN_SLABS_BF16 0x2000    /* ... bf16 partials */
#define STSWIN_VAR_TN_FUSED 0x8000         /* the split-K partials were combined inside the GEMM launch (no tn_reduce pass) */
#define STSWIN_VAR_TN_TAPMINOR 0x4000      /* the combine stored the result tap-minor (STSWIN_TN_OUT_TAPMINOR was honoured) */
#define STSWIN_TN_OUT_TAPMINOR (1 << 25)   /* bit of `splits` (with bseg > 0): store C[i][c * S + s] for GEMM column j = s * bseg + c, S = Nj / bseg - the [cout][cin][k][k]
                                            * layout of a convolution weight gradient; only where split-K slabs are combined (check stswin_last_variant) */
/* The QKV projection with an fp8 (OCP e4m3) result + one fp32 scale per (window problem, head of q / k / v): BASELINE.json configs[4]
 * ("fp8 MFMA attention"), swin_512.py:115-121 with q | k | v stored in 8 bits.  out8 [M][ld8] bytes, scales [M / rows_per_problem][ld_scales]
 * (column = n / head_dim: the heads of q, then of k, then of v); value = e4m3 byte * scale.  stswin_win_attn_fwd_f8 / _bwd_f8 read them. */
int stswin_gemm_nt_qkv_fp8(const void* A, long lda, const int* a_rows, const void* B, long ldb, void* out8, long ld8, float* scales,
                           long ld_scales, const float* bias, int M, int N, int Kseg, float scale, int scale_cols, int rows_per_problem,
                           int head_dim, void* stream);
/* Window attention on fp8-STORED q | k | v (stswin_gemm_nt_qkv_fp8's output): the core of swin_512.py:117-138 with both products on the fp8
 * MFMA, fp32 softmax; out bf16 [rows][ldo].  Geometries: (T ws^2, head dim) = (128, 128) and (32, 256) - the two stages of the model;
 * anything else returns -1201 (the caller keeps bf16 storage there). */
int stswin_win_attn_fwd_f8(const void* qkv8, long ld8, const float* scales, long ld_scales, void* out, long ldo, const float* biasT,
                           const float* maskT, int nB_, int nW, int T_frames, int ws, int heads, int C, int bias_windows,
                           const int* bias_index, void* stream);
/* ... and its backward: dqkv (bf16 [rows][lddq], the gradient of the dequantised q | k | v), dbiasT / dqkv_colsum as stswin_win_attn_bwd; scratch =
 * stswin_win_attn_bwd_scratch(nB_, ws, heads, C) floats.  Geometries: 8x8 windows x 2 frames x head dim 128, 4x4 x 2 x 256. */
int stswin_win_attn_bwd_f8(const void* qkv8, long ld8, const float* scales, long ld_scales, const void* dout, long lddo, void* dqkv, long lddq,
                           const float* biasT, const float* maskT, float* dbiasT, float* dqkv_colsum, int nB_, int nW, int T_frames, int ws,
                           int heads, int C, float scale, int bias_windows, const int* bias_index, float* scratch, long scratch_floats,
                           void* stream);
int stswin_last_variant(int family);
/* 1: the library was built with -DSTSWIN_TUNING (STSWIN_TUNING=1 python __graft_entry__.py --force) and holds the A/B-only gemm_nt variants
 * (STSWIN_GF_MID / _HALF / _NOPIPE / duo / stream: each measured slower than the default dispatch); 0: the product build, which ignores
 * those flags and picks the default kernel. */
int stswin_tuning_build(void);

/* ---- deterministic cross-workgroup sums.  No kernel of this library adds fp32 values with atomics any more: every kernel that sums
 * over workgroups (bias / LayerNorm / BatchNorm parameter gradients, BatchNorm statistics, the relative-position-bias gradient, the
 * OHEM statistics) stores one partial vector per workgroup ("slab") into caller-owned scratch and stswin_slab_fold's kernel adds the
 * slabs in a fixed order behind the launch boundary, so a training step gives the same bits every time it runs.  The `scratch`
 * arguments below are that space: uninitialised fp32, at least stswin_<entry>_scratch(...) floats, 16-byte aligned, private to the
 * call until the next kernel on the stream has run (one buffer per stream can serve every call).
 *   out_s[b*obs + j] (+)= sum_{p < nslabs} ws[b*ws_batch_stride + p*slab_stride + s*seg_len + j],  s < nseg <= 3, j < seg_len */
int stswin_slab_fold(const float* ws, long slab_stride, long ws_batch_stride, int nslabs, int seg_len, int nseg, float* out0, float* out1,
                     float* out2, long out_batch_stride, int batch, int accumulate, void* stream);

/* stswin_fold_defer(1): from now on (calling thread) the folds of the entry points are QUEUED instead of launched - each call must
 * then get its own scratch region - and stswin_fold_flush launches them as one kernel; stswin_fold_defer(0) flushes and returns to
 * immediate folds.  Legal when no folded output is read before the flush (the backward of a Swin block: all parameter gradients). */
int stswin_fold_defer(int on, void* stream);
int stswin_fold_flush(void* stream);

/* out[n] += sum_m Y[m][n]  (bias gradients) */
long stswin_colsum_scratch(int M, int N);
int stswin_colsum(int dtype, const void* y, long ldy, float* out, int M, int N, float* scratch, void* stream);

/* ---- nn.LayerNorm (swin_512.py:160,166 norm1/norm2; :253,:274 PatchMerging.norm with the 2x2 gather fused:
 * row r = concat_s x[rows[s][r]][0:Cseg]).  mean/rstd (fp32 [M]) are saved for the backward. */
int stswin_layernorm_fwd(int dtype, const void* x, long ldx, const int* rows, int S, int Cseg, void* y, long ldy,
                         const float* gamma, const float* beta, float* mean, float* rstd, int M, float eps, void* stream);
int stswin_layernorm_bwd(int dtype, const void* dy, long lddy, const void* x, long ldx, const int* rows, int S, int Cseg,
                         const float* gamma, const float* mean, const float* rstd, void* dx, long lddx, float* dgamma,
                         float* dbeta, int M, int accumulate_dx,
                         float* dxsum /* optional fp32 [S*Cseg]: += column sums of the dx written */,
                         float* scratch /* >= stswin_layernorm_bwd_scratch(M, S*Cseg) floats: one [3][S*Cseg] slab (dgamma | dbeta |
                                           dxsum partial sums) per workgroup; folded into dgamma / dbeta / dxsum (+=) in slab order */,
                         void* stream);
/* The same with the accumulated-into tensor as a separate, read-only operand: dx = add + LN'(dy).  The backward of a pre-norm block
 * (swin_512.py:228-234: x = shortcut + attn; x = x + mlp(norm2(x))) then leaves the incoming gradient `add` intact for the weight-gradient
 * GEMM that reads it later (stswin_gemm_tn_group). */
int stswin_layernorm_bwd_add(int dtype, const void* dy, long lddy, const void* x, long ldx, const int* rows, int S, int Cseg,
                             const float* gamma, const float* mean, const float* rstd, const void* add, long ldadd, void* dx, long lddx,
                             float* dgamma, float* dbeta, int M, float* dxsum, float* scratch, void* stream);
long stswin_layernorm_bwd_scratch(int M, int C);

/* out[i] = map[i] >= 0 ? v[map[i]] : fill, i < n: padding of per-channel parameter vectors (BatchNorm weight / bias / running
 * statistics) to the 64-aligned channel layout of the token matrices and the gather back (base18.py:60-77 concat layout). */
int stswin_vec_gather(const float* v, const int* map, float* out, int n, float fill, void* stream);
/* the same for `count` (<= 4) vectors that share the map, in one launch: out[k][i] = map[i] >= 0 ? v[k][map[i]] : fill[k]
 * (v, out, fill: host arrays).  out[k] may be a parameter / buffer itself (the un-padded running statistics are written in place). */
int stswin_vec_gather_multi(int count, const float* const* v, const int* map, float* const* out, int n, const float* fill, void* stream);

/* ---- relative position bias (swin_512.py:122-131).  expand: out[w][h][j][i] = table[index[i*N + j]][h] (+ mask[w][i][j]),
 * the [key][query] layout of stswin_win_attn_fwd's biasT (nW = 1, mask = NULL: plain [heads][N][N]; with the SW-MSA mask the
 * per-window pre-summed table).  scatter: dtable[index[i*N + j]][h] += dbiasT[h][j][i] (the backward of the gather).
 * table / dtable fp32 [(2ws-1)^2][heads], index int64 [N*N] (the module's relative_position_index buffer). */
int stswin_bias_expand(const float* table, const long* index, const float* mask, float* out, int N, int heads, int nW,
                       void* stream);
/* the same for `count` (<= 16) tables in one launch (host arrays of device pointers / sizes; mask[e] may be NULL): every Swin block's
 * table right after an optimizer step instead of one launch per block forward */
int stswin_bias_expand_multi(int count, const float* const* table, const long* const* index, const float* const* mask, float* const* out,
                             const int* N, const int* heads, const int* nW, void* stream);
/* scatter in gather form (no atomics: thread (table row e, head h) adds the pairs of row e in a fixed order).  order int32 [N*N] =
 * the pairs i*N + j sorted by index[i*N + j] (stable), offs int32 [table_rows + 1] = start of every table row's range in it
 * (table_rows = (2ws-1)^2); both are functions of the index buffer alone (stswincl_amd/hip.py caches them per buffer).
 * nslabs > 1: dbiasT is [nslabs][heads][N][N] and the slabs are added on the way. */
int stswin_bias_scatter(const float* dbiasT, const int* order, const int* offs, float* dtable, int N, int heads, int table_rows, int nslabs,
                        void* stream);

/* ---- a6: windowed attention core (swin_512.py:117-138).  qkv [nB_*T*N][3C] = q (pre-scaled) | k | v in window
 * order; biasT [heads][N][N] and maskT [nW][N][N] are the expanded relative-position bias (:122-124) and the
 * SW-MSA mask (:126-131), both transposed to [key][query]; out [nB_*T*N][C] is the (B_, T*N, heads*d) layout of :136.
 * bwd: dqkv gets (scale*dS k | dS^T q_s | P^T dO); dbiasT [heads][N][N] += the bias gradient (per-workgroup partial slabs in
 * `scratch`, folded in a fixed order: deterministic).
 * bias_windows = 1: biasT is [heads][N][N] and maskT (or NULL) is added per window as in :126-131.
 * bias_windows = nW: biasT is [nW][heads][N][N] = bias + mask already summed by the caller, maskT must be NULL
 * (one table read per score instead of two).
 * bias_windows = U with bias_index (int [nW], values < U): biasT is [U][heads][N][N] and window w uses slot bias_index[w] - the
 * SW-MSA mask has only 4 distinct window patterns (interior, last column, last row, corner), so the 64-window table of
 * stage 1 (64 MB) shrinks to 4 MB and stays in L2. */
int stswin_win_attn_fwd(int dtype, const void* qkv, long ld, void* out, long ldo, const float* biasT, const float* maskT,
                        int nB_, int nW, int T_frames, int ws, int heads, int C, int bias_windows,
                        const int* bias_index, void* stream);
/* QKV-fused forward (swin_512.py:115-141 in one kernel; bf16, T * ws * ws = 128 tokens per window, head dim 128: the stage-1 shape):
 * out = attention(gather(x, rmap) . qkv.weight^T + qkv.bias) per (window, head), where the projection, the q scaling, the bias /
 * mask table and the softmax never leave the CU.  x [.][C] token rows, rmap int [nB_ * 128] = token row of every window row (-1:
 * zero row; NULL: identity), w = qkv.weight [3C][C] (bf16), bqkv fp32 [3C] or NULL, biasT as in stswin_win_attn_fwd with
 * bias_windows / bias_index (the pre-summed bias + mask table; no separate maskT).  qkv_out (optional, [nB_ * 128][3C]) receives
 * q * scale | k | v for stswin_win_attn_bwd; NULL in no-grad passes, which then never write q, k, v to memory.
 * x_rows = token rows of x (every rmap value is < x_rows; 0: nB_ * 128).  The kernel addresses x with 32-bit byte offsets:
 * x_rows * ldx * 2 > 0xFFFF0000 returns -1208 and launches nothing (the caller runs stswin_gemm_nt + stswin_win_attn_fwd). */
int stswin_win_attn_qkv_fwd(const void* x, long ldx, long x_rows, const int* rmap, const void* w, long ldw, const float* bqkv, void* qkv_out,
                            long ldq, void* out, long ldo, const float* biasT, int nB_, int nW, int T_frames, int ws, int heads, int C,
                            float scale, int bias_windows, const int* bias_index, void* stream);
/* BASELINE.json configs[4] "fp8 MFMA attention": the forward above with q, k, v and the probabilities quantised to OCP e4m3 in
 * registers (per (window, head) amax scales, P x 128) and both products on v_mfma_f32_32x32x16_fp8_fp8; qkv / out stay bf16 in
 * memory (same arguments, dtype fixed to bf16 storage).  A numerics mode (the non-scaled fp8 MFMA has the bf16 rate on gfx950);
 * the backward is stswin_win_attn_bwd on the unquantised qkv (straight-through). */
int stswin_win_attn_fwd_fp8(const void* qkv, long ld, void* out, long ldo, const float* biasT, const float* maskT, int nB_, int nW,
                            int T_frames, int ws, int heads, int C, int bias_windows, const int* bias_index, void* stream);
int stswin_win_attn_bwd(int dtype, const void* qkv, long ld, const void* dout, long lddo, void* dqkv, long lddq,
                        const float* biasT, const float* maskT, float* dbiasT,
                        float* dqkv_q_colsum /* optional fp32 [C]: += column sums of the dq third (q bias gradient).
                           The other two thirds need no pass over dqkv: sum_rows dk = 0 exactly (rows of dS sum to
                           zero) and sum_rows dv = column sums of dout (softmax rows sum to one) */,
                        int nB_, int nW, int T_frames, int ws,
                        int heads, int C, float scale, int bias_windows, const int* bias_index,
                        float* scratch /* >= stswin_win_attn_bwd_scratch(nB_, ws, heads, C) floats */, long scratch_floats, void* stream);
long stswin_win_attn_bwd_scratch(int nB_, int ws, int heads, int C);

/* ---- decode head on NHWC token matrices [M = frames*H*W][C]  (ASPP.py:33-52, base18.py:60-106) ---------------
 * Grouped BatchNorm2d: rows are `groups` equal groups with separate batch statistics (1 for the head; the number of
 * frames when the per-frame ResNet calls of base18.py:86-89 are batched).  colstats accumulates pivot-shifted sums
 * (sumsq may be NULL: plain grouped column sums = adaptive_avg_pool numerator, ASPP.py:43); bn_finalize turns them
 * into mean / rstd and applies nn.BatchNorm2d's running-stat update group by group; bn_apply fuses affine +
 * residual + ReLU (resnet.py:42-51); bn_bwd = the two-pass backward (s1/s2 fp32 [groups][C], zeroed by the caller).
 * unit_rows = 0: the groups are contiguous row blocks.  unit_rows > 0: group g owns the units g, g + groups, g + 2 groups, ...
 * of unit_rows rows each - frame t of every clip when the 4-frame clips are stored clip-major, so that the per-frame
 * statistics of base18.py:86-89 need no frame-major copy of the batch. */
int stswin_colstats(int dtype, const void* x, long ldx, float* sum /* += */, float* sumsq /* += */, int M, int C, int groups, int unit_rows,
                    float* scratch /* >= stswin_colstats_scratch(...) floats */, void* stream);
long stswin_colstats_scratch(int dtype, int M, int C, int groups, int unit_rows);
/* sum / sumsq [groups][N] of the statistic groups from the two-plane table a STSWIN_GF_CS_SQ stswin_gemm_nt wrote (same M, N; groups
 * of whole 256-row tiles: contiguous, or interleaved units of unit_rows rows).  Then stswin_bn_finalize with x = NULL (raw sums, no
 * pivot) replaces stswin_colstats: nn.BatchNorm2d batch statistics (resnet.py:42-51, ASPP.py:37-50) without a pass over the tensor. */
int stswin_cs_group_reduce(const float* table, int M, int N, int groups, int unit_rows, float* sum, float* sumsq, void* stream);

/* BatchNorm(train) statistics from a gemm_nt STSWIN_GF_CS_SQ table in one launch: mean / rstd [groups][N] and the running
 * statistic updates applied group by group in order (the sequential per-frame BatchNorm calls of seg18/net/Ours/base18.py:86-89;
 * torch.nn.BatchNorm2d semantics: biased variance to normalise, unbiased into running_var).  Same group geometry rules as
 * stswin_cs_group_reduce; groups <= 32.  running_* may be NULL. */
int stswin_bn_table_finalize(const float* table, int M, int N, int groups, int unit_rows, float* mean, float* rstd,
                             float* running_mean, float* running_var, float eps, float momentum, void* stream);
int stswin_bn_finalize(int dtype, const void* x, long ldx, const float* sum, const float* sumsq, float* mean, float* rstd,
                       float* running_mean, float* running_var, int M, int C, int groups, float eps, float momentum,
                       int unit_rows, void* stream);
int stswin_bn_apply(int dtype, const void* x, long ldx, const float* mean, const float* rstd, const float* gamma,
                    const float* beta, const void* resid, long ldr, void* y, long ldy, int M, int C, int groups, int relu,
                    int unit_rows, void* stream);
int stswin_bn_bwd(int dtype, const void* dy, long lddy, const void* x, long ldx,
                  const void* y /* stored output (ReLU mask); NULL with relu: the mask is recomputed from x, needs beta */,
                  long ldy, const float* mean, const float* rstd, const float* gamma, const float* beta /* may be NULL if y is given */,
                  float* s1, float* s2, void* dx, long lddx, void* dresid, long lddr,
                  int M, int C, int groups, int relu, int training,
                  int phase /* 0 both passes, 1 reduce only, 2 dx only: SyncBatchNorm all-reduces s1/s2 in between */,
                  long rows_total /* rows per group over all ranks (0 = local) */, int unit_rows,
                  float* group_sums /* optional fp32 [2][C], written by the dx pass: s1 | s2 summed over the groups = the
                                       bias | weight gradients of the BatchNorm (ASPP.py:37-50 etc.: autograd of nn.BatchNorm2d) */,
                  float* scratch /* phases 0 / 1: >= stswin_bn_bwd_scratch(...) floats (per-chunk partial sums, folded into s1 / s2) */,
                  void* stream);
long stswin_bn_bwd_scratch(int dtype, int M, int C, int groups, int unit_rows);
/* BatchNorm + ReLU + nn.MaxPool2d(3, 2, 1) in one pass (torchvision resnet18 conv1 -> bn1 -> relu -> maxpool, resnet.py:98-102):
 * x [frames*H*W][C] = the BatchNorm's input, out [frames*Hp*Wp][C] (Hp = (H-1)/2 + 1), arg (uint8, same shape) = winning tap per
 * value (first maximum in (ky, kx) scan order).  Candidates are stswin_bn_apply's values rounded to the compute dtype, so out and
 * arg equal those of stswin_bn_apply + stswin_maxpool3x3s2 (whose backward form takes arg).  Statistic groups as in stswin_bn_apply
 * (whole frames per group). */
int stswin_bn_relu_pool(int dtype, const void* x, long ldx, const float* mean, const float* rstd, const float* gamma, const float* beta,
                        void* out, long ldo, unsigned char* arg, int frames, int H, int W, int C, int groups, int unit_rows, void* stream);
/* out[r][c] (+)= v[r / (M/groups)][c] * scale : image-pool broadcast (ASPP.py:46) and avg-pool backward */
int stswin_rows_broadcast(int dtype, const float* v, void* out, long ldo, int M, int C, int groups, float scale,
                          int accumulate, void* stream);
/* bilinear, align_corners=False (base18.py:102-103): forward in[F][h][w][C] -> out[F][H][W][C]; backward: in = d(out),
 * out = d(in) (gather form, no atomics) */
int stswin_bilinear(int dtype, const void* in, long ldi, void* out, long ldo, int frames, int h, int w, int H, int W, int C,
                    int backward, void* stream);
/* final interpolate of base18.py:106: tokens [F][h][w][nc] <-> NCHW logits [F][nc][H][W] */
int stswin_logits_upsample(int dtype, const void* tokens, long ldt, void* nchw, int frames, int h, int w, int H, int W,
                           int nc, int backward, void* stream);

/* ---- a15: OHEM cross entropy (seg18/utils/losses.py:32-40).  ce_fwd: per-pixel CE (ignore_index -> 0) and
 * stats (fp32 [4], zeroed by the caller, 8-byte aligned): [0] = #(loss > thresh) (an exact integer count), [2..3] = one unsigned 64-bit
 * sum of those losses in 2^-32 fixed point (integer atomics: associative, so the value is reproducible).  ce_bwd: dlogits = gscale[0] * sel[1] * (softmax - onehot) for
 * pixels with loss > sel[0] (>= when sel[2] != 0); sel/gscale live on the device, so no host sync is needed. */
int stswin_ce_fwd(int dtype, const void* logits, const long* labels, float* loss, float* stats, int frames, long HW, int nc,
                  int ignore_index, float thresh, void* stream);
/* The selection of losses.py:35-39 without the sort: value[0] = OHEM loss, sel[3] = (cut, 1/count, inclusive) for ce_bwd.
 * stats = output of ce_fwd; if stats[0] = #(loss > thresh) > n_min the threshold branch is taken, otherwise the mean of the
 * n_min largest losses is computed by an exact 3-level radix select (counts + sums per bin).  work: caller-owned scratch
 * of >= STSWIN_OHEM_WORK_BYTES bytes (zeroed by the call). */
#define STSWIN_OHEM_WORK_BYTES (3 * 2048 * 12 + 48)
int stswin_ohem_select(const float* loss, long n, long n_min, float thresh, const float* stats, void* work, long work_bytes,
                       float* value, float* sel, void* stream);
int stswin_ce_bwd(int dtype, const void* logits, const long* labels, const float* loss, const float* sel,
                  const float* gscale, void* dlogits, int frames, long HW, int nc, int ignore_index, void* stream);

/* ---- a16/a17: label-guided pixel-contrastive similarity (pixcontrast_18/contrast/models/PixPro_swin_v5.py:71-129).
 * Q [N][HW][C] query embeddings, K5[j] [N][HW][C] the five key maps (k, adj1, adj2, adj3, neg3), lq / lk5[j] int32
 * labels [N][HW].  Writes pos[n][i][j] = sum_p (q_i . k_jp) [lq_i == lk_jp] and all[n][i][j] = sum_p q_i . k_jp
 * (fp32 [N][HW][5]); the HW x HW logits / posMask / negMask tensors of :82-113 are never materialised. */
int stswin_contrast_fwd(int dtype, const void* Q, long ldq, const void* const* K5, long ldk, const int* lq,
                        const int* const* lk5, float* pos, float* all, int N, int HW, int C, void* stream);

/* Bank mode of the same loss: every query pixel against a bank of key embeddings, ONE launch for both loss directions and all
 * key maps (PixPro_swin_v5.py:594-595 = two regression_loss calls, :71-129), and the inter-video bank that the reference
 * sketches in its unused dist_collect (pixcontrast_18/contrast/util.py:47-58).
 *   Q [M][C] + lq[M]: the M query rows are q_sets (1 or 2: the loss directions) equal sets of nblk = M / (q_sets q_block) blocks of
 *     q_block rows (a block = one sample's HW pixels in the reference's per-sample mode; q_block = M / q_sets: one block, every
 *     query sees the whole segment);
 *   bank [maps][seg][C] + lb[maps][seg]: query set s uses map gmap[s*groups + g] as its group g; a query of block b sees rows
 *     [b bank_block, (b+1) bank_block) of that map (bank_block = seg / nblk, or = seg when nblk = 1).
 * Writes pos[m][g] = sum_p S[m][p] [lq[m] == lb[p]], all[m][g] = sum_p S[m][p] over the visible rows of group g (fp32 [M][groups]) and,
 * if not NULL, rowmax[m] / lse[m] = max / log-sum-exp of inv_tau * S[m][p] over the visible rows of ALL groups (the InfoNCE
 * denominator; monitoring and hard-negative statistics - the reference loss itself is linear in S).  C <= 256, a multiple of 64
 * (bf16) / 32 (fp32).  workspace: caller-owned fp32 scratch, >= 4 M groups floats (more lets the launcher split long bank segments
 * over workgroups; the partials are combined in a fixed order, so results are deterministic). */
int stswin_contrast_bank_fwd(int dtype, const void* Q, long ldq, const int* lq, int M, int C, int q_sets, int q_block,
                             const void* bank, long ldb, const int* lb, int maps, int seg, int bank_block, int groups,
                             const int* gmap /* host memory, [q_sets][groups] */, float inv_tau, float* pos, float* all,
                             float* rowmax, float* lse, float* workspace, long workspace_floats, void* stream);
/* The same for L2-NORMALISED query and bank rows (the embeddings ConsistencyLoss feeds it: F.normalize, PixPro_swin_v5.py:369-557):
 * |S| <= 1, so rowmax / lse are formed with inv_tau as a FIXED reference point of the sum of exponentials - one fma + one exponential
 * per score instead of a running maximum with rescaling (same values to fp32 rounding; the caller guarantees the norms). */
int stswin_contrast_bank_fwd_unit(int dtype, const void* Q, long ldq, const int* lq, int M, int C, int q_sets, int q_block,
                             const void* bank, long ldb, const int* lb, int maps, int seg, int bank_block, int groups,
                             const int* gmap /* host memory, [q_sets][groups] */, float inv_tau, float* pos, float* all,
                             float* rowmax, float* lse, float* workspace, long workspace_floats, void* stream);
/* Backward to the queries (keys are no-grad, PixPro_swin_v5.py:366): the masked sums are linear in the scores, so
 * dq[m] = sum_g dpos[m][g] Kcls[map(g)][blk][lq[m]] + dneg[m][g] (Ktot[map(g)][blk] - Kcls[..][lq[m]]), where dpos / dneg are the
 * gradients of pos and of neg = all - pos.  class_sums writes ksum [maps][seg / bank_block][ncls + 1][C] fp32 (slot ncls = all
 * rows; zeroed by the call; labels outside [0, ncls) only count in the total); bank_dq combines them per query row into dq
 * fp32 [M][C].  cnt[m][g] (fp32) = number of visible rows of group g with label lq[m]: where it equals bank_block the negative
 * set is empty and its term is skipped, so the gradient is exactly zero like the reference's masked products (:103-113). */
int stswin_contrast_class_sums(int dtype, const void* bank, long ldb, const int* lb, int maps, int seg, int bank_block, int C,
                               int ncls, float* ksum, float* scratch /* >= stswin_contrast_class_sums_scratch(...) floats */, void* stream);
long stswin_contrast_class_sums_scratch(int maps, int seg, int bank_block, int C, int ncls);
int stswin_contrast_bank_dq(const float* dpos, const float* dneg, const float* cnt, const int* lq, const float* ksum, float* dq,
                            long lddq, int M, int C, int q_sets, int q_block, int seg, int bank_block, int ncls, int groups, const int* gmap,
                            void* stream);

/* ---- the glue of the contrastive step (round 5; pixcontrast_18/contrast/models/PixPro_swin_v5.py): what the reference does in ~125
 * elementwise torch launches per step.
 * stswin_rownorm_scatter: Y[(view, sample, pixel)] = X[row] / max(||X[row]||_2, 1e-12), fp32 arithmetic, compute-dtype result
 *   (F.normalize(proj.float(), dim=1) of PixPro._embed, :369-557) with the batched views de-interleaved: X rows are clip-major
 *   ((sample * views + view) * HW + pixel), Y is the view-major [views][samples * HW][C] matrix the pair loss reads (its key bank /
 *   query matrix: no NCHW permute, slice, stack or cat in between).  inv (optional) keeps 1 / norm per row for the backward:
 *   dX = inv (dY - y (y . dY)).  C % 64 == 0, C <= 1024, R == views * samples * HW.
 * stswin_labels_resize: `maps` (<= 8) float label maps [N][1][Hs][Ws] -> int32 lb[maps][N * h * w], nearest neighbour with ATen's index
 *   rule (F.interpolate(mode='nearest') + .to(int32), ConsistencyLoss.forward :590-593).
 * stswin_label_counts: cnt[m][g] = number of rows of bank block blk(m) of map gmap[set(m)][g] whose label equals lq[m] (labels clamped
 *   to [0, ncls - 1]): the row sums of posMask (:48-57, :116-118) from per-block class histograms (hist: int scratch
 *   [maps][seg / bank_block][ncls]).
 * stswin_pair_loss / _bwd: loss = sum over the query sets of mean(-log(e^P / (e^P + e^N) + 1e-6)), P = sum_g pos_g / (sum_g cnt_g + 1e-6),
 *   N = sum_g (all_g - pos_g) / (visible - cnt_g + 1e-6) (:119-129); one workgroup, fixed-order sums (deterministic); the backward gives
 *   d loss / d pos and d loss / d (all - pos). */
int stswin_rownorm_scatter(int dtype, const void* X, long ldx, void* Y, long ldy, float* inv, int R, int C, int views, int HW, int samples,
                           void* stream);
int stswin_rownorm_scatter_bwd(int dtype, const void* X, long ldx, const float* inv, const float* dY, long lddy, void* dX, long lddx, int R,
                               int C, int views, int HW, int samples, void* stream);
int stswin_labels_resize(const float* const* masks, int maps, int N, int Hs, int Ws, int h, int w, int* lb, void* stream);
int stswin_label_counts(const int* lq, const int* lb, int M, int maps, int seg, int q_sets, int q_block, int bank_block, int ncls, int groups,
                        const int* gmap, int* hist, float* cnt, void* stream);
int stswin_pair_loss(const float* pos, const float* all, const float* cnt, int M, int groups, int q_sets, int visible, float* loss,
                     void* stream);
int stswin_pair_loss_bwd(const float* pos, const float* all, const float* cnt, const float* dloss, int M, int groups, int q_sets, int visible,
                         float* dpos, float* dneg, void* stream);

/* ---- f4: inference post-processing (seg18/test.py:153-158 + utils/EndoMetric.py): labels[f][y][x] = argmax_c of the
 * bilinear (align_corners = True) resize of NCHW logits [F][nc][h][w] to (H, W); with gt (int64 [F][H][W]) also
 * counts[f][0|1|2][c] = |gt == c|, |pred == c|, |gt == c and pred == c|  (int32, zeroed by the caller). */
int stswin_upsample_argmax(int dtype, const void* logits, unsigned char* labels, const long* gt, int* counts, int frames,
                           int nc, int h, int w, int H, int W, void* stream);

/* ---- f2: multi-tensor optimizer / EMA step, up to 48 fp32 tensors per launch (host arrays of device pointers).
 * mode 0 = torch.optim.Adam (seg18/train_swin.py:122; c1 = 1 - b1^t, c2 = sqrt(1 - b2^t)), 1 = torch.optim.SGD with
 * momentum b1 (train_CL_ft_mswin_sgd_minput.py:147-162; c1 != 0 marks the first step: buf = grad), 2 = EMA
 * p = p*b1 + g*(1-b1) with g = the query parameter (PixPro_swin_v5.py:258-289). */
int stswin_multi_tensor(int mode, int count, void* const* p, const void* const* g, void* const* m, void* const* v,
                        const int* n, float lr, float b1, float b2, float eps, float wd, float c1, float c2, void* stream);

/* LARS over SGD-momentum (pixcontrast_18/contrast/lars.py:109-152 wrapping torch.optim.SGD, main_pretrain_swinv5.py:37-47) for up
 * to 48 fp32 tensors of ONE parameter group, three launches: per-block partial ||p||^2 and ||g + wd p||^2, their fixed-order fold
 * into `norms` (caller-owned fp32 scratch, norms_floats >= 2 * count + 2 * sum_i ceil(n_i / 8192); no atomics: reproducible), then g' = (g + wd p) * (adaptive && both norms > 0 ? trust_coef ||p|| / (||g'|| + eps) : 1),
 * buf = first ? g' : momentum * buf + g', p -= lr * buf.  adaptive = 0: the 'ignore' group of add_weight_decay (biases, norms). */
int stswin_multi_tensor_lars(int count, void* const* p, const void* const* g, void* const* m, const int* n, float* norms,
                             long norms_floats, float lr, float momentum, float wd, float trust_coef, float eps, int first, int adaptive,
                             void* stream);

/* Step-dependent optimizer scalars in DEVICE memory, so that a hipGraph replay of a training step advances like eager steps do
 * (the reference's loops step Adam / LARS / the key-encoder momentum from host state every iteration: seg18/train_swin.py:122,171-173,
 * pixcontrast_18/main_pretrain_swinv5.py:37-47, contrast/models/PixPro_swin_v5.py:258-289).  hyper = fp32 [4] {lr, c1, c2, EMA momentum}.
 * stswin_optim_tick (one thread, double precision like the host expressions it replaces):
 *   kind 0: t = ++counter[0]; hyper[1] = 1 - a^t; hyper[2] = sqrt(1 - b^t)                   (Adam bias corrections, a, b = betas)
 *   kind 1: k = counter[0]++; hyper[3] = 1 - (1 - a) (cos(pi k / b) + 1) / 2                  (PixPro_swin_v5.py:260; a = base momentum, b = K)
 * stswin_multi_tensor_dev / stswin_multi_tensor_lars_dev: stswin_multi_tensor / stswin_multi_tensor_lars reading lr (hyper[0]), the Adam
 * corrections (hyper[1], hyper[2]) or the EMA momentum (hyper[3]) from `hyper` instead of their arguments; `first` as c1 != 0 above.
 * hyper[0] is written by the host with a stream-ordered fill whenever the scheduler changes the learning rate. */
int stswin_optim_tick(int kind, int* counter, float* hyper, double a, double b, void* stream);
int stswin_multi_tensor_dev(int mode, int count, void* const* p, const void* const* g, void* const* m, void* const* v, const int* n,
                            const float* hyper, float b1, float b2, float eps, float wd, int first, void* stream);
int stswin_multi_tensor_lars_dev(int count, void* const* p, const void* const* g, void* const* m, const int* n, float* norms,
                                 long norms_floats, const float* hyper, float momentum, float wd, float trust_coef, float eps, int first,
                                 int adaptive, void* stream);

/* ---- device self-test of the MFMA / LDS primitives the kernels are built on; writes a report into `out`
 * (fp32, >= 64 KiB) and returns the number of failed checks (0 = all good). Used by tests only. */
int stswin_selftest(float* out, int which, void* stream);
/* In-run calibration probes of bench.py (measurement infrastructure, never on the product path): stswin_calib_mfma launches 256
 * workgroups of `waves_per_cu` waves, each running `iters` rounds of 8 independent v_mfma_f32_16x16x32_bf16, and returns the MFMA
 * instructions per round of the whole launch (flops = that x iters x 16384); stswin_calib_copy copies `bytes` with 16-byte lanes. */
long stswin_calib_mfma(int waves_per_cu, int iters, float* sink, void* stream);
int stswin_calib_copy(const void* src, void* dst, long bytes, void* stream);
/* Measurement stand-in for an RCCL all-reduce of one gradient bucket as the main stream sees it (tools/overlap_proxy.py): `workgroups`
 * workgroups of 256 threads copy src -> dst (`bytes`, 16-byte pieces) `passes` times, i.e. hold that many compute units for that long.
 * No peer traffic; never on the product path. */
int stswin_proxy_collective(const void* src, void* dst, long bytes, int workgroups, int passes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
