"""ctypes binding of libstswin_hip.so (the C ABI declared in include/stswin_hip.h).

There is NO fallback: every wrapper raises if the library is missing or a kernel reports an error, and
every wrapper demands CUDA (ROCm) tensors.  PyTorch is used only for device memory and the stream handle.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Optional

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("STSWIN_HIP_LIB") or os.path.join(_ROOT, "stswincl_amd", "lib", "libstswin_hip.so")   # override: A/B of two builds in one process pool (tools/)
HEADER_PATH = os.path.join(_ROOT, "include", "stswin_hip.h")
_lib: Optional[ctypes.CDLL] = None

GF_GELU, GF_RESID, GF_MUL_DGELU, GF_OUT_F32, GF_ACCUM, GF_RELU, GF_WAVES4, GF_BIG, GF_NOBIG, GF_MID, GF_NOPIPE, GF_HALF, GF_ROT = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096
GF_MUL_R, GF_C2_DGELU, GF_CS_PARTIAL = 8192, 16384, 32768
GF_CS_SQ = 1 << 16
GF_NOREGEPI = 1 << 22
GF_NOSTREAM = 1 << 23
GF_DUO = 1 << 24
GF_STREAM = 1 << 25
GF_NONARROW = 1 << 26
GF_TAPSKIP = 1 << 29
GF_M32PP = -(1 << 31)        # tuning builds: the 8-wave ping-pong ring on 32x32x16 MFMA tiles (bit 31 of the int flags word)
GF_W4R = 1 << 30            # 256x256 ring: 4 waves of 128x128, register-pipelined main loop (32x32x16 MFMA)
GF_NODEEP, GF_DEEP = 1 << 27, 1 << 28        # tuning: prefetch depth of the 128x64 / 128x128 / 256x64 kernels
TN_OVERWRITE = 1 << 27

_c_int, _c_long, _c_float, _c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p


class StswinHipError(RuntimeError):
    pass


def declared_symbols():
    """Every entry point include/stswin_hip.h declares."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long)\s+(stswin_\w+)\s*\(", text)))


def _long_symbols():
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return set(re.findall(r"\blong\s+(stswin_\w+)\s*\(", text))


def load() -> ctypes.CDLL:
    """dlopen the library and check that it exports the whole declared ABI (works without a GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StswinHipError(f"{LIB_PATH} not found: run `python __graft_entry__.py` (build()) first; "
                             "stswincl_amd has no CPU or eager fallback")
    lib = ctypes.CDLL(LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    if missing and not os.environ.get("STSWIN_HIP_LIB"):     # (an older build named for an A/B run may lack new entries)
        raise StswinHipError(f"libstswin_hip.so lacks declared symbols: {missing}")
    longs = _long_symbols()
    for s in declared_symbols():
        if s not in missing:
            getattr(lib, s).restype = _c_long if s in longs else _c_int
    _lib = lib
    return lib


def tuning_build() -> bool:
    """True when the loaded library was built with STSWIN_TUNING=1 (A/B-only kernel variants and the in-kernel diagnosis switches)."""
    return bool(load().stswin_tuning_build())


def tn_fused_holds() -> int:
    """Outstanding holds on the fused split-K combine of stswin_gemm_tn (0 = fused where it applies)."""
    return int(load().stswin_tn_fused_hold(0))


class TnFusedHold:
    """While alive (until release()), weight-gradient GEMMs use the separate split-K combine pass instead of the one fused into the
    launch.  Refcounted inside the library (include/stswin_hip.h, stswin_tn_fused_hold): several holders, any order of release, and a
    holder dropped without release() gives its hold back when it is collected."""

    def __init__(self):
        import weakref
        lib = load()
        lib.stswin_tn_fused_hold(1)
        self._fin = weakref.finalize(self, lib.stswin_tn_fused_hold, -1)

    def release(self):
        self._fin()            # (idempotent: a finalizer runs once)


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise StswinHipError(f"{what} failed with code {rc}")


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return 0
    if t.dtype == torch.float32:
        return 1
    raise StswinHipError(f"unsupported dtype {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return _c_void_p(0)
    if not t.is_cuda:
        raise StswinHipError("stswincl_amd ops need tensors on the GPU (no CPU path exists)")
    return _c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    # the raw handle of the calling thread's current stream on the current device: torch.cuda.current_stream() builds a Stream
    # object per call (9 us, ~460 calls per training step = 4 ms of host time); the private accessor is 20x cheaper
    if _RAW_STREAM is not None:
        return _c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return _c_void_p(torch.cuda.current_stream().cuda_stream)


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D view"
    return t.stride(0)


# ----------------------------------------------------------------------------------------------- zero arena
# Accumulators that the kernels add into (BatchNorm / LayerNorm statistic sums, bias gradients, CE counters) must start at
# zero; a fill launch per buffer costs ~4.5 us of GPU time and the training step needs ~80 of them.  zeros() hands out
# 64-byte aligned slices of one zero-filled block per device instead: ONE fill per block (4 M floats), a fresh block when
# the current one is used up or when arena_reset() is called (model forward: once per step).  A slice is handed out
# once and never recycled - the block is freed when its last slice dies - so nothing can observe stale contents.
_ARENA_FLOATS = 1 << 22
_ARENA_MAX_REQUEST = 0 if os.environ.get("STSWIN_NO_ARENA") == "1" else 1 << 18      # (switch for A/B runs)
_ARENAS = {}


def arena_reset(device=None) -> None:
    """Drop the current block(s): the next zeros() starts a fresh one (called at the start of a model forward so that a
    step's forward and backward share one fill and a long-lived slice does not pin a mostly unused block for ever)."""
    if device is None:
        _ARENAS.clear()
    else:
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:      # ("cuda" means the current device, as in zeros())
            device = torch.device("cuda", torch.cuda.current_device())
        _ARENAS.pop(device, None)


def zeros(*shape, device) -> torch.Tensor:
    """fp32 zeros of the given shape: a slice of the device's zero arena (small requests) or torch.zeros (large ones)."""
    if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
        shape = tuple(shape[0])
    n = 1
    for d in shape:
        n *= int(d)
    if n > _ARENA_MAX_REQUEST or n == 0:
        return torch.zeros(shape, dtype=torch.float32, device=device)
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    st = _ARENAS.get(device)
    need = (n + 15) // 16 * 16
    if st is None or st[1] + need > _ARENA_FLOATS:
        st = [torch.zeros(_ARENA_FLOATS, dtype=torch.float32, device=device), 0]
        _ARENAS[device] = st
    out = st[0][st[1]:st[1] + n].view(shape)
    st[1] += need
    return out


# ----------------------------------------------------------------------------------------------- scratch for deterministic sums
# Kernels that sum over workgroups store per-workgroup partial vectors into caller-owned scratch and a fold kernel adds them in a
# fixed order (include/stswin_hip.h, "deterministic cross-workgroup sums").  One buffer per device: every use is write-then-read on
# the launch stream and kernels of a stream run in order.  It only ever grows (warm-up steps size it before a hipGraph capture).
_SCRATCH = {}
_SCRATCH_RETIRED = []        # superseded (smaller) scratch blocks: see scratch()
_DEFER = [False, 0]          # [folds are being queued (deferred_folds), bump offset into the scratch buffer]
_CAPTURED = [False]          # a hipGraph capture has happened in this process (note_capture)


def note_capture() -> None:
    """Tell the scratch allocator that a hipGraph has been (or is being) captured: from now on superseded scratch blocks are retired,
    not freed (a captured kernel node keeps the address it was captured with).  scratch() also sets it when it sees a capture."""
    _CAPTURED[0] = True


def scratch(device, floats: int) -> torch.Tensor:
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    floats = (int(floats) + 63) // 64 * 64
    t = _SCRATCH.get(device)
    off = _DEFER[1] if _DEFER[0] else 0
    if not _CAPTURED[0] and device.type == "cuda" and torch.cuda.is_current_stream_capturing():
        _CAPTURED[0] = True
    if t is None or t.numel() < off + floats:
        if _DEFER[0] and off > 0:            # queued folds still read the regions handed out so far: run them, then start over
            _check(load().stswin_fold_flush(_stream()), "fold_flush")
            _DEFER[1] = off = 0
        if t is None or t.numel() < floats:
            if t is not None and torch.cuda.is_current_stream_capturing():
                raise StswinHipError("scratch buffer would have to grow during a hipGraph capture: run the step once before capturing")
            if t is not None and _CAPTURED[0]:
                # A hipGraph captured earlier may still write and read the old block on every replay (its address is baked into the
                # graph's kernel arguments): superseded blocks are kept alive, never handed back to the caching allocator - but only
                # once a capture has happened in this process (note_capture(), called by whoever captures: bench.py, the tests);
                # before that nothing can refer to the old block and it is simply freed (round-4 advisor)
                _SCRATCH_RETIRED.append(t)
            # geometric growth: a run of increasing requests re-allocates O(log) times and retains at most ~2x the final size
            t = torch.empty(max(floats, 2 * (t.numel() if t is not None else 0), 1 << 23), dtype=torch.float32, device=device)
            _SCRATCH[device] = t
    if _DEFER[0]:
        _DEFER[1] = off + floats
        return t[off:off + floats]
    return t


class deferred_folds:
    """with deferred_folds() as d: the fixed-order folds of the kernels called inside (LayerNorm / bias-gradient / attention partial
    sums) are queued and launched together by d.flush() or at the end of the block - legal when none of their results is read
    in between (a Swin block's backward: they are all parameter gradients).  Every call gets its own region of the scratch buffer."""

    def __enter__(self):
        _check(load().stswin_fold_defer(1, _stream()), "fold_defer")
        _DEFER[0], _DEFER[1] = True, 0
        return self

    def flush(self):
        _check(load().stswin_fold_flush(_stream()), "fold_flush")
        _DEFER[1] = 0

    def abort(self):
        """Leave the deferred mode after an error inside the block (what was queued is still launched: its scratch is intact)."""
        _DEFER[0], _DEFER[1] = False, 0
        load().stswin_fold_defer(0, _stream())

    def __exit__(self, *exc):
        _DEFER[0], _DEFER[1] = False, 0
        _check(load().stswin_fold_defer(0, _stream()), "fold_defer")
        return False


# ----------------------------------------------------------------------------------------------- live profiling
# bench.py brackets every launch of the dominant kernels with HIP events recorded on the launch stream (torch's
# current stream IS the stream the kernels are launched on) and reads them back after the timed region.
# Two event records per launch cost ~1.7 ms of host time per training step (480 spans), which would make the timed
# region launch-bound; `stride` > 1 brackets one launch in `stride`.  WHICH ones rotates with the step (profile_step(k), called by
# the caller before step k): launch number n of a kernel family inside step k is timed iff (n + k) % stride == 0, so over `stride`
# steps every launch site of the step is timed exactly once and the average is the exact per-site average - not a sample whose
# composition changes whenever a launch is added or removed elsewhere (a seeded random choice moved the reported gemm_nt
# fraction by +-0.01 between trees with identical gemm_nt launches).  Without profile_step the choice is the seeded random one.
_PROFILE = None
_PROFILE_STRIDE = 1
_PROFILE_COUNT = {}
_PROFILE_STEP_COUNT = None            # per-step launch numbers (rotating mode) or None (random mode)
_PROFILE_OFFSET = 0
_PROFILE_RNG = None
_PROFILE_ALWAYS = ("contrast_",)      # kernels launched once per step: every launch is timed


def profile_begin(stride: int = 1):
    global _PROFILE, _PROFILE_STRIDE, _PROFILE_COUNT, _PROFILE_RNG, _PROFILE_STEP_COUNT, _PROFILE_OFFSET
    import random
    _PROFILE, _PROFILE_STRIDE, _PROFILE_COUNT, _PROFILE_RNG = {}, max(1, int(stride)), {}, random.Random(12345)
    _PROFILE_STEP_COUNT, _PROFILE_OFFSET = None, 0


def profile_step(k: int):
    """Start of timed step k: switch to (or stay in) the rotating choice of timed launches."""
    global _PROFILE_STEP_COUNT, _PROFILE_OFFSET
    if _PROFILE is not None:
        _PROFILE_STEP_COUNT, _PROFILE_OFFSET = {}, int(k) % _PROFILE_STRIDE


def profile_end():
    """-> {kernel: dict(launches, sampled, ms_total, ms_avg, work)}: ms_total / work cover the `sampled` launches that
    were bracketed by events (work = their algorithmic flops or bytes), `launches` counts every launch."""
    global _PROFILE
    prof, _PROFILE = _PROFILE, None
    torch.cuda.synchronize()
    out = {}
    for name, recs in (prof or {}).items():
        ms = sum(a.elapsed_time(b) for a, b, _ in recs)
        out[name] = dict(launches=_PROFILE_COUNT.get(name, len(recs)), sampled=len(recs), ms_total=ms,
                         ms_avg=ms / max(len(recs), 1), work=sum(w for _, _, w in recs))
    return out


_SHAPE_NAMES = os.environ.get("STSWIN_SHAPE_PROFILE", "0") == "1"   # tools/shape_profile.py: one span name per GEMM shape


class _Span:
    def __init__(self, name, work):
        self.name, self.work = name, work

    def __enter__(self):
        self.a = None
        if _PROFILE is not None:
            n = _PROFILE_COUNT.get(self.name, 0)
            _PROFILE_COUNT[self.name] = n + 1
            if _PROFILE_STEP_COUNT is not None:
                m = _PROFILE_STEP_COUNT.get(self.name, 0)
                _PROFILE_STEP_COUNT[self.name] = m + 1
                pick = (m + _PROFILE_OFFSET) % _PROFILE_STRIDE == 0
            else:
                pick = _PROFILE_RNG.random() * _PROFILE_STRIDE < 1.0
            if _PROFILE_STRIDE == 1 or self.name.startswith(_PROFILE_ALWAYS) or pick:
                self.a = torch.cuda.Event(enable_timing=True)
                self.a.record()
        return self

    def cancel(self):
        """Nothing was launched inside the span (a declined grouped launch): no sample, and the launch / per-step site numbers
        __enter__ handed out go back, so that the family's launch count and the rotation of timed sites are those of the launches
        that did happen (the round-5 line counted a declined group as a launch and credited its work to the next sample)."""
        if _PROFILE is not None and not getattr(self, "_cancelled", False):
            self._cancelled = True
            _PROFILE_COUNT[self.name] = _PROFILE_COUNT.get(self.name, 1) - 1
            if _PROFILE_STEP_COUNT is not None and self.name in _PROFILE_STEP_COUNT:
                _PROFILE_STEP_COUNT[self.name] -= 1
        self.a = None

    def __exit__(self, *exc):
        if self.a is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _PROFILE.setdefault(self.name, []).append((self.a, b, self.work))
        return False


# ----------------------------------------------------------------------------------------------- row maps
def win_rowmap(B, T, H, W, ws, shift, f0=0, frames_total=None, device="cuda") -> torch.Tensor:
    frames_total = T if frames_total is None else frames_total
    m = torch.empty(B * T * H * W, dtype=torch.int32, device=device)
    _check(load().stswin_win_rowmap(_p(m), B, T, H, W, ws, shift, f0, frames_total, _stream()), "win_rowmap")
    return m


def win_move(x: torch.Tensor, B, T, H, W, ws, shift, direction: int, f0=0, frames_total=None) -> torch.Tensor:
    """direction 0: (B,Ftot,H*W,C) tokens -> (B*nW*T*N, C) window rows; 1: the inverse (into zeros)."""
    frames_total = T if frames_total is None else frames_total
    C = x.shape[-1]
    x = x.contiguous()
    if direction == 0:
        out = torch.empty(B * T * H * W, C, dtype=x.dtype, device=x.device)
    else:
        out = torch.zeros(B * frames_total * H * W, C, dtype=x.dtype, device=x.device)
    _check(load().stswin_win_move(_dt(x), _p(x), _p(out), B, T, H, W, C, ws, shift, f0, frames_total, direction,
                                  _stream()), "win_move")
    return out


def merge_rowmap(frames, H, W, device="cuda") -> torch.Tensor:
    m = torch.empty(4, frames * (H // 2) * (W // 2), dtype=torch.int32, device=device)
    _check(load().stswin_merge_rowmap(_p(m), frames, H, W, _stream()), "merge_rowmap")
    return m


def conv3x3_rowmap(frames, H, W, dilation, device="cuda") -> torch.Tensor:
    m = torch.empty(9, frames * H * W, dtype=torch.int32, device=device)
    _check(load().stswin_conv3x3_rowmap(_p(m), frames, H, W, dilation, _stream()), "conv3x3_rowmap")
    return m


def conv_rowmap(frames, Hin, Win, Hout, Wout, k, stride, pad, dil, inverse, device="cuda") -> torch.Tensor:
    n = frames * (Hin * Win if inverse else Hout * Wout)
    m = torch.empty(k * k, n, dtype=torch.int32, device=device)
    _check(load().stswin_conv_rowmap(_p(m), frames, Hin, Win, Hout, Wout, k, stride, pad, dil, 1 if inverse else 0,
                                     _stream()), "conv_rowmap")
    return m


def conv_pack(w: torch.Tensor, dt: torch.dtype, omap: torch.Tensor, imap: torch.Tensor, want_dgrad: bool = True):
    """(Cout,Cin,k,k) fp32 -> (fwd [cop][S*cip], dgrad [cip][S*cop] or None) in dtype dt; see include/stswin_hip.h."""
    co, ci, k, _ = w.shape
    S, cop, cip = k * k, omap.numel(), imap.numel()
    wf = w.detach().float().contiguous()
    fwd = torch.empty(cop, S * cip, dtype=dt, device=w.device)
    dg = torch.empty(cip, S * cop, dtype=dt, device=w.device) if want_dgrad else None
    _check(load().stswin_conv_pack(0 if dt == torch.bfloat16 else 1, _p(wf), _p(fwd), _p(dg), _p(omap), _p(imap), co, ci, S,
                                   cop, cip, _stream()), "conv_pack")
    return fwd, dg


def linear_pack(w: torch.Tensor, dt: torch.dtype, want_tr: bool = True):
    """nn.Linear weight [n][k] fp32 -> (W, W^T or None) in dtype dt, one launch."""
    n, k = w.shape
    wf = w.detach().float().contiguous()
    fwd = torch.empty(n, k, dtype=dt, device=w.device)
    tr = torch.empty(k, n, dtype=dt, device=w.device) if want_tr else None
    _check(load().stswin_linear_pack(0 if dt == torch.bfloat16 else 1, _p(wf), _p(fwd), _p(tr), n, k, _stream()), "linear_pack")
    return fwd, tr


def linear_pack_multi(entries, dt: torch.dtype):
    """entries: [(w fp32 [n][k], fwd [n][k], tr [k][n] or None)] -> re-pack all of them in place, 64 weights per launch."""
    lib, st = load(), _stream()
    for lo in range(0, len(entries), 64):
        ch = entries[lo:lo + 64]
        c = len(ch)
        ws = (_c_void_p * c)(*[e[0].data_ptr() for e in ch])
        fs = (_c_void_p * c)(*[e[1].data_ptr() for e in ch])
        ts = (_c_void_p * c)(*[(e[2].data_ptr() if e[2] is not None else 0) for e in ch])
        ns = (_c_int * c)(*[e[0].shape[0] for e in ch])
        ks = (_c_int * c)(*[e[0].shape[1] for e in ch])
        _check(lib.stswin_linear_pack_multi(0 if dt == torch.bfloat16 else 1, c, ws, fs, ts, ns, ks, st), "linear_pack_multi")


def conv_pack_multi(entries, dt: torch.dtype):
    """entries: [(w fp32 (co,ci,k,k), fwd, dgrad or None, omap, imap)] -> re-pack in place, 32 weights per launch."""
    lib, st = load(), _stream()
    for lo in range(0, len(entries), 32):
        ch = entries[lo:lo + 32]
        c = len(ch)
        arr = lambda i: (_c_void_p * c)(*[(e[i].data_ptr() if e[i] is not None else 0) for e in ch])  # noqa: E731
        ints = lambda f: (_c_int * c)(*[f(e) for e in ch])  # noqa: E731
        _check(lib.stswin_conv_pack_multi(0 if dt == torch.bfloat16 else 1, c, arr(0), arr(1), arr(2), arr(3), arr(4),
                                          ints(lambda e: e[0].shape[1]), ints(lambda e: e[0].shape[2] * e[0].shape[3]),
                                          ints(lambda e: e[3].numel()), ints(lambda e: e[4].numel()), st), "conv_pack_multi")


def stem_im2col(img: torch.Tensor, dtype: torch.dtype, Ho: int, Wo: int, ld: int = 192) -> torch.Tensor:
    """img fp32 NCHW [F][3][H][W] -> patches [F*Ho*Wo][ld] (7x7 / stride 2 / pad 3)."""
    F_, c, H, W = img.shape
    assert c == 3 and img.dtype == torch.float32 and img.is_contiguous()
    out = torch.empty(F_ * Ho * Wo, ld, dtype=dtype, device=img.device)
    _check(load().stswin_stem_im2col(_dt(out), _p(img), _p(out), _c_long(ld), F_, H, W, Ho, Wo, _stream()), "stem_im2col")
    return out


def stem_s2d(img: torch.Tensor, dtype: torch.dtype):
    """img fp32 NCHW [F][3][H][W] -> (A, Hs, Ws): A = the [F*Hs*Ws][64] row view (row pitch 16: every row is the 4-record segment
    starting at its record) of the padded 2 x 2 space-to-depth image; see include/stswin_hip.h."""
    F_, c, H, W = img.shape
    assert c == 3 and img.dtype == torch.float32 and img.is_contiguous()
    Hs, Ws = (H - 1) // 2 + 4, (W - 1) // 2 + 4
    n = F_ * Hs * Ws
    flat = torch.empty(n * 16 + 64, dtype=dtype, device=img.device)
    flat[n * 16:].zero_()
    _check(load().stswin_stem_s2d(_dt(flat), _p(img), _p(flat), F_, H, W, _stream()), "stem_s2d")
    return torch.as_strided(flat, (n, 64), (16, 1)), Hs, Ws


def stem_conv(A, wmat, y, frames, H, W, stats_out=None):
    """y [F*Ho*Wo][64] = the stem convolution of the hip.stem_s2d view A with the packed weights wmat [64][256] (bf16; Wo % 128 == 0)."""
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    M = frames * Ho * Wo
    assert A.dtype == torch.bfloat16 and A.stride() == (16, 1) and A.shape[0] == frames * (Ho + 3) * (Wo + 3)
    assert wmat.dtype == torch.bfloat16 and wmat.shape == (64, 256) and wmat.is_contiguous()
    assert y.dtype == torch.bfloat16 and y.shape == (M, 64) and y.is_contiguous()
    assert stats_out is None or (stats_out.dtype == torch.float32 and stats_out.numel() == 2 * 2 * ((M + 255) // 256) * 64)
    with _Span("stem_conv_bf16", 2.0 * M * 64 * 147):
        rc = load().stswin_stem_conv(_p(A), _p(wmat), _p(y), _p(stats_out), frames, H, W, _stream())
    _check(rc, "stem_conv")
    return y


def stem_wgrad_ok(H, W, dtype) -> bool:
    return dtype == torch.bfloat16 and ((W - 1) // 2 + 1) % 128 == 0


def stem_wgrad(dy, A, dw, frames, H, W, accumulate=False):
    """dw fp32 [64][256] (= [cout][tap row][record][16]) from dy [F*Ho*Wo][64] and the hip.stem_s2d view A of the same images."""
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    assert dy.dtype == torch.bfloat16 and dy.shape == (frames * Ho * Wo, 64) and dy.is_contiguous()
    assert A.dtype == torch.bfloat16 and A.stride() == (16, 1) and A.shape[0] == frames * (Ho + 3) * (Wo + 3)
    assert dw.dtype == torch.float32 and dw.numel() == 64 * 256 and dw.is_contiguous()
    ws = scratch(dy.device, load().stswin_stem_wgrad_scratch(frames, Ho, Wo))
    with _Span("stem_wgrad_bf16", 2.0 * frames * Ho * Wo * 64 * 147):
        rc = load().stswin_stem_wgrad(_p(dy), _p(A), _p(dw), 1 if accumulate else 0, _p(ws), _c_long(ws.numel()), frames, H, W, _stream())
    _check(rc, "stem_wgrad")
    return dw


def maxpool3x3s2(src, dst, arg, frames, H, W, Ho, Wo, backward=False):
    C = src.shape[1]
    _check(load().stswin_maxpool3x3s2(_dt(src), _p(src), _c_long(_ld(src)), _p(dst), _c_long(_ld(dst)), _p(arg), frames, H,
                                      W, Ho, Wo, C, 1 if backward else 0, _stream()), "maxpool3x3s2")
    return dst


def conv3x3_c64_ok(frames, H, W, cin, cout, k, stride, pad, dil, dt) -> bool:
    """Geometries stswin_conv3x3_c64 takes (everything else stays on the gather GEMM)."""
    return (dt == torch.bfloat16 and cin == 64 and cout == 64 and k == 3 and stride == 1 and pad == 1 and dil == 1
            and W in (16, 32, 64, 128) and (H * W) % 256 == 0 and frames > 0)


def conv3x3_c64(x, wmat, y, frames, H, W, sign=1, resid=None, stats_out=None):
    """y = conv3x3(x) (sign = +1, wmat = forward matrix) or its input gradient (sign = -1, x = dy, wmat = dgrad matrix)."""
    M = frames * H * W
    for t in (x, y, resid):
        assert t is None or (t.dtype == torch.bfloat16 and t.shape == (M, 64) and t.is_contiguous())
    assert wmat.dtype == torch.bfloat16 and wmat.shape == (64, 576) and wmat.is_contiguous()
    assert stats_out is None or (stats_out.dtype == torch.float32 and stats_out.numel() == 2 * 2 * ((M + 255) // 256) * 64)
    with _Span("conv3x3_c64_bf16", 2.0 * M * 64 * 576):
        rc = load().stswin_conv3x3_c64(_p(x), _p(wmat), _p(y), _p(resid), _p(stats_out), frames, H, W, sign, _stream())
    _check(rc, "conv3x3_c64")
    return y


def conv3x3_c64_wgrad_ok(frames, H, W) -> bool:
    return W in (32, 64, 128) and (H * W) % 128 == 0 and frames > 0


def conv3x3_c64_wgrad(dy, x, dw, frames, H, W, tapminor=True, accumulate=False):
    """dw (fp32, 64*576 values: [64][64][3][3] if tapminor else [64][9][64]) = weight gradient of the 64 -> 64 3x3 convolution."""
    M = frames * H * W
    for t in (dy, x):
        assert t.dtype == torch.bfloat16 and t.shape == (M, 64) and t.is_contiguous()
    assert dw.dtype == torch.float32 and dw.numel() == 64 * 576 and dw.is_contiguous()
    need = load().stswin_conv3x3_c64_wgrad_scratch(frames, H, W)
    ws = scratch(x.device, need)
    with _Span("conv3x3_c64_wgrad_bf16", 2.0 * M * 64 * 576):
        rc = load().stswin_conv3x3_c64_wgrad(_p(dy), _p(x), _p(dw), 1 if tapminor else 0, 1 if accumulate else 0, _p(ws), _c_long(ws.numel()),
                                             frames, H, W, _stream())
    _check(rc, "conv3x3_c64_wgrad")
    return dw


# ----------------------------------------------------------------------------------------------- GEMMs
def stats_table(M: int, N: int, device) -> torch.Tensor:
    """fp32 [2][2*ceil(M/256)][N] table for gemm_nt(stats_out=...): per-128-row-block column sums | sums of squares."""
    return torch.empty(2, 2 * ((M + 255) // 256), N, dtype=torch.float32, device=device)


def cs_group_reduce(table: torch.Tensor, M: int, groups: int, unit: int = 0):
    """-> (sum, sumsq) fp32 [groups][N] of the statistic groups from a stats_table a gemm_nt filled."""
    N = table.shape[2]
    both = torch.empty(2, groups, N, dtype=torch.float32, device=table.device)
    _check(load().stswin_cs_group_reduce(_p(table), M, N, groups, unit, _p(both[0]), _p(both[1]), _stream()), "cs_group_reduce")
    return both[0], both[1]


def bn_table_finalize(table: torch.Tensor, M: int, running_mean, running_var, groups=1, eps=1e-5, momentum=0.1, unit=0):
    """-> (mean, rstd) fp32 [groups][N] from a stats_table a gemm_nt filled; running statistics updated in place."""
    N = table.shape[2]
    mean = torch.empty(groups, N, dtype=torch.float32, device=table.device)
    rstd = torch.empty_like(mean)
    _check(load().stswin_bn_table_finalize(_p(table), M, N, groups, unit, _p(mean), _p(rstd), _p(running_mean), _p(running_var),
                                           _c_float(eps), _c_float(momentum), _stream()), "bn_table_finalize")
    return mean, rstd


def gemm_nt(A: torch.Tensor, Bw: torch.Tensor, out: torch.Tensor, *, M: int, a_rows=None, c_rows=None, bias=None,
            resid=None, r_rows=None, out2=None, S: int = 1, scale: float = 1.0, scale_cols: int = 0, flags: int = 0,
            colsum_out=None, stats_out=None):
    """out[c_rows[m]] = epi(sum_s A[a_rows[s][m], :Kseg] @ Bw[:, s*Kseg:(s+1)*Kseg].T); see include/stswin_hip.h."""
    N, Ktot = Bw.shape
    assert Ktot % S == 0
    Kseg = Ktot // S
    assert A.dtype == Bw.dtype and A.shape[1] >= Kseg
    if not (flags & GF_OUT_F32):
        assert out.dtype == A.dtype
    else:
        assert out.dtype == torch.float32
    cs_table = None
    if stats_out is not None:        # BatchNorm statistics of the output: per-block sums and sums of squares, no reduce here
        assert colsum_out is None and stats_out.shape == (2, 2 * ((M + 255) // 256), N) and stats_out.is_contiguous()
        flags |= GF_CS_PARTIAL | GF_CS_SQ
        colsum_out = stats_out
    elif colsum_out is not None and M >= _CS_PARTIAL_MIN_M and N % 4 == 0 and not (flags & (1 << 19)):
        # >= 64 row tiles would each add into the same N addresses: per-block partial sums + one small reduce instead
        cs_table = scratch(A.device, 2 * ((M + 255) // 256) * N)
        flags |= GF_CS_PARTIAL
    elif colsum_out is not None and N % 4 and not _WARNED.get("cs_atomic"):
        _WARNED["cs_atomic"] = True
        import warnings
        warnings.warn(f"stswincl_amd.hip.gemm_nt: column sums of an output with N = {N} (not a multiple of 4) are added with fp32 "
                      f"atomics (order-dependent last bits); pad N to a multiple of 4 for bitwise-reproducible sums")
    name = "gemm_nt_bf16" if A.dtype == torch.bfloat16 else "gemm_nt_f32"
    if _SHAPE_NAMES:
        name += f" M={M} N={N} K={Kseg} S={S} a={int(a_rows is not None)} c={int(c_rows is not None)} fl={flags}"
    # few output tiles, long K (ASPP's dilated convolutions, the 448-channel classifier convolution): ring kernel over tiles x K-splits
    if (_NT_SPLITK and A.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and c_rows is None and resid is None and out2 is None
            and colsum_out is None and scale == 1.0 and scale_cols == 0 and not (flags & ~GF_RELU)):
        need = load().stswin_gemm_nt_splitk_scratch(M, N, Kseg, S)
        if need > 0:
            ws = scratch(A.device, need)
            with _Span(name, 2.0 * M * N * Ktot):
                rc = load().stswin_gemm_nt_splitk(_p(A), _c_long(_ld(A)), _p(a_rows), _p(Bw), _c_long(_ld(Bw)), _p(out), _c_long(_ld(out)),
                                                  _p(bias), M, N, Kseg, S, 1 if (flags & GF_RELU) else 0, _p(ws), _c_long(ws.numel()), _stream())
            if rc != -1008:                  # -1008: not a split-K candidate after all (an `out` column slice that is not 16-byte
                _check(rc, "gemm_nt_splitk")  # aligned, a row pitch that is not a multiple of 8): the tiled kernels below take it
                _log_variant("nt", (M, N, Kseg, S), 0)
                return out
    with _Span(name, 2.0 * M * N * Ktot):
        rc = load().stswin_gemm_nt(
            _dt(A), _p(A), _c_long(_ld(A)), _p(a_rows), _p(Bw), _c_long(_ld(Bw)), _p(out), _c_long(_ld(out)),
            _p(c_rows), _p(out2), _c_long(_ld(out2) if out2 is not None else 0), _p(bias), _p(resid),
            _c_long(_ld(resid) if resid is not None else 0), _p(r_rows), M, N, Kseg, S, _c_float(scale), scale_cols,
            flags, _p(cs_table if cs_table is not None else colsum_out), _stream())
    _check(rc, "gemm_nt")
    _log_variant("nt", (M, N, Kseg, S), 0)
    if cs_table is not None:
        _check(load().stswin_cs_reduce(_p(cs_table), M, N, _p(colsum_out), _stream()), "cs_reduce")
    return out


def gemm_nt_qkv_fp8(A: torch.Tensor, Bw: torch.Tensor, *, M: int, a_rows=None, bias=None, scale: float = 1.0, scale_cols: int = 0,
                    rows_per_problem: int, head_dim: int):
    """QKV projection with an e4m3 result: returns (out8 uint8 [M][N], scales fp32 [M / rows_per_problem][N / head_dim]); see
    include/stswin_hip.h.  Raises StswinHipError on shapes the kernel does not take (the caller keeps the bf16 path for those)."""
    N, K = Bw.shape
    assert A.dtype == torch.bfloat16 and Bw.dtype == torch.bfloat16 and A.shape[1] >= K
    out8 = torch.empty(M, N, dtype=torch.uint8, device=A.device)
    scales = torch.empty(M // rows_per_problem, N // head_dim, dtype=torch.float32, device=A.device)
    with _Span("gemm_nt_bf16" + (f" M={M} N={N} K={K} S=1 a={int(a_rows is not None)} c=0 fl=fp8out" if _SHAPE_NAMES else ""), 2.0 * M * N * K):
        rc = load().stswin_gemm_nt_qkv_fp8(_p(A), _c_long(_ld(A)), _p(a_rows), _p(Bw), _c_long(_ld(Bw)), _p(out8), _c_long(N), _p(scales),
                                           _c_long(N // head_dim), _p(bias), M, N, K, _c_float(scale), scale_cols, rows_per_problem, head_dim,
                                           _stream())
    _check(rc, "gemm_nt_qkv_fp8")
    return out8, scales


VARIANT_LOG = None     # tests: set to a list -> every gemm_nt / gemm_tn / gemm_tn_group launch appends (family, shape, stswin_last_variant)


def _log_variant(family: str, shape, fam_id: int) -> None:
    if VARIANT_LOG is not None:
        VARIANT_LOG.append((family, tuple(int(v) for v in shape), int(load().stswin_last_variant(fam_id))))


_WARNED = {}
_NT_SPLITK = os.environ.get("STSWIN_NO_NT_SPLITK") != "1"                 # (A/B switch)
_CS_PARTIAL_MIN_M = int(os.environ.get("STSWIN_CS_PARTIAL_MIN_M", "1"))   # (the table + fold path is the deterministic one: always)
_CS_TABLES = {}


def _cs_table(device, floats):
    """Caller-owned scratch of the STSWIN_GF_CS_PARTIAL column-sum table (one per device, grown on demand; every use is
    write-then-read on the launch stream)."""
    t = _CS_TABLES.get(device)
    if t is None or t.numel() < floats:
        t = torch.empty(max(floats, 1 << 20), dtype=torch.float32, device=device)
        _CS_TABLES[device] = t
    return t


_TN_WS = {}


def _tn_workspace(device, floats=48 * 1024 * 1024):
    """One scratch buffer per device for the split-K partial slabs (kernels on a stream serialise, so it can be shared)."""
    ws = _TN_WS.get(device)
    if ws is None:
        ws = torch.empty(floats, dtype=torch.float32, device=device)
        _TN_WS[device] = ws
    return ws


TN_NO_COMBINE = 1 << 26
_TN_SIDE = {}            # device -> side stream of the deferred combines
_TN_DEFER = 0            # nesting depth of tn_deferred()
_TN_PENDING = None       # event of the last combine launched on the side stream (one at a time: the partials share one workspace)
# OFF by default: measured 0.8-1.0 ms per step SLOWER than the inline combine (536-539 vs 552-556 frames/s, eager and under hipGraph
# replay alike): every cross-stream event edge costs more than the 8 us kernel it hides.  STSWIN_TN_DEFERRED_COMBINE=1 enables it.
_TN_DEFER_ON = os.environ.get("STSWIN_TN_DEFERRED_COMBINE") == "1"


class tn_deferred:
    """with tn_deferred(): the split-K combine of every gemm_tn inside runs on a side stream, ordered behind its GEMM by an event,
    and the calling stream waits for the last one when the block ends - so each combine (5-12 us of a kernel too small to fill
    the chip, plus a launch boundary on either side) overlaps the input-gradient GEMM that follows every weight-gradient GEMM of a
    backward pass.  The next gemm_tn waits for the previous combine before it overwrites the shared partials."""

    def __enter__(self):
        global _TN_DEFER
        _TN_DEFER += 1
        return self

    def __exit__(self, *exc):
        global _TN_DEFER
        _TN_DEFER -= 1
        if _TN_DEFER == 0:
            tn_join()
        return False


def tn_deferred_backward(fn):
    """Decorator for autograd backward functions that issue weight-gradient GEMMs: body inside tn_deferred()."""
    import functools

    @functools.wraps(fn)
    def wrapper(ctx, *grads):
        with tn_deferred():
            return fn(ctx, *grads)
    return wrapper


def tn_join():
    """The current stream waits for the outstanding deferred combine (if any)."""
    global _TN_PENDING
    if _TN_PENDING is not None:
        torch.cuda.current_stream().wait_event(_TN_PENDING)
        _TN_PENDING = None


TN_OUT_TAPMINOR, VAR_TN_TAPMINOR = 1 << 25, 0x4000


def gemm_tn(At: torch.Tensor, Bt: torch.Tensor, out_f32: torch.Tensor, *, Mk: int, at_rows=None, bt_rows=None,
            splits: int = 0, bseg: int = 0, atomics: bool = False, overwrite: bool = False, debug_ts: bool = False, tapminor: bool = False):
    """out_f32[i][j] += sum_m At[at_rows[m]][i] * Bt[bt_rows[m]][j]  (fp32); overwrite=True stores instead of adding, so
    out_f32 may come from torch.empty."""
    if overwrite:
        splits |= TN_OVERWRITE
    if tapminor and bseg > 0:            # ask the combine for the [row][channel][tap] order; last_tn_tapminor() says whether it was honoured
        splits |= TN_OUT_TAPMINOR
    Ni, Nj = out_f32.shape
    assert out_f32.dtype == torch.float32 and At.dtype == Bt.dtype
    ws = None if atomics or torch.cuda.is_current_stream_capturing() and At.device not in _TN_WS else _tn_workspace(At.device)
    name = "gemm_tn_bf16" if At.dtype == torch.bfloat16 else "gemm_tn_f32"
    if _SHAPE_NAMES:
        name += f" Mk={Mk} Ni={Ni} Nj={Nj} bseg={bseg} a={int(at_rows is not None)} b={int(bt_rows is not None)}"
    defer = _TN_DEFER > 0 and _TN_DEFER_ON and ws is not None and not debug_ts
    tn_join()                                            # the previous combine still reads the workspace this launch overwrites
    with _Span(name, 2.0 * Mk * Ni * Nj):
        rc = load().stswin_gemm_tn(_dt(At), _p(At), _c_long(_ld(At)), _p(at_rows), _p(Bt), _c_long(_ld(Bt)),
                                   _p(bt_rows), _p(out_f32), _c_long((-1 if debug_ts is True else -int(debug_ts)) if debug_ts else _ld(out_f32)), Mk, Ni, Nj,
                                   (splits | TN_NO_COMBINE) if defer else splits, bseg, _p(ws),
                                   _c_long(ws.numel() if ws is not None else 0), _stream())
    _check(rc, "gemm_tn")
    _log_variant("tn", (Mk, Ni, Nj, bseg), 1)
    if defer:
        v = load().stswin_last_variant(1)
        if v & (VAR_TN_SLABS_F32 | VAR_TN_SLABS_BF16):   # partials are waiting in the workspace
            global _TN_PENDING
            cur = torch.cuda.current_stream()
            side = _TN_SIDE.get(At.device)
            if side is None:
                side = _TN_SIDE[At.device] = torch.cuda.Stream(device=At.device)
            ev = torch.cuda.Event()
            ev.record(cur)
            side.wait_event(ev)
            _check(load().stswin_tn_combine(_p(ws), _p(out_f32), _c_long(_ld(out_f32)), Ni, Nj, v >> 16, 1 if (splits & TN_OVERWRITE) else 0,
                                            1 if v & VAR_TN_SLABS_BF16 else 0, ctypes.c_void_p(side.cuda_stream)), "tn_combine")
            _TN_PENDING = torch.cuda.Event()
            _TN_PENDING.record(side)
    return out_f32


class _TnProblem(ctypes.Structure):
    """stswin_tn_problem (include/stswin_hip.h)"""
    _fields_ = [("At", ctypes.c_void_p), ("lda", ctypes.c_long), ("at_rows", ctypes.c_void_p),
                ("Bt", ctypes.c_void_p), ("ldb", ctypes.c_long), ("bt_rows", ctypes.c_void_p),
                ("C", ctypes.c_void_p), ("ldc", ctypes.c_long),
                ("Mk", ctypes.c_int), ("Ni", ctypes.c_int), ("Nj", ctypes.c_int), ("bseg", ctypes.c_int),
                ("overwrite", ctypes.c_int), ("tapminor", ctypes.c_int)]


TN_GROUP_DECLINED = -1050
LAST_TN_GROUP_SPLITS: list = []      # split counts of the last grouped launch (tests, tools)


def gemm_tn_group(problems) -> bool:
    """problems: [dict(At=, Bt=, out=, Mk=, at_rows=None, bt_rows=None, bseg=0, overwrite=True, tapminor=False)] - up to 4 bf16 weight
    gradients of one backward step in ONE launch (stswin_gemm_tn_group).  Returns False when the library declines the set (nothing has
    been written: call gemm_tn for each), True when it was launched."""
    n = len(problems)
    if n == 0:
        return True
    first = problems[0]["At"]
    if n > 4 or first.dtype != torch.bfloat16 or torch.cuda.is_current_stream_capturing() and first.device not in _TN_WS:
        return False
    arr = (_TnProblem * n)()
    flops = 0.0
    for i, q in enumerate(problems):
        At, Bt, out = q["At"], q["Bt"], q["out"]
        assert out.dtype == torch.float32 and At.dtype == Bt.dtype == torch.bfloat16
        ar, br = q.get("at_rows"), q.get("bt_rows")
        arr[i] = _TnProblem(At.data_ptr(), _ld(At), ar.data_ptr() if ar is not None else None, Bt.data_ptr(), _ld(Bt),
                            br.data_ptr() if br is not None else None, out.data_ptr(), _ld(out), q["Mk"], out.shape[0], out.shape[1],
                            q.get("bseg", 0), 1 if q.get("overwrite", True) else 0, 1 if q.get("tapminor", False) else 0)
        flops += 2.0 * q["Mk"] * out.shape[0] * out.shape[1]
    ws = _tn_workspace(first.device)
    tn_join()
    sp = (_c_int * n)()
    name = "gemm_tn_bf16"
    if _SHAPE_NAMES:
        name += " group " + " + ".join(f"{q['out'].shape[0]}x{q['out'].shape[1]}" for q in problems) + f" Mk={problems[0]['Mk']}"
    with _Span(name, flops) as span:
        rc = load().stswin_gemm_tn_group(0, n, arr, _p(ws), _c_long(ws.numel()), sp, _stream())
        if rc == TN_GROUP_DECLINED:
            span.cancel()
    if rc == TN_GROUP_DECLINED:
        return False
    _check(rc, "gemm_tn_group")
    LAST_TN_GROUP_SPLITS[:] = list(sp)
    _log_variant("tn_group", (n, problems[0]["Mk"]) + tuple(sp), 1)
    return True


# kernel-variant codes of stswin_last_variant (include/stswin_hip.h)
VAR_F32 = 100
(VAR_NT_RING256_REGEPI, VAR_NT_RING256_LDSEPI, VAR_NT_RING256_NOPIPE, VAR_NT_STREAM, VAR_NT_DUO, VAR_NT_RING256x128_PP,
 VAR_NT_MID, VAR_NT_256x64, VAR_NT_128x64, VAR_NT_128x128, VAR_NT_128x128_W4, VAR_NT_ROWS, VAR_NT_SPLITK) = range(1, 14)
VAR_TN_RING_PLAIN, VAR_TN_RING_ATROWS, VAR_TN_RING_BTROWS, VAR_TN_RING_BSEG = 20, 21, 22, 23
VAR_TN_128x128, VAR_TN_128x128_W4, VAR_TN_ROWS = 30, 31, 32
VAR_TN_SLABS_F32, VAR_TN_SLABS_BF16 = 0x1000, 0x2000
VAR_TN_FUSED = 0x8000        # the split-K partials were combined inside the GEMM launch


def last_tn_tapminor() -> bool:
    """Whether the most recent gemm_tn(tapminor=True) of this thread stored its result tap-minor (it does where split-K slabs are
    combined; a direct-store launch keeps the GEMM order and the caller permutes)."""
    return bool(load().stswin_last_variant(1) & VAR_TN_TAPMINOR)


def last_variant(family: int) -> dict:
    """Which kernel the most recent gemm_nt (family 0) / gemm_tn (family 1) call of this thread launched:
    {kernel, slabs ('' / 'f32' / 'bf16'), splits}.  Test instrumentation (asserts the production dispatch)."""
    v = load().stswin_last_variant(family)
    return {"kernel": v & 0xFFF, "slabs": "bf16" if v & VAR_TN_SLABS_BF16 else ("f32" if v & VAR_TN_SLABS_F32 else ""),
            "splits": v >> 16}


def vec_gather(v: torch.Tensor, imap: torch.Tensor, fill: float = 0.0) -> torch.Tensor:
    """out[i] = v[imap[i]] (fill where imap[i] < 0); v fp32 contiguous, imap int32."""
    out = torch.empty(imap.numel(), dtype=torch.float32, device=v.device)
    _check(load().stswin_vec_gather(_p(v), _p(imap), _p(out), imap.numel(), _c_float(fill), _stream()), "vec_gather")
    return out


def vec_gather_multi(vs, imap: torch.Tensor, fills=None, outs=None):
    """[out_k[i] = vs[k][imap[i]] (fills[k] where imap[i] < 0)] for up to 4 fp32 contiguous vectors sharing the map, ONE launch.
    outs: write into these tensors (e.g. the running-statistic buffers themselves) instead of fresh ones."""
    k, n = len(vs), imap.numel()
    assert 1 <= k <= 4 and all(v.dtype == torch.float32 and v.is_contiguous() for v in vs)
    if outs is None:
        flat = torch.empty(k, (n + 15) // 16 * 16, dtype=torch.float32, device=imap.device)
        outs = [flat[i, :n] for i in range(k)]
    else:
        assert len(outs) == k and all(o.dtype == torch.float32 and o.is_contiguous() and o.numel() == n for o in outs)
    fills = list(fills) if fills is not None else [0.0] * k
    _check(load().stswin_vec_gather_multi(k, (_c_void_p * k)(*[v.data_ptr() for v in vs]), _p(imap), (_c_void_p * k)(*[o.data_ptr() for o in outs]),
                                          n, (_c_float * k)(*[float(f) for f in fills]), _stream()), "vec_gather_multi")
    return outs


def colsum(y: torch.Tensor, out_f32: torch.Tensor, M: Optional[int] = None):
    M = y.shape[0] if M is None else M
    lib = load()
    ws = scratch(y.device, lib.stswin_colsum_scratch(M, y.shape[1]))
    _check(lib.stswin_colsum(_dt(y), _p(y), _c_long(_ld(y)), _p(out_f32), M, y.shape[1], _p(ws), _stream()), "colsum")
    return out_f32


# ----------------------------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x, gamma, beta, *, M, rows=None, S=1, Cseg=None, eps=1e-5, save_stats=True, out=None):
    Cseg = x.shape[1] if Cseg is None else Cseg
    if out is not None:                       # caller-provided destination (a row block of a larger buffer)
        assert out.shape == (M, S * Cseg) and out.dtype == x.dtype and out.stride(1) == 1
        y = out
    else:
        y = torch.empty(M, S * Cseg, dtype=x.dtype, device=x.device)
    mean = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    rc = load().stswin_layernorm_fwd(_dt(x), _p(x), _c_long(_ld(x)), _p(rows), S, Cseg, _p(y), _c_long(_ld(y)),
                                     _p(gamma), _p(beta), _p(mean), _p(rstd), M, _c_float(eps), _stream())
    _check(rc, "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, *, M, rows=None, S=1, Cseg=None, dx=None, accumulate=False,
                  dxsum=None, add=None):
    """add (optional): dx = add + LN'(dy) into a fresh (or the given) dx, `add` left intact."""
    Cseg = x.shape[1] if Cseg is None else Cseg
    if dx is None:
        dx = torch.empty_like(x)
        accumulate = False
    ws = scratch(x.device, load().stswin_layernorm_bwd_scratch(M, S * Cseg))
    if add is not None:
        assert not accumulate and add.dtype == x.dtype
        rc = load().stswin_layernorm_bwd_add(_dt(x), _p(dy), _c_long(_ld(dy)), _p(x), _c_long(_ld(x)), _p(rows), S, Cseg,
                                             _p(gamma), _p(mean), _p(rstd), _p(add), _c_long(_ld(add)), _p(dx), _c_long(_ld(dx)),
                                             _p(dgamma), _p(dbeta), M, _p(dxsum), _p(ws), _stream())
        _check(rc, "layernorm_bwd_add")
        return dx
    rc = load().stswin_layernorm_bwd(_dt(x), _p(dy), _c_long(_ld(dy)), _p(x), _c_long(_ld(x)), _p(rows), S, Cseg,
                                     _p(gamma), _p(mean), _p(rstd), _p(dx), _c_long(_ld(dx)), _p(dgamma), _p(dbeta), M,
                                     1 if accumulate else 0, _p(dxsum), _p(ws), _stream())
    _check(rc, "layernorm_bwd")
    return dx


# ----------------------------------------------------------------------------------------------- attention
def bias_expand(table, index, mask, N, heads):
    """relative_position_bias_table[index] as [heads][key][query] fp32, or [nW][heads][key][query] with the mask added."""
    nW = mask.shape[0] if mask is not None else 1
    out = torch.empty((nW, heads, N, N) if mask is not None else (heads, N, N), dtype=torch.float32, device=table.device)
    assert table.dtype == torch.float32 and table.is_contiguous() and index.dtype == torch.int64 and index.is_contiguous()
    assert mask is None or (mask.dtype == torch.float32 and mask.is_contiguous())
    _check(load().stswin_bias_expand(_p(table), _p(index), _p(mask), _p(out), N, heads, nW, _stream()), "bias_expand")
    return out


def bias_expand_multi(entries) -> None:
    """entries: [(table, index, mask or None, out, N, heads)] - bias_expand of every entry into its existing `out`, 16 per launch."""
    lib, st = load(), _stream()
    for lo in range(0, len(entries), 16):
        ch = entries[lo:lo + 16]
        k = len(ch)
        arr = lambda i: (_c_void_p * k)(*[(e[i].data_ptr() if e[i] is not None else 0) for e in ch])   # noqa: E731
        _check(lib.stswin_bias_expand_multi(k, arr(0), arr(1), arr(2), arr(3), (_c_int * k)(*[e[4] for e in ch]), (_c_int * k)(*[e[5] for e in ch]),
                                            (_c_int * k)(*[(e[2].shape[0] if e[2] is not None else 1) for e in ch]), st), "bias_expand_multi")


def scatter_lists(index: torch.Tensor, table_rows: int):
    """(order, offs) of stswin_bias_scatter for an index buffer: the pairs i * N + j sorted (stably) by their table row and the start of
    every row's range.  Three small torch kernels, no host sync; callers that scatter through the same index every step keep the
    result (ops.py caches it per window size: relative_position_index is a function of the window size alone)."""
    flat = index.reshape(-1)
    order = torch.argsort(flat, stable=True).to(torch.int32).contiguous()
    counts = torch.bincount(flat, minlength=table_rows)[:table_rows]
    offs = torch.zeros(table_rows + 1, dtype=torch.int32, device=index.device)
    offs[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return order, offs.contiguous()


def bias_scatter(dbiasT, index, dtable, N, heads, lists=None):
    assert dbiasT.is_contiguous() and dtable.is_contiguous() and index.dtype == torch.int64 and index.is_contiguous()
    assert dtable.shape[0] * dtable.shape[1] == dtable.numel() and dtable.shape[1] == heads
    order, offs = lists if lists is not None else scatter_lists(index, dtable.shape[0])
    _check(load().stswin_bias_scatter(_p(dbiasT), _p(order), _p(offs), _p(dtable), N, heads, dtable.shape[0], 1, _stream()), "bias_scatter")
    return dtable


def _bias_windows(biasT, maskT, nW, bias_index=None):
    """biasT [heads][N][N] (+ optional maskT) or the pre-summed per-window table [nW][heads][N][N] (maskT None)."""
    if biasT.dim() == 4:
        assert maskT is None and (biasT.shape[0] == nW or bias_index is not None)
        return biasT.shape[0]
    return 1


def win_attn_fwd(qkv, biasT, maskT, *, nB_, nW, T, ws, heads, C, bias_index=None, fp8=False):
    out = torch.empty(qkv.shape[0], C, dtype=qkv.dtype, device=qkv.device)
    if fp8:
        if qkv.dtype != torch.bfloat16:
            raise StswinHipError("fp8 attention quantises bf16 q / k / v")
        rc = load().stswin_win_attn_fwd_fp8(_p(qkv), _c_long(_ld(qkv)), _p(out), _c_long(_ld(out)), _p(biasT), _p(maskT), nB_, nW, T, ws,
                                            heads, C, _bias_windows(biasT, maskT, nW, bias_index), _p(bias_index), _stream())
        _check(rc, "win_attn_fwd_fp8")
        return out
    rc = load().stswin_win_attn_fwd(_dt(qkv), _p(qkv), _c_long(_ld(qkv)), _p(out), _c_long(_ld(out)), _p(biasT),
                                    _p(maskT), nB_, nW, T, ws, heads, C, _bias_windows(biasT, maskT, nW, bias_index), _p(bias_index),
                                    _stream())
    _check(rc, "win_attn_fwd")
    return out


def win_attn_fwd_f8(qkv8, scales, biasT, maskT, *, nB_, nW, T, ws, heads, C, bias_index=None):
    """Attention forward on e4m3-stored q | k | v (gemm_nt_qkv_fp8): bf16 [rows][C]."""
    assert qkv8.dtype == torch.uint8 and qkv8.is_contiguous() and scales.dtype == torch.float32 and scales.is_contiguous()
    out = torch.empty(qkv8.shape[0], C, dtype=torch.bfloat16, device=qkv8.device)
    with _Span("attn_fwd_f8", 4.0 * nB_ * heads * (T * ws * ws) ** 2 * (C // heads)):
        rc = load().stswin_win_attn_fwd_f8(_p(qkv8), _c_long(qkv8.shape[1]), _p(scales), _c_long(scales.shape[1]), _p(out), _c_long(_ld(out)),
                                           _p(biasT), _p(maskT), nB_, nW, T, ws, heads, C, _bias_windows(biasT, maskT, nW, bias_index),
                                           _p(bias_index), _stream())
    _check(rc, "win_attn_fwd_f8")
    return out


def win_attn_bwd_f8(qkv8, scales, dout, biasT, maskT, dbiasT, *, nB_, nW, T, ws, heads, C, scale, colsum_out=None, bias_index=None):
    """Attention backward on e4m3-stored q | k | v: bf16 dqkv [rows][3C]."""
    dqkv = torch.empty(qkv8.shape[0], 3 * C, dtype=torch.bfloat16, device=qkv8.device)
    lib = load()
    need = lib.stswin_win_attn_bwd_scratch(nB_, ws, heads, C)
    if need < 0:
        raise StswinHipError(f"win_attn_bwd_f8: bad geometry ({need})")
    sc = scratch(qkv8.device, need)
    with _Span("attn_bwd_f8", 10.0 * nB_ * heads * (T * ws * ws) ** 2 * (C // heads)):
        rc = lib.stswin_win_attn_bwd_f8(_p(qkv8), _c_long(qkv8.shape[1]), _p(scales), _c_long(scales.shape[1]), _p(dout), _c_long(_ld(dout)),
                                        _p(dqkv), _c_long(_ld(dqkv)), _p(biasT), _p(maskT), _p(dbiasT), _p(colsum_out), nB_, nW, T, ws, heads, C,
                                        _c_float(scale), _bias_windows(biasT, maskT, nW, bias_index), _p(bias_index), _p(sc),
                                        _c_long(sc.numel()), _stream())
    _check(rc, "win_attn_bwd_f8")
    return dqkv


def win_attn_qkv_fwd(x, rmap, w, bqkv, biasT, *, nB_, nW, T, ws, heads, C, scale, bias_index=None, want_qkv=True, debug_ts=None):
    """QKV-fused window attention forward (include/stswin_hip.h: stswin_win_attn_qkv_fwd) -> (out [nB_*T*ws*ws][C], qkv or None)."""
    rows = nB_ * T * ws * ws
    if x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16:
        raise StswinHipError("win_attn_qkv_fwd is a bf16 kernel")
    out = torch.empty(rows, C, dtype=x.dtype, device=x.device)
    qkv = torch.empty(rows, 3 * C, dtype=x.dtype, device=x.device) if want_qkv else None
    if debug_ts is not None:                     # tools/attn_qkv_timeline.py: int64 [256][8][8] stamp buffer in place of qkv_out
        rc = load().stswin_win_attn_qkv_fwd(_p(x), _c_long(_ld(x)), _c_long(x.shape[0]), _p(rmap), _p(w), _c_long(_ld(w)), _p(bqkv), _p(debug_ts), _c_long(0),
                                            _p(out), _c_long(_ld(out)), _p(biasT), nB_, nW, T, ws, heads, C, _c_float(scale),
                                            _bias_windows(biasT, None, nW, bias_index) | (1 << 30), _p(bias_index), _stream())
        _check(rc, "win_attn_qkv_fwd")
        return out, None
    name = "attn_qkv_fwd_bf16"
    with _Span(name, 2.0 * rows * 3 * C * C + 4.0 * nB_ * heads * (T * ws * ws) ** 2 * (C // heads)):
        rc = load().stswin_win_attn_qkv_fwd(_p(x), _c_long(_ld(x)), _c_long(x.shape[0]), _p(rmap), _p(w), _c_long(_ld(w)), _p(bqkv), _p(qkv),
                                            _c_long(_ld(qkv) if qkv is not None else 0), _p(out), _c_long(_ld(out)), _p(biasT), nB_, nW, T,
                                            ws, heads, C, _c_float(scale), _bias_windows(biasT, None, nW, bias_index), _p(bias_index),
                                            _stream())
    _check(rc, "win_attn_qkv_fwd")
    return out, qkv


def win_attn_qkv_fwd_ok(x: torch.Tensor) -> bool:
    """Whether the fused kernel's 32-bit buffer offsets reach every token row of x (stswin_win_attn_qkv_fwd returns -1208 otherwise)."""
    return x.shape[0] * _ld(x) * 2 <= 0xFFFF0000


def win_attn_bwd(qkv, dout, biasT, maskT, dbiasT, *, nB_, nW, T, ws, heads, C, scale, colsum_out=None, debug_ts=False,
                 bias_index=None):
    dqkv = torch.empty_like(qkv)
    lib = load()
    need = lib.stswin_win_attn_bwd_scratch(nB_, ws, heads, C)
    if need < 0:
        raise StswinHipError(f"win_attn_bwd: bad geometry ({need})")
    sc = scratch(qkv.device, need)
    rc = load().stswin_win_attn_bwd(_dt(qkv), _p(qkv), _c_long(_ld(qkv)), _p(dout), _c_long(_ld(dout)), _p(dqkv),
                                    _c_long(_ld(dqkv)), _p(biasT), _p(maskT), _p(dbiasT), _p(colsum_out), nB_, nW, T, ws, heads, C,
                                    _c_float(scale), _bias_windows(biasT, maskT, nW, bias_index) | ((1 << 30) if debug_ts else 0),
                                    _p(bias_index), _p(sc), _c_long(sc.numel()), _stream())
    _check(rc, "win_attn_bwd")
    return dqkv


# ----------------------------------------------------------------------------------------------- head ops
def colstats(x, groups=1, squares=True, M=None, unit=0):
    M = x.shape[0] if M is None else M
    C = x.shape[1]
    both = zeros(2 if squares else 1, groups, C, device=x.device)
    s = both[0]
    ss = both[1] if squares else None
    lib = load()
    need = lib.stswin_colstats_scratch(_dt(x), M, C, groups, unit)
    if need < 0:
        raise StswinHipError(f"colstats: bad geometry ({need})")
    _check(lib.stswin_colstats(_dt(x), _p(x), _c_long(_ld(x)), _p(s), _p(ss), M, C, groups, unit, _p(scratch(x.device, need)), _stream()),
           "colstats")
    return s, ss


def bn_finalize(x, s, ss, running_mean, running_var, groups=1, eps=1e-5, momentum=0.1, M=None, unit=0, raw=False):
    """raw=True: s / ss are plain sums (from a GEMM epilogue, hip.cs_group_reduce), not pivot-shifted ones (hip.colstats)."""
    M = x.shape[0] if M is None else M
    C = x.shape[1]
    mean = torch.empty(groups, C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(groups, C, dtype=torch.float32, device=x.device)
    _check(load().stswin_bn_finalize(_dt(x), _p(None if raw else x), _c_long(_ld(x)), _p(s), _p(ss), _p(mean), _p(rstd), _p(running_mean),
                                     _p(running_var), M, C, groups, _c_float(eps), _c_float(momentum), unit, _stream()),
           "bn_finalize")
    return mean, rstd


def bn_apply(x, mean, rstd, gamma, beta, out, resid=None, groups=1, relu=True, M=None, unit=0):
    M = x.shape[0] if M is None else M
    _check(load().stswin_bn_apply(_dt(x), _p(x), _c_long(_ld(x)), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(resid),
                                  _c_long(_ld(resid) if resid is not None else 0), _p(out), _c_long(_ld(out)), M,
                                  x.shape[1], groups, 1 if relu else 0, unit, _stream()), "bn_apply")
    return out


def bn_bwd(dy, x, y, mean, rstd, gamma, dx, dresid=None, groups=1, relu=True, training=True, M=None, phase=0, sums=None,
           rows_total=0, beta=None, unit=0, group_sums=None):
    M = x.shape[0] if M is None else M
    C = x.shape[1]
    if sums is None:
        both = zeros(2, groups, C, device=x.device)
        s1, s2 = both[0], both[1]
    else:
        s1, s2 = sums
    lib = load()
    need = lib.stswin_bn_bwd_scratch(_dt(x), M, C, groups, unit) if phase != 2 else 0
    if need < 0:
        raise StswinHipError(f"bn_bwd: bad geometry ({need})")
    _check(lib.stswin_bn_bwd(_dt(x), _p(dy), _c_long(_ld(dy)), _p(x), _c_long(_ld(x)), _p(y),
                                _c_long(_ld(y) if y is not None else 0), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(s1), _p(s2),
                                _p(dx), _c_long(_ld(dx)), _p(dresid), _c_long(_ld(dresid) if dresid is not None else 0), M, C,
                                groups, 1 if relu else 0, 1 if training else 0, phase, _c_long(rows_total), unit, _p(group_sums),
                                _p(scratch(x.device, need) if phase != 2 else None), _stream()), "bn_bwd")
    return s1, s2


def bn_relu_pool(x, mean, rstd, gamma, beta, frames, H, W, groups=1, unit=0):
    """-> (pooled [frames*Hp*Wp][C], arg uint8): BatchNorm + ReLU + MaxPool2d(3, 2, 1) of x [frames*H*W][C] in one pass."""
    C = x.shape[1]
    Hp, Wp = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty(frames * Hp * Wp, C, dtype=x.dtype, device=x.device)
    arg = torch.empty(frames * Hp * Wp, C, dtype=torch.uint8, device=x.device)
    _check(load().stswin_bn_relu_pool(_dt(x), _p(x), _c_long(_ld(x)), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(out), _c_long(_ld(out)),
                                      _p(arg), frames, H, W, C, groups, unit, _stream()), "bn_relu_pool")
    return out, arg


def rows_broadcast(v, out, groups, scale=1.0, accumulate=False, M=None):
    M = out.shape[0] if M is None else M
    _check(load().stswin_rows_broadcast(_dt(out), _p(v), _p(out), _c_long(_ld(out)), M, v.shape[1], groups,
                                        _c_float(scale), 1 if accumulate else 0, _stream()), "rows_broadcast")
    return out


def bilinear(src, dst, frames, h, w, H, W, backward=False):
    """forward: src [F*h*w][C] -> dst [F*H*W][C]; backward: src = d(out) [F*H*W][C] -> dst = d(in) [F*h*w][C]."""
    C = dst.shape[1]
    _check(load().stswin_bilinear(_dt(src), _p(src), _c_long(_ld(src)), _p(dst), _c_long(_ld(dst)), frames, h, w, H, W, C,
                                  1 if backward else 0, _stream()), "bilinear")
    return dst


def logits_upsample(tokens, nchw, frames, h, w, H, W, nc, backward=False):
    _check(load().stswin_logits_upsample(_dt(tokens), _p(tokens), _c_long(_ld(tokens)), _p(nchw), frames, h, w, H, W, nc,
                                         1 if backward else 0, _stream()), "logits_upsample")


def ce_fwd(logits, labels, ignore_index, thresh):
    F_, nc = logits.shape[:2]
    HW = logits[0, 0].numel()
    loss = torch.empty(F_ * HW, dtype=torch.float32, device=logits.device)
    stats = zeros(4, device=logits.device)          # [count, -, 64-bit fixed-point sum]: include/stswin_hip.h
    _check(load().stswin_ce_fwd(_dt(logits), _p(logits), _p(labels), _p(loss), _p(stats), F_, _c_long(HW), nc, ignore_index,
                                _c_float(thresh), _stream()), "ce_fwd")
    return loss, stats


_OHEM_WORK_BYTES = 3 * 2048 * 12 + 48


def ohem_select(loss, stats, n_min: int, thresh: float):
    """(value [1], sel [3]) of OhemCELoss2D's selection (losses.py:35-39) on the device: no sort, no host sync."""
    work = torch.empty(_OHEM_WORK_BYTES // 4, dtype=torch.int32, device=loss.device)
    value = torch.empty((), dtype=torch.float32, device=loss.device)
    sel = torch.empty(3, dtype=torch.float32, device=loss.device)
    _check(load().stswin_ohem_select(_p(loss), _c_long(loss.numel()), _c_long(n_min), _c_float(thresh), _p(stats), _p(work),
                                     _c_long(_OHEM_WORK_BYTES), _p(value), _p(sel), _stream()), "ohem_select")
    return value, sel


def ce_bwd(logits, labels, loss, sel, gscale, ignore_index):
    F_, nc = logits.shape[:2]
    HW = logits[0, 0].numel()
    d = torch.empty_like(logits)
    _check(load().stswin_ce_bwd(_dt(logits), _p(logits), _p(labels), _p(loss), _p(sel), _p(gscale), _p(d), F_, _c_long(HW),
                                nc, ignore_index, _stream()), "ce_bwd")
    return d


def contrast_fwd(q, keys, lq, lks, N, HW):
    """q [N*HW][C], keys 5 x [N*HW][C] (same pitch), lq / lks int32 [N][HW] -> pos, all fp32 [N][HW][5]."""
    C = q.shape[1]
    assert len(keys) == 5 and len(lks) == 5 and all(k.dtype == q.dtype and _ld(k) == _ld(keys[0]) for k in keys)
    pos = torch.empty(N, HW, 5, dtype=torch.float32, device=q.device)
    tot = torch.empty(N, HW, 5, dtype=torch.float32, device=q.device)
    K5 = (_c_void_p * 5)(*[k.data_ptr() for k in keys])
    L5 = (_c_void_p * 5)(*[l.data_ptr() for l in lks])
    with _Span("contrast_fwd_bf16" if q.dtype == torch.bfloat16 else "contrast_fwd_f32", 2.0 * N * HW * 5 * HW * C):
        rc = load().stswin_contrast_fwd(_dt(q), _p(q), _c_long(_ld(q)), K5, _c_long(_ld(keys[0])), _p(lq), L5, _p(pos),
                                        _p(tot), N, HW, C, _stream())
    _check(rc, "contrast_fwd")
    return pos, tot


_BANK_WS = {}


def _bank_workspace(device, floats):
    ws = _BANK_WS.get(device)
    if ws is None or ws.numel() < floats:
        ws = torch.empty(max(floats, 1 << 22), dtype=torch.float32, device=device)
        _BANK_WS[device] = ws
    return ws


def contrast_bank_fwd(Q, lq, bank, lb, *, q_sets, q_block, bank_block, gmap, inv_tau=1.0, want_lse=False, unit_rows=False):
    """Q [M][C], lq int32 [M]; bank [maps][seg][C], lb int32 [maps][seg]; gmap: q_sets lists of map indices (one per group).
    -> pos, all fp32 [M][groups] (+ rowmax, lse fp32 [M] or None, None); see include/stswin_hip.h."""
    M, C = Q.shape
    maps, seg = bank.shape[0], bank.shape[1]
    groups = len(gmap[0])
    assert Q.dtype == bank.dtype and bank.is_contiguous() and lb.is_contiguous() and lq.is_contiguous()
    assert lq.dtype == torch.int32 and lb.dtype == torch.int32 and len(gmap) == q_sets and all(len(g) == groups for g in gmap)
    if lq.numel() != M or lb.shape != (maps, seg):
        raise StswinHipError(f"contrast_bank_fwd: {lq.numel()} query labels for {M} query rows, bank labels {tuple(lb.shape)} for a "
                             f"{maps} x {seg} bank")
    pos = torch.empty(M, groups, dtype=torch.float32, device=Q.device)
    tot = torch.empty(M, groups, dtype=torch.float32, device=Q.device)
    rowmax = torch.empty(M, dtype=torch.float32, device=Q.device) if want_lse else None
    lse = torch.empty(M, dtype=torch.float32, device=Q.device) if want_lse else None
    ws = _bank_workspace(Q.device, 4 * M * groups * 8)
    gm = (_c_int * (q_sets * groups))(*[int(v) for row in gmap for v in row])
    name = "contrast_bank_fwd_bf16" if Q.dtype == torch.bfloat16 else "contrast_bank_fwd_f32"
    fn = load().stswin_contrast_bank_fwd_unit if unit_rows else load().stswin_contrast_bank_fwd     # unit_rows: L2-normalised Q / bank rows
    with _Span(name, 2.0 * M * groups * bank_block * C):
        rc = fn(_dt(Q), _p(Q), _c_long(_ld(Q)), _p(lq), M, C, q_sets, q_block, _p(bank),
                                             _c_long(bank.stride(1)), _p(lb), maps, seg, bank_block, groups, gm, _c_float(inv_tau),
                                             _p(pos), _p(tot), _p(rowmax), _p(lse), _p(ws), _c_long(ws.numel()), _stream())
    _check(rc, "contrast_bank_fwd")
    return pos, tot, rowmax, lse


def contrast_class_sums(bank, lb, bank_block, ncls):
    """-> ksum fp32 [maps][seg / bank_block][ncls + 1][C]: per-class sums of the bank rows (slot ncls: all rows)."""
    maps, seg, C = bank.shape
    ksum = torch.empty(maps, seg // bank_block, ncls + 1, C, dtype=torch.float32, device=bank.device)
    lib = load()
    need = lib.stswin_contrast_class_sums_scratch(maps, seg, bank_block, C, ncls)
    if need < 0:
        raise StswinHipError(f"contrast_class_sums: bad geometry ({need})")
    _check(lib.stswin_contrast_class_sums(_dt(bank), _p(bank), _c_long(bank.stride(1)), _p(lb), maps, seg, bank_block, C, ncls,
                                          _p(ksum), _p(scratch(bank.device, need)), _stream()), "contrast_class_sums")
    return ksum


def contrast_bank_dq(dpos, dneg, cnt, lq, ksum, *, q_sets, q_block, seg, bank_block, gmap):
    M, groups = dpos.shape
    C, ncls = ksum.shape[3], ksum.shape[2] - 1
    dq = torch.empty(M, C, dtype=torch.float32, device=dpos.device)
    gm = (_c_int * (q_sets * groups))(*[int(v) for row in gmap for v in row])
    _check(load().stswin_contrast_bank_dq(_p(dpos.contiguous()), _p(dneg.contiguous()), _p(cnt.contiguous()), _p(lq), _p(ksum), _p(dq), _c_long(C), M, C,
                                          q_sets, q_block, seg, bank_block, ncls, groups, gm, _stream()), "contrast_bank_dq")
    return dq


def rownorm_scatter(X: torch.Tensor, Y: torch.Tensor, views: int, HW: int, samples: int, want_inv: bool = False):
    """Y[view][sample * HW + px] = normalize(X[(sample * views + view) * HW + px]) (fp32 arithmetic, Y's dtype = X's); -> inv or None."""
    R, C = X.shape
    assert Y.dtype == X.dtype and Y.shape == (R, C) and X.stride(1) == 1 and Y.stride(1) == 1
    inv = torch.empty(R, dtype=torch.float32, device=X.device) if want_inv else None
    _check(load().stswin_rownorm_scatter(_dt(X), _p(X), _c_long(X.stride(0)), _p(Y), _c_long(Y.stride(0)), _p(inv), R, C, views, HW, samples,
                                         _stream()), "rownorm_scatter")
    return inv


def rownorm_scatter_bwd(X: torch.Tensor, inv: torch.Tensor, dY: torch.Tensor, views: int, HW: int, samples: int) -> torch.Tensor:
    R, C = X.shape
    assert dY.dtype == torch.float32 and dY.shape == (R, C) and dY.stride(1) == 1
    dX = torch.empty(R, C, dtype=X.dtype, device=X.device)
    _check(load().stswin_rownorm_scatter_bwd(_dt(X), _p(X), _c_long(X.stride(0)), _p(inv), _p(dY), _c_long(dY.stride(0)), _p(dX), _c_long(C), R, C,
                                             views, HW, samples, _stream()), "rownorm_scatter_bwd")
    return dX


def labels_resize(masks, h: int, w: int) -> torch.Tensor:
    """len(masks) float label maps (N, 1, Hs, Ws) -> int32 [maps][N * h * w] (nearest neighbour + truncation)."""
    N, _, Hs, Ws = masks[0].shape
    ms = [m.contiguous() if (m.dtype == torch.float32 and m.is_contiguous()) else m.float().contiguous() for m in masks]
    assert all(m.shape == (N, 1, Hs, Ws) and m.is_cuda for m in ms)
    lb = torch.empty(len(ms), N * h * w, dtype=torch.int32, device=ms[0].device)
    arr = (_c_void_p * len(ms))(*[m.data_ptr() for m in ms])
    _check(load().stswin_labels_resize(arr, len(ms), N, Hs, Ws, h, w, _p(lb), _stream()), "labels_resize")
    return lb


def label_counts(lq: torch.Tensor, lb: torch.Tensor, *, q_sets, q_block, bank_block, ncls, gmap) -> torch.Tensor:
    """cnt fp32 [M][groups]: visible bank rows of each group whose label equals the query's."""
    M, (maps, seg), groups = lq.numel(), lb.shape, len(gmap[0])
    if M % q_sets or (M // q_sets) % q_block or seg % bank_block or not lq.is_contiguous() or not lb.is_contiguous():
        raise StswinHipError(f"label_counts: {M} query labels do not split into {q_sets} sets of {q_block}-row blocks "
                             f"(bank {maps} x {seg}, blocks of {bank_block})")
    hist = torch.empty(maps, seg // bank_block, ncls, dtype=torch.int32, device=lq.device)
    cnt = torch.empty(M, groups, dtype=torch.float32, device=lq.device)
    gm = (_c_int * (q_sets * groups))(*[int(v) for row in gmap for v in row])
    _check(load().stswin_label_counts(_p(lq), _p(lb), M, maps, seg, q_sets, q_block, bank_block, ncls, groups, gm, _p(hist), _p(cnt), _stream()),
           "label_counts")
    return cnt


def pair_loss(pos, tot, cnt, q_sets: int, visible: int) -> torch.Tensor:
    M, groups = pos.shape
    loss = torch.empty(1, dtype=torch.float32, device=pos.device)
    _check(load().stswin_pair_loss(_p(pos), _p(tot), _p(cnt), M, groups, q_sets, visible, _p(loss), _stream()), "pair_loss")
    return loss


def pair_loss_bwd(pos, tot, cnt, dloss, q_sets: int, visible: int):
    M, groups = pos.shape
    dpos, dneg = torch.empty_like(pos), torch.empty_like(pos)
    _check(load().stswin_pair_loss_bwd(_p(pos), _p(tot), _p(cnt), _p(dloss), M, groups, q_sets, visible, _p(dpos), _p(dneg), _stream()),
           "pair_loss_bwd")
    return dpos, dneg


def upsample_argmax(logits, H, W, gt=None):
    """NCHW logits [F][nc][h][w] -> uint8 labels [F][H][W] (+ int32 counts [F][3][nc] when gt int64 [F][H][W] is given)."""
    F_, nc, h, w = logits.shape
    lg = logits.contiguous()
    labels = torch.empty(F_, H, W, dtype=torch.uint8, device=logits.device)
    counts = torch.zeros(F_, 3, nc, dtype=torch.int32, device=logits.device) if gt is not None else None
    _check(load().stswin_upsample_argmax(_dt(lg), _p(lg), _p(labels), _p(gt.contiguous() if gt is not None else None),
                                         _p(counts), F_, nc, h, w, H, W, _stream()), "upsample_argmax")
    return labels, counts


def optim_tick(kind: int, counter: torch.Tensor, hyper: torch.Tensor, a: float, b: float) -> None:
    """Advance a device-resident step counter (int32 [1]) and derive the step's scalars into hyper (fp32 [4]); include/stswin_hip.h."""
    assert counter.dtype == torch.int32 and hyper.dtype == torch.float32 and hyper.numel() >= 4 and counter.is_cuda and hyper.is_cuda
    _check(load().stswin_optim_tick(int(kind), _p(counter), _p(hyper), ctypes.c_double(a), ctypes.c_double(b), _stream()), "optim_tick")


def multi_tensor(mode, ps, gs, ms=None, vs=None, lr=0.0, b1=0.0, b2=0.0, eps=0.0, wd=0.0, c1=1.0, c2=1.0, hyper=None):
    """mode 0 Adam / 1 SGD-momentum / 2 EMA over lists of fp32 tensors (chunks of 48 tensors per launch).  hyper (device fp32 [4] =
    {lr, c1, c2, EMA momentum}): read the step-dependent scalars from it instead of lr / c1 / c2 / b1(EMA) - for SGD c1 != 0 still
    marks the first step."""
    lib = load()
    st = _stream()
    for t in ps:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise StswinHipError("multi_tensor needs contiguous fp32 GPU tensors")
    for lo in range(0, len(ps), 48):
        hi = min(len(ps), lo + 48)
        k = hi - lo
        arr = lambda ts: (_c_void_p * k)(*[t.data_ptr() for t in ts[lo:hi]]) if ts is not None else None  # noqa: E731
        ns = (_c_int * k)(*[t.numel() for t in ps[lo:hi]])
        if hyper is not None:
            _check(lib.stswin_multi_tensor_dev(mode, k, arr(ps), arr(gs), arr(ms), arr(vs), ns, _p(hyper), _c_float(b1), _c_float(b2),
                                               _c_float(eps), _c_float(wd), 1 if (mode == 1 and c1 != 0.0) else 0, st), "multi_tensor_dev")
        else:
            _check(lib.stswin_multi_tensor(mode, k, arr(ps), arr(gs), arr(ms), arr(vs), ns, _c_float(lr), _c_float(b1),
                                           _c_float(b2), _c_float(eps), _c_float(wd), _c_float(c1), _c_float(c2), st),
                   "multi_tensor")


def multi_tensor_lars(ps, gs, ms, norms, *, lr, momentum, wd, trust_coef, eps, first, adaptive, hyper=None):
    """LARS-scaled SGD-momentum step of one parameter group (lists of contiguous fp32 GPU tensors; chunks of 48 tensors,
    two launches each: norms, update).  norms: fp32 scratch of >= 96 floats."""
    lib = load()
    st = _stream()
    for t in list(ps) + list(gs) + list(ms):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise StswinHipError("multi_tensor_lars needs contiguous fp32 GPU tensors")
    for lo in range(0, len(ps), 48):
        hi = min(len(ps), lo + 48)
        k = hi - lo
        arr = lambda ts: (_c_void_p * k)(*[t.data_ptr() for t in ts[lo:hi]])  # noqa: E731
        ns = (_c_int * k)(*[t.numel() for t in ps[lo:hi]])
        need = 2 * k + 2 * sum((t.numel() + 8191) // 8192 for t in ps[lo:hi])
        if norms is None or norms.numel() < need:
            norms = scratch(ps[lo].device, need)
        if hyper is not None:          # the learning rate from device memory (hyper[0])
            _check(lib.stswin_multi_tensor_lars_dev(k, arr(ps), arr(gs), arr(ms), ns, _p(norms), _c_long(norms.numel()), _p(hyper),
                                                    _c_float(momentum), _c_float(wd), _c_float(trust_coef), _c_float(eps), 1 if first else 0,
                                                    1 if adaptive else 0, st), "multi_tensor_lars_dev")
            continue
        _check(lib.stswin_multi_tensor_lars(k, arr(ps), arr(gs), arr(ms), ns, _p(norms), _c_long(norms.numel()), _c_float(lr), _c_float(momentum),
                                            _c_float(wd), _c_float(trust_coef), _c_float(eps), 1 if first else 0,
                                            1 if adaptive else 0, st), "multi_tensor_lars")


def proxy_collective(src: torch.Tensor, dst: torch.Tensor, workgroups: int, passes: int) -> None:
    """Measurement stand-in for an RCCL all-reduce of a bucket (include/stswin_hip.h; tools/overlap_proxy.py): `workgroups` workgroups hold
    their compute units while they copy src -> dst `passes` times, on the current stream."""
    assert src.numel() * src.element_size() == dst.numel() * dst.element_size() and src.is_contiguous() and dst.is_contiguous()
    _check(load().stswin_proxy_collective(_p(src), _p(dst), _c_long(src.numel() * src.element_size()), workgroups, passes, _stream()),
           "proxy_collective")


def set_cu_budget(cus: int) -> int:
    """Compute units the one-workgroup-per-CU launches plan for (0 = the whole device); returns the previous setting."""
    return int(load().stswin_set_cu_budget(int(cus)))


def calibrate(device=None, seconds: float = 0.1) -> dict:
    """Box calibration for bench lines (include/stswin_hip.h: stswin_calib_mfma / stswin_calib_copy): sustained bf16 MFMA TFLOP/s of a
    register-only loop (4 waves per CU) and sustained copy TB/s (read + write) over 2 x 512 MB, each the best of three timed launches
    of ~`seconds`/3 after a warm-up launch.  HIP events on the current stream."""
    lib = load()
    dev = torch.device(device if device is not None else "cuda")
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    st = _stream()

    def timed(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) * 1e-3

    iters = int(300000 * max(seconds, 0.01) / 0.1)
    per = int(lib.stswin_calib_mfma(4, 20000, _p(sink), st))                      # warm-up
    t = min(timed(lambda: lib.stswin_calib_mfma(4, iters, _p(sink), st)) for _ in range(3))
    tflops = per * iters * 16384.0 / t / 1e12
    n = 512 << 20
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    src.zero_()
    _check(lib.stswin_calib_copy(_p(src), _p(dst), _c_long(n), st), "calib_copy")
    tc = min(timed(lambda: lib.stswin_calib_copy(_p(src), _p(dst), _c_long(n), st)) for _ in range(3))
    del src, dst
    return {"mfma_bf16_tflops": tflops, "mfma_probe_ms": t * 1e3, "copy_tbps": 2.0 * n / tc / 1e12, "copy_probe_ms": tc * 1e3}


def selftest(which: int) -> torch.Tensor:
    out = torch.zeros(16384, dtype=torch.float32, device="cuda")
    if which == 5:
        out[:512] = torch.arange(512, dtype=torch.float32, device="cuda")
    _check(load().stswin_selftest(_p(out), which, _stream()), "selftest")
    torch.cuda.synchronize()
    return out.cpu()
