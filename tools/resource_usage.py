#!/usr/bin/env python3
"""Register / scratch budget of every kernel of libstswin_hip, read from hipcc's `-Rpass-analysis=kernel-resource-usage` remarks
(the compiler's own report: no GPU needed).  A scratch-resident value is not just slow memory on this path: hipcc follows every
scratch reload with `s_waitcnt vmcnt(0)`, i.e. a wait for the kernel's own output stores and for every prefetch in flight
 - so the product kernels must have ScratchSize 0, and tests/test_resource_usage.py fails when one does not.

    python tools/resource_usage.py                 # table of all kernels
    python tools/resource_usage.py --scratch-only   # only the ones with scratch
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stswincl_amd", "csrc")
UNITS = ["gemm", "rowops", "attention", "headops", "contrast", "optim", "conv_halo"]

# Kernels that may keep scratch, each with the reason it is not on the bf16 product path.  Everything else must be clean.
ALLOWED_SCRATCH = [
    (r"^void attn_bwd_kernel<float,", "fp32 parity instantiations (exact-f32 MFMA path of the tests, not the bf16 product path)"),
    (r"^void attn_bwd_kernel<__bf16, 128, (128|64|32), 0,", "4-wave stage-1-shaped backward: reduced-width test geometries and the "
                                                           "STSWIN_ATTN_BWD4 A/B switch; production stage 1 runs attn_bwd8_kernel"),
    (r"^void attn_bwd_kernel<__bf16, 32, 256, 0,", "stage-2 backward with a run-time window size: test geometries only (the model's 4x4 "
                                                   "windows take the <.., 16, ..> instantiations)"),
    (r"^void attn_qkv_fwd_kernel<0, ", "fused QKV + attention with a run-time window size / other widths: test geometries only (the "
                                       "model's stage 1 takes <64, 512>)"),
    (r"^void contrast_bank_kernel<float,", "fp32 parity instantiation"),
    (r"^(void )?ln_bwd_kernel<__bf16, (Li)?8", "LayerNorm backward over 2048 < C <= 4096 (eight 16-byte pieces per lane; 256 VGPRs + AGPR spill space + a "
                                              "few bytes of scratch in the accumulate + column-sum form): wider than any LayerNorm of the models "
                                              "(<= 2048 = PatchMerging's 4C at stage 1), kept for the ABI's stated range"),
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (not os.path.isabs(cand) or os.path.exists(cand)):
            return cand
    return "hipcc"


def _demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return [o.strip() for o in out[:len(names)]]
    except Exception:
        return list(names)


def _unit(unit: str, extra=()):
    with tempfile.TemporaryDirectory() as td:
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-inline-asm", "-I",
               os.path.join(ROOT, "include"), *extra, "-c", os.path.join(CSRC, unit + ".hip"), "-o", os.path.join(td, unit + ".o"),
               "-Rpass-analysis=kernel-resource-usage"]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {unit}.hip:\n{p.stderr[-2000:]}")
        text = p.stderr + p.stdout
    rows = []
    for blk in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        def g(key):
            m = re.search(key + r": (\d+)", blk)
            return int(m.group(1)) if m else -1
        rows.append(dict(unit=unit, mangled=blk.split()[0], scratch=g(r"ScratchSize \[bytes/lane\]"), vgpr=g("VGPRs"), agpr=g("AGPRs"),
                         sgpr=g("SGPRs"), occupancy=g(r"Occupancy \[waves/SIMD\]"), lds=g(r"LDS Size \[bytes/block\]"),
                         sgpr_spill=g("SGPRs Spill")))
    return rows


def collect(units=UNITS, jobs: int = 0, extra=()):
    jobs = jobs or min(len(units), os.cpu_count() or 1)
    with ThreadPoolExecutor(jobs) as ex:
        rows = [r for rs in ex.map(lambda u: _unit(u, extra), units) for r in rs]
    for r, n in zip(rows, _demangle([r["mangled"] for r in rows])):
        r["kernel"] = re.sub(r"^_Z\d+", "", n) if n.startswith("_Z") else n
        # c++filt leaves the __bf16 template argument of some names mangled (DF16b): read it for the allow-list
        r["kernel"] = r["kernel"].replace("IDF16b", "<__bf16, ")
    return rows


def allowed_reason(kernel: str):
    k = kernel if kernel.startswith("void ") else "void " + kernel
    k = re.sub(r"attn_bwd_kernel<__bf16, Li(\d+)ELi(\d+)ELi(\d+)ELb(\d)EEv8AttnArgs", lambda m: f"attn_bwd_kernel<__bf16, {m.group(1)}, {m.group(2)}, {m.group(3)}, {'true' if m.group(4) == '1' else 'false'}>(AttnArgs)", k)
    for pat, why in ALLOWED_SCRATCH:
        if re.search(pat, k):
            return why
    return None


def main():
    rows = collect(extra=("-DSTSWIN_TUNING",) if "--tuning" in sys.argv else ())
    only = "--scratch-only" in sys.argv
    print(f"{'unit':10s} {'scratch':>7s} {'VGPR':>4s} {'AGPR':>4s} {'SGPR':>4s} {'sspl':>4s} {'occ':>3s} {'LDS':>7s}  kernel   (sspl = SGPRs spilled into VGPR lanes)")
    bad = 0
    for r in sorted(rows, key=lambda r: (r["unit"], r["kernel"])):
        if only and r["scratch"] <= 0:
            continue
        why = allowed_reason(r["kernel"]) if r["scratch"] > 0 else None
        mark = "" if r["scratch"] <= 0 else (f"   [allowed: {why}]" if why else "   <-- PRODUCT KERNEL WITH SCRATCH")
        bad += r["scratch"] > 0 and not why
        print(f"{r['unit']:10s} {r['scratch']:7d} {r['vgpr']:4d} {r['agpr']:4d} {r['sgpr']:4d} {r['sgpr_spill']:4d} {r['occupancy']:3d} {r['lds']:7d}  {r['kernel'][:150]}{mark}")
    print(f"{len(rows)} kernels, {sum(r['scratch'] > 0 for r in rows)} with scratch, {bad} of them product kernels")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
