#!/usr/bin/env python3
"""bench.py -- frames/s of the STswinCL segmentation training step (fwd + bwd + Adam) on MI355X.

Workload = BASELINE.json configs[1], as far as the reference can run it: TswinPlus(12),
B = 4 clips/GPU of T = 4 frames (the reference asserts T == 4), 3x512x512, bf16 compute / fp32 master weights,
OHEM cross-entropy (n_min = 512*512/16), Adam lr 1e-4 (seg18/train_swin.py:122), synthetic frames and labels
resident in HBM, random-init weights.  One step = model fwd + loss + bwd (+ gradient all-reduce) + optimizer.

    python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process starts N ranks itself (a child `python -m torch.distributed.run
--nproc-per-node N ... bench.py <same flags>`, BEFORE anything here touches the GPU) and exits with the child's code;
under torch.distributed.run (WORLD_SIZE set) it is one rank of the job.

Prints ONE JSON line on rank 0 (contract in the task statement).  Order of a run (round 6): W warm-up steps -> `calibration`
probes (0.1 s: an MFMA loop on pseudo-random operands, a 1 GB copy) -> 2 more untimed steps -> EXACTLY K timed steps between barrier +
synchronize, with NO event brackets inside (`value`; a hipGraph replay of the whole step at N = 1 - a training mode, stswincl_amd/graph.py -,
eager launches at N > 1) -> the graph is released -> a separate eager pass of 2 x stride steps in which one launch in `stride` of every
kernel family is bracketed by HIP events on the launch stream (`roofline`, `profile_pass`; not part of `value`) -> `secondary` (BASELINE's
second metric, contrastive pairs/s, from a short run of the configs[3] step, and its inter-video bank mode) -> `cpu_baseline` (the CPU
oracle timed on this host, N = 1 only).  `value_normalised` = `value` referred to a box whose MFMA probe reads bench.py's CAL_REF.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="clips per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-size", type=int, default=512)
    ap.add_argument("--no-profile", action="store_true", help="skip the bracketed roofline pass behind the timed region")
    ap.add_argument("--no-calibration", action="store_true", help="skip the 0.2 s box-calibration probes")
    ap.add_argument("--profile-stride", type=int, default=9,
                    help="bracket one launch in n of each kernel family (seeded random choice) with HIP events (1 = all; the "
                         "records cost host time)")
    ap.add_argument("--bank", default="sample", choices=["sample", "batch", "world"],
                    help="contrast workload: key set a query pixel sees - its own sample (the reference), every sample of the rank, "
                         "or every sample of every rank (RCCL all-gather of the keys: the inter-video bank of BASELINE configs[3])")
    ap.add_argument("--comm-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="N > 1: wire dtype of the gradient all-reduce buckets (bf16 halves the bytes on the xGMI links; the averaged "
                         "gradients are converted back to fp32 behind the collective)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short contrastive run that fills `secondary`")
    ap.add_argument("--secondary-steps", type=int, default=20)
    ap.add_argument("--dump-prof", default=None,
                    help="write the per-span timing table here (with STSWIN_SHAPE_PROFILE=1: one row per GEMM shape)")
    ap.add_argument("--workload", default="seg", choices=["seg", "contrast"],
                    help="seg (default, BASELINE configs[1]) or contrast (configs[3]: ConsistencyLoss pre-training step)")
    ap.add_argument("--graph", type=int, default=-1, help="1: capture the whole step in a hipGraph and replay it (a training mode: "
                    "stswincl_amd/graph.py); 0: eager; -1 (default): graph when N == 1")
    return ap.parse_args()


def physical_cores():
    """(physical cores this process may run on, logical CPUs it may run on): distinct (package, core) pairs of /proc/cpuinfo
    among the CPUs of the affinity mask."""
    try:
        allowed = set(os.sched_getaffinity(0))
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    cores, cpu, pkg = set(), None, 0
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("processor"):
                    cpu, pkg = int(line.split(":")[1]), 0
                elif line.startswith("physical id"):
                    pkg = int(line.split(":")[1])
                elif line.startswith("core id") and cpu in allowed:
                    cores.add((pkg, int(line.split(":")[1])))
    except OSError:
        pass
    return (len(cores) or len(allowed)), len(allowed)


def cpu_baseline(size: int, budget_s: float = 60.0):
    """The oracle (a port of the reference graph) doing the same training step on the host cores (SURVEY 8(d) protocol).

    fwd+bwd+Adam steps of B = 2 clips (B = 1 cannot train: ASPP's BatchNorm on a 1x1 map) in fp32 AT THE BENCH SIZE (`size` x `size`
    frames: SURVEY 8(d) says B = 2 at 512x512) - measured, never extrapolated from a smaller frame (the CPU step does not scale with
    the pixel count: 256x256 is ~7x cheaper than 512x512, not 4x).  Best of as many repeats as fit the budget (at least one, at
    most three; every time is listed in `sample`).  Threads: the physical cores of the affinity mask are detected and stated; the
    thread count actually used is the faster of {32, min(physical cores, 64)} on a 128x128 calibration step (both timings are in
    `sample`, with torch.__config__.parallel_info())."""
    from oracle import stswin_oracle as O
    from stswincl_amd.net.Ours.base18 import TswinPlus
    phys, logical = physical_cores()

    def one_step(sz):
        torch.manual_seed(0)
        model = TswinPlus(12, (sz // 8, sz // 8))
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        params = {k: sd[k].requires_grad_(True) for k, _ in model.named_parameters()}
        del model
        opt = torch.optim.Adam(list(params.values()), 1e-4)
        x = torch.randn(2, 4, 3, sz, sz)
        y = torch.randint(0, 12, (2, sz, sz))
        t0 = time.perf_counter()
        logits = O.tswin_plus(x, sd, True)
        loss = O.ohem_ce(logits, y, sz * sz // 16)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return time.perf_counter() - t0

    t_start = time.perf_counter()
    cal = {}
    # candidates: 32 and min(physical cores, 64).  One thread per physical core on the pool's 128-core hosts was measured in rounds
    # 3-4 (this calibration step: 0.97 vs 5.20 s, 1.22 vs 17.50 s) - 5-14x SLOWER than 32 threads on the shared machine, and
    # that one measurement ate the budget of the 512x512 repeats; it is no longer tried above 64
    phys_note = ""
    for th in sorted({max(1, min(32, phys)), max(1, min(64, phys)), max(1, phys)}):
        torch.set_num_threads(th)
        t64 = one_step(64)             # thread-pool / allocator warm-up
        if th > 64 and t64 > 2.0:      # one thread per physical core (SURVEY 8(d)'s N): tried, but not allowed to eat the budget again
            phys_note = (f"; one thread per physical core ({th}) not used: its 64x64 warm-up step alone took {t64:.2f} s "
                         f"(oversubscribed shared host)")
            continue
        cal[th] = one_step(128)
    threads = min(cal, key=cal.get)
    torch.set_num_threads(threads)
    times = []
    while len(times) < 3:
        times.append(one_step(size))
        if time.perf_counter() - t_start + min(times) > budget_s:
            break
    dt = min(times)
    pinfo = " ".join(torch.__config__.parallel_info().split())
    return {"value": 2 * 4 / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "physical_cores": phys, "logical_cpus": logical,
            "sample": f"best of {len(times)} fwd+bwd+Adam step(s) of the CPU oracle, B=2 clips x 4 frames at {size}x{size} fp32 (the bench's "
                      f"frame size, measured directly, nothing scaled): {', '.join(f'{t:.2f}' for t in times)} s on {threads} threads "
                      f"({phys} physical cores / {logical} logical CPUs in the affinity mask; 128x128 calibration step: "
                      f"{', '.join(f'{t:.2f} s on {th} threads' for th, t in sorted(cal.items()))}{phys_note}); "
                      f"parallel_info: {pinfo[:400]}"}


class Ctx:
    """Rank / device / process group of this process (one process per GPU)."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        self.share = os.environ.get("STSWIN_BENCH_SHARE_GPU") == "1"   # functional test of the N > 1 path on a 1-GPU box
        ndev = torch.cuda.device_count()
        if self.share:
            local = local % max(1, ndev)
        elif self.world > ndev:
            raise SystemExit(f"bench.py: {self.world} ranks but {ndev} GPUs visible (STSWIN_BENCH_SHARE_GPU=1 + gloo shares one "
                             f"GPU for a functional test)")
        torch.cuda.set_device(local)
        self.dev = torch.device("cuda", local)
        self.backend = os.environ.get("STSWIN_DIST_BACKEND", "gloo" if self.share else "nccl")
        self.ranks_seen = 1
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(self.backend, rank=self.rank, world_size=self.world,
                                    **({"device_id": self.dev} if self.backend == "nccl" else {}))
            one = torch.ones(1, device=self.dev if self.backend == "nccl" else "cpu")
            dist.all_reduce(one)                       # every rank really is in the collective
            self.ranks_seen = int(one.item())

    def barrier(self):
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, dt: float) -> float:
        if self.world == 1:
            return dt
        tt = torch.tensor([dt], dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def dist_info(self, reducer):
        from stswincl_amd import hip as _hip
        return {"backend": ("rccl (torch.distributed 'nccl')" if self.backend == "nccl" else self.backend),
                "rccl_ranks": self.ranks_seen,
                "comm_dtype": (str(reducer.comm_dtype).replace("torch.", "") if (reducer is not None and reducer.comm_dtype is not None) else "float32"),
                "allreduce_bytes_per_step_per_rank": reducer.bytes_per_step() if reducer is not None else 0,
                # (an overlapped reducer switches the weight-gradient GEMMs to the separate split-K combine pass: stswincl_amd/dp.py)
                "tn_split_k_combine": ("separate pass" if (os.environ.get("STSWIN_TN_FUSED") == "0" or
                                                           (os.environ.get("STSWIN_TN_FUSED") is None and _hip.tn_fused_holds() > 0))
                                       else "fused into the GEMM launch"),
                "shared_gpu_functional_test": self.share}


def timed_steps(ctx, step, steps, warmup, profile_stride=0):
    """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides; MAX over ranks.  With
    profile_stride the launches of rank 0 are bracketed by HIP events (hip._Span) - the headline region is timed WITHOUT them
    (profile_stride = 0) and the roofline figures come from a separate bracketed pass of the same step."""
    from stswincl_amd import hip
    loss = None
    for _ in range(warmup):
        loss = step()
    ctx.barrier()
    if profile_stride and ctx.rank == 0:
        hip.profile_begin(profile_stride)
    t0 = time.perf_counter()
    for k in range(steps):
        if profile_stride and ctx.rank == 0:
            hip.profile_step(k)               # which launches are timed rotates with the step: every site once per `stride` steps
        loss = step()
    ctx.barrier()
    dt = time.perf_counter() - t0
    prof = hip.profile_end() if (profile_stride and ctx.rank == 0) else {}
    return ctx.max_over_ranks(dt), prof, loss


def capture(step_fn, zero_grad, optimizers=()):
    """hipGraph capture of a whole training step as a TRAINING mode (stswincl_amd/graph.py): every kernel of libstswin_hip is launched
    on the current stream with caller-owned workspaces and no host sync, and the optimizers' step-dependent scalars live in device
    memory, so N replays are N training steps (tests/test_hip_graph_training.py: bit for bit the eager steps).  The helper's two
    warm-up executions are real (untimed) steps."""
    from stswincl_amd.graph import GraphedStep
    return GraphedStep(step_fn, list(optimizers), zero_grad=zero_grad, warmup=2)


# Calibration reference of `value_normalised` (see calibration_block).  From the round-6 development boxes
# (profiles/r06_calibration_boxes.txt): MFMA probe 1870-1876 TFLOP/s <-> 597.8-598.9 frames/s, 1944 <-> 610.2, 1975 <-> 610.3,
# 2010 <-> 632.9.  Least squares of (value / 603) against (probe / 1900) over the four kinds of box gives a slope of 0.70: about 70 % of the
# step (the MFMA-bound GEMM families) moves with the probe, the HBM-bound passes and the output-bound K = 512 GEMMs do not.  The probe
# itself repeats to +-1.5 %, which is the residual spread of the normalised values (594-609 against 598-633 raw).  The copy probe is
# reported but NOT used: it reads 4.85-5.11 TB/s without ordering the boxes.
CAL_REF = {"mfma_bf16_tflops": 1900.0, "copy_tbps": 5.0}
CAL_WEIGHT_MFMA = 0.7


def matmul_probe(dev):
    """A third, code-independent probe: the vendor library's bf16 GEMM (torch.matmul, 8192^3, random operands), best of three timings of ten
    launches.  Measurement only - the library is not on the product path - but it loads the chip like the step's GEMMs do (MFMA under real
    operand traffic, power-limited), which the register-only loop does not fully capture."""
    A = torch.randn(8192, 8192, device=dev).bfloat16()
    B = torch.randn(8192, 8192, device=dev).bfloat16()
    C = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(A, B.t(), out=C)
    best = None
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            torch.matmul(A, B.t(), out=C)
        b.record()
        b.synchronize()
        t = a.elapsed_time(b) * 1e-3 / 10
        best = t if best is None or t < best else best
    del A, B, C
    return 2.0 * 8192.0 ** 3 / best / 1e12


def calibration_block(cal, value):
    """`cal` = hip.calibrate() readings taken after the warm-up steps (~0.1 s of fixed probes: a register-only MFMA loop on pseudo-random
    operands - power-limited like the step's GEMMs - and a 1 GB copy), so that lines from different boxes can be compared:
    value_normalised = value x (w / (mfma / mfma_ref) + (1 - w)) - the throughput this tree would show on a box whose MFMA probe reads
    the reference value, under the model that a share w of the step scales with the probe and the rest does not."""
    cal = dict(cal)
    rm, rc = cal["mfma_bf16_tflops"] / CAL_REF["mfma_bf16_tflops"], cal["copy_tbps"] / CAL_REF["copy_tbps"]
    scale = CAL_WEIGHT_MFMA / rm + (1.0 - CAL_WEIGHT_MFMA)
    cal.update({"reference": dict(CAL_REF), "mfma_weight": CAL_WEIGHT_MFMA, "relative_mfma": rm, "relative_copy": rc,
                "normalisation": "value x (w / relative_mfma + 1 - w); the copy probe and the vendor-matmul probe (torch.matmul 8192^3 bf16) are informational",
                "probe": "256 workgroups x 4 waves x 8 independent v_mfma_f32_16x16x32_bf16 chains on pseudo-random operands (~20 ms, best "
                         "of 3); 16-byte-lane copy 512 MB -> 512 MB (best of 3); HIP events; after the warm-up steps, before the timed region"})
    return cal, value * scale


def contrast_run(a, ctx, steps, warmup, profile_stride, batch=8, bank="sample", eager=False):
    """BASELINE.json configs[3] as far as the reference can run it: ConsistencyLoss (PixPro-style, 2 query + 6 momentum-key
    encoder passes) at 256x256 (224 is illegal for the window sizes), B clips/GPU, LARS over SGD-momentum as in
    main_pretrain_swinv5.py:37-47.  Reports contrastive pairs/s = 2 directions x B x HW x 5 HW per step."""
    import types
    from stswincl_amd.contrast.models.PixPro_swin_v5 import ConsistencyLoss
    from stswincl_amd.dp import GradBucketReducer
    from stswincl_amd.optim import make_contrast_optimizer
    world, rank, dev = ctx.world, ctx.rank, ctx.dev
    S, B = 256, batch
    args = types.SimpleNamespace(pixpro_p=1.0, pixpro_momentum=0.99, pixpro_clamp_value=0.0, pixpro_transform_layer=1,
                                 pixpro_ins_loss_weight=0.0, pixpro_pos_ratio=0.7, data="endo18", tag="1", pretrainpth=None,
                                 num_instances=2235, batch_size=B, epochs=150, start_epoch=1, pixpro_bank=bank)
    torch.manual_seed(0)
    model = ConsistencyLoss(args, input_resolution=(S // 8, S // 8)).to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt, opt_name = make_contrast_optimizer(params, batch_size=B * world)
    reducer = GradBucketReducer(params, bucket_mb=32.0, comm_dtype=torch.bfloat16 if a.comm_dtype == "bf16" else None) if world > 1 else None
    torch.manual_seed(1234 + rank)
    ims = [torch.randn(B, 4, 3, S, S, device=dev) for _ in range(6)]
    masks = [torch.randint(0, 12, (B, 1, S, S), device=dev).float() for _ in range(6)]

    def eager_step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = model(*ims, *masks)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        return loss

    step = eager_step
    graphed = not eager and (a.graph == 1 or (a.graph == -1 and world == 1))
    if graphed:
        # the contrastive step is ~2400 launches of 20-70 us kernels (8 encoder passes at 256x256): eager launches keep the GPU only
        # partly busy; one hipGraph replay removes the host.  LARS learning rate, EMA momentum schedule and its step counter are
        # device-resident (stswincl_amd/optim.py), so the replays walk the schedules like main_pretrain_swinv5.py's loop does
        step = capture(eager_step, lambda: opt.zero_grad(set_to_none=True), [opt])
        warmup = max(0, warmup - 2)            # (the capture helper ran two real steps)
    dt, _, loss = timed_steps(ctx, step, steps, warmup, 0)          # the headline region: no event brackets
    prof = {}
    if profile_stride:                         # separate bracketed eager pass: the similarity kernel's live roofline figures
        if graphed:                            # (drop the graph's memory pool first: see seg_run)
            import gc
            loss = loss.detach().clone()
            step = eager_step
            gc.collect()
            torch.cuda.empty_cache()
        _, prof, _ = timed_steps(ctx, eager_step, 8, 1, profile_stride)
    hw = (S // 8) ** 2
    visible = {"sample": hw, "batch": B * hw, "world": world * B * hw}[bank]      # key pixels of one key map a query pixel sees
    pairs = world * 2 * B * hw * 5 * visible * steps
    res = {"metric": f"contrastive pairs/s, ConsistencyLoss fwd+bwd+{opt_name} (2 query + 6 key encoder passes), 256x256",
           "value": pairs / dt, "unit": "pairs/s", "n_gpus": world, "steps": steps, "warmup": warmup + (2 if graphed else 0),
           "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"BASELINE.json configs[3]: PixPro-style ConsistencyLoss, {B} x 6 clips/GPU x T=4 x 3x{S}x{S} "
                                  f"(224 is illegal for windows 8/4), " +
                                  ("per-sample label-guided loss as in the reference" if bank == "sample" else
                                   f"inter-video key bank of {visible} entries per key map ({bank}: " +
                                   ("all-gathered over the ranks)" if bank == "world" else "every sample of the rank)")) + f"; {opt_name}",
                      "bank_entries_per_key_map": visible,
                      "clips_per_gpu": 6 * B, "parallelism": f"dp{world}",
                      "input_frames_per_s": world * 6 * B * 4 * steps / dt, "loss": float(loss.detach()),
                      "launch": "hipGraph replay of the whole step (device-resident optimizer / EMA scalars: a training mode)" if graphed else "eager launches"},
           "dist": ctx.dist_info(reducer)}
    for name in ("contrast_fwd_bf16", "contrast_bank_fwd_bf16"):
        if name in prof:
            q = prof[name]
            tf = q["work"] / (q["ms_total"] * 1e-3) / 1e12
            res["roofline"] = {"kernel": name, "bound": "mfma", "achieved": tf, "peak": PEAK_BF16_TFLOPS,
                               "unit": "TFLOP/s", "frac": tf / PEAK_BF16_TFLOPS, "traffic": None,
                               "avg_launch_ms": q["ms_avg"], "pairs_per_s_in_kernel": q["work"] / 512.0 / (q["ms_total"] * 1e-3)}
    del model, opt, reducer, ims, masks, step, eager_step
    return res


def seg_run(a, ctx):
    from stswincl_amd import hip
    from stswincl_amd.dp import GradBucketReducer
    from stswincl_amd.net.Ours.base18 import TswinPlus
    from stswincl_amd.optim import FusedAdam
    from stswincl_amd.utils.losses import OhemCELoss2D
    world, rank, dev = ctx.world, ctx.rank, ctx.dev
    S, B = a.size, a.batch
    torch.manual_seed(0)
    model = TswinPlus(12, (S // 8, S // 8)).to(dev)
    model.train()
    if world > 1:   # identical initial weights on every rank
        for p in model.parameters():
            if ctx.backend == "nccl":
                dist.broadcast(p.data, 0)
            else:
                t = p.data.cpu()
                dist.broadcast(t, 0)
                p.data.copy_(t)
    profile_stride = 0 if a.no_profile else max(1, a.profile_stride)
    use_graph = a.graph == 1 or (a.graph == -1 and world == 1)
    opt = FusedAdam(model.parameters(), 1e-4)          # == torch.optim.Adam (tests/test_hip_optim.py), 8 launches per step
    crit = OhemCELoss2D(S * S // 16)
    comm_dtype = torch.bfloat16 if a.comm_dtype == "bf16" else None
    reducer = GradBucketReducer(model.parameters(), bucket_mb=32.0, comm_dtype=comm_dtype) if world > 1 else None
    torch.manual_seed(1234 + rank)            # each rank owns different clips (weak scaling)
    x = torch.randn(B, 4, 3, S, S, device=dev)
    y = torch.randint(0, 12, (B, S, S), device=dev)

    def eager_step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(x)
            loss = crit(out, y)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        return loss

    step = eager_step
    graph_note = None
    graph_wanted = use_graph
    warm_left = a.warmup
    if use_graph and world > 1 and ctx.backend != "nccl":
        use_graph, graph_note = False, "eager launches (hipGraph capture needs the RCCL backend: gloo collectives run on the host)"
    if use_graph:
        try:      # N > 1: the bucket all-reduces are enqueued on the reducer's side stream inside the capture (RCCL supports capture)
            step = capture(eager_step, lambda: opt.zero_grad(set_to_none=True), [opt])
            warm_left = max(0, a.warmup - 2)   # (the capture helper ran two real steps)
        except Exception as e:     # noqa: BLE001
            if world == 1:
                raise
            use_graph, graph_note = False, f"eager launches (FALLBACK: hipGraph capture of the {world}-rank step failed: {type(e).__name__}: {str(e)[:120]})"
            if os.environ.get("STSWIN_BENCH_STRICT_GRAPH") == "1":      # tests: a requested capture must not fall back silently
                raise
            torch.cuda.synchronize()
    for _ in range(warm_left):
        step()
    ctx.barrier()
    cal = hip.calibrate(dev) if not a.no_calibration else None      # every rank probes its own GPU at the same time; rank 0 reports
    if cal is not None:
        try:
            cal["matmul_bf16_tflops"] = matmul_probe(dev)
        except Exception as e:     # noqa: BLE001   (informational: a missing / failing vendor library must not cost the line)
            cal["matmul_bf16_tflops"] = None
            cal["matmul_probe_error"] = f"{type(e).__name__}: {e}"[:200]
    # headline: exactly K steps, NO event brackets, graph replay when N == 1 (two more untimed steps re-warm the caches behind the probes)
    dt, _, loss = timed_steps(ctx, step, a.steps, 2 if cal is not None else 0, 0)
    # roofline pass: the same step, eager, rank 0's launches bracketed by HIP events (one launch in `stride` per family, rotating, so that
    # 2 x stride steps time every launch site twice); NOT part of `value`
    prof, prof_steps, dt_prof = {}, 0, None
    if profile_stride:
        prof_steps = 2 * profile_stride if profile_stride > 1 else 4
        # (the graph and its private memory pool are dropped first: eager steps beside a live capture of the same step run 6-7 % slower on
        #  the DEVICE - 28.05 against 26.30 ms, back to 26.7 once the pool is released; tools/eager_after_graph.py - so the kernels would be
        #  timed in a memory layout the headline never sees)
        if use_graph:
            import gc
            loss = loss.detach().clone()
            step = eager_step
            gc.collect()
            torch.cuda.empty_cache()
        dt_prof, prof, _ = timed_steps(ctx, eager_step, prof_steps, 2, profile_stride)
    frames = world * B * 4 * a.steps
    res = {
        "metric": "input frames/s, TswinPlus fwd+bwd+Adam, 4-frame 512x512 clips", "value": frames / dt,
        "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[1]: TswinPlus(12) seg train step, {B} clips/GPU x T=4 frames "
                               f"x 3x{S}x{S}, swin res ({S // 8},{S // 8}), OHEM-CE, Adam; T=4 because the reference "
                               f"asserts it (swin_512.py:313)",
                   "clips_per_gpu": B, "frames_per_clip": 4, "parallelism": f"dp{world}",
                   "launch": ("hipGraph replay of the whole step (device-resident Adam step count / bias corrections: a training mode, "
                              "bit-identical to eager steps - tests/test_hip_graph_training.py); no event brackets in the timed region"
                              if use_graph else (graph_note or "eager launches; no event brackets in the timed region")),
                   "graph_requested": bool(graph_wanted),
                   "loss": float(loss.detach())},
        "dist": ctx.dist_info(reducer),
    }
    if cal is not None:
        res["calibration"], res["value_normalised"] = calibration_block(cal, frames / dt)
    if dt_prof is not None:
        res["profile_pass"] = {"steps": prof_steps, "ms_per_step": 1e3 * dt_prof / prof_steps, "launch": "eager launches, one launch in "
                               f"{profile_stride} per kernel family bracketed by HIP events (the source of `roofline`; not part of `value`)"}
    psteps = max(prof_steps, 1)
    if prof and a.dump_prof and rank == 0:
        with open(a.dump_prof, "w") as f:
            f.write(f"# {psteps} steps; ms are totals over those steps\n")
            for n in sorted(prof, key=lambda n: -prof[n]["ms_total"]):
                q = prof[n]
                f.write(f"{q['ms_avg'] * q['launches'] / psteps:9.3f} ms/step {q['launches'] // psteps:4d} x {1e3 * q['ms_avg']:8.1f} us "
                        f"{q['work'] / max(q['ms_total'], 1e-9) / 1e9:8.1f} TF/s  {n}\n")
    if prof:
        k = max(prof, key=lambda n: prof[n]["ms_avg"] * prof[n]["launches"])
        q = prof[k]
        tf = q["work"] / (q["ms_total"] * 1e-3) / 1e12
        # HBM bytes per launch: PMC counters need rocprofv3 around the process, so they are collected by tools/profile_round.sh
        # (separate --pmc passes over this same command) into profiles/rNN_pmc_dominant_kernel.json, stamped with the commit it
        # was measured on; the newest round's file is read here
        traffic, traffic_src, traffic_commit = None, None, None
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_dominant_kernel.json")), reverse=True):
            try:
                with open(path) as f:
                    pmc = json.load(f)
                if pmc.get("kernel") == k:
                    traffic, traffic_src, traffic_commit = pmc["hbm_bytes_per_launch"], os.path.basename(path), pmc.get("commit")
                    break
            except Exception:
                pass
        res["roofline"] = {"kernel": k, "bound": "mfma", "achieved": tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                           "frac": tf / PEAK_BF16_TFLOPS, "traffic": traffic,
                           "traffic_static_source": f"profiles/{traffic_src}: bytes per launch, (2*FETCH_SIZE + WRITE_SIZE)*1024 from "
                                                    f"separate rocprofv3 --pmc passes over this command; a committed number, "
                                                    f"not collected during this run (measured on commit {traffic_commit})" if traffic_src else None,
                           "algorithmic_flops_per_launch": q["work"] / q["sampled"],
                           # (context for `frac`, measured once, not during this run: under MFMA load on random operands the shader clock
                           #  of this kernel's main loop is 1.66-1.78 GHz, not 2.4 - profiles/r04_gemm_clock_under_load.txt)
                           "peak_note": "data-sheet dense bf16 peak at 2.4 GHz; the chip is power-limited under MFMA load (1.66-1.78 GHz "
                                        "measured inside this kernel's main loop on random operands: 1.73-1.85 PFLOP/s at that clock)",
                           "launches_per_step": q["launches"] / psteps,
                           "launches_timed": q["sampled"],
                           "timing": f"HIP events on the launch stream around one launch in {profile_stride} in a separate bracketed pass of the same step behind the timed region; "
                                     f"which ones rotates with the step, so every launch site of the step is timed "
                                     f"{psteps // profile_stride if profile_stride and psteps % profile_stride == 0 else '~' + str(round(psteps / max(profile_stride, 1), 1))} "
                                     f"time(s); averages are over the timed launches",
                           "avg_launch_ms": q["ms_avg"], "ms_per_step": q["ms_avg"] * q["launches"] / psteps,
                           "other_kernels": {n: {"ms_per_step": v["ms_avg"] * v["launches"] / psteps,
                                                 "tflops": v["work"] / (v["ms_total"] * 1e-3) / 1e12}
                                             for n, v in prof.items() if n != k}}
    del model, opt, reducer, x, y, step, eager_step
    return res


def under_profiler() -> bool:
    """rocprofv3 preloads its tool library into the profiled process, and that library initialises the GPU before main() runs."""
    if any(k.startswith(("ROCPROFILER_", "ROCP_", "ROCPROF_")) for k in os.environ):
        return True
    return any(t in os.environ.get("LD_PRELOAD", "") for t in ("rocprofiler", "rocprofv3", "roctracer"))


def launch_ranks(a) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run job.  Nothing in this
    process has touched the GPU (only `import torch`), and nothing is exec'ed: the parent waits and returns the child's code.
    NOT under rocprofv3: there the profiler's preloaded library has already initialised the GPU in this process, and starting
    launchers from such a process is what takes machines of this pool down - profiled runs are single-GPU."""
    if under_profiler():
        print("bench.py: refusing to start ranks from a profiled process (rocprofv3 has initialised the GPU here). Profile with "
              "--gpus 1, or put the profiler inside the job: torch.distributed.run ... bench.py under WORLD_SIZE.", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    ctx = Ctx()
    if a.gpus != ctx.world and ctx.rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={ctx.world}; reporting n_gpus = {ctx.world}", file=sys.stderr)
    import __graft_entry__ as ge
    if ctx.rank == 0 and not os.path.exists(ge.LIB):
        ge.build(verbose=False)
    if ctx.world > 1:
        dist.barrier()
    from stswincl_amd import hip
    hip.load()
    if a.workload == "contrast":
        res = contrast_run(a, ctx, a.steps, a.warmup, 0 if a.no_profile else a.profile_stride, batch=(a.batch if a.batch != 4 else 8),
                           bank=a.bank)
    else:
        res = seg_run(a, ctx)
        if not a.no_secondary:
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            hip.arena_reset()
            try:                       # the primary line must survive whatever happens in the second workload
                # graph replay when N == 1: the LARS learning rate, the EMA momentum schedule and its counter are device-resident, so
                # every replay advances them as main_pretrain_swinv5.py's loop does
                sec = contrast_run(a, ctx, a.secondary_steps, 3, 0)
                res["secondary"] = {k: sec[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype")}
                res["secondary"]["config"] = sec["config"]
                # BASELINE configs[3]'s own mode - the inter-video key bank (every sample of the rank: 8192 entries per key map at
                # 8 clips; `world`: all-gathered over the ranks) - on the driver's line too: a short run (graph replay at N = 1), then the
                # similarity kernel timed live in a bracketed eager pass (every launch of it: hip._PROFILE_ALWAYS)
                del sec
                gc.collect()
                torch.cuda.empty_cache()
                hip.arena_reset()
                bank_mode = "world" if ctx.world > 1 else "batch"
                secb = contrast_run(a, ctx, 8, 3, 9, bank=bank_mode)
                res["secondary"]["bank"] = {"mode": bank_mode, "value": secb["value"], "unit": secb["unit"], "steps": secb["steps"],
                                            "ms_per_step": secb["ms_per_step"],
                                            "bank_entries_per_key_map": secb["config"]["bank_entries_per_key_map"],
                                            "roofline": {k: secb["roofline"][k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms")}
                                            if "roofline" in secb else None}
            except Exception as e:     # noqa: BLE001
                res["secondary"] = {"error": f"{type(e).__name__}: {e}"[:400]}
        if ctx.world == 1 and not a.no_cpu_baseline and ctx.rank == 0:
            res["cpu_baseline"] = cpu_baseline(a.cpu_size)
    if ctx.rank == 0:
        print(json.dumps(res), flush=True)
    if ctx.world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
