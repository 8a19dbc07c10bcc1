#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): restored 512x512 images/s at 4 denoise steps.

One "step" = one pass of the whole hot path over one batch of synthetic degraded images that are already resident
in HBM:  vae_encode -> q_sample(t=200) -> 4 x (ControlNet + ControlledUNet + sampler update) -> vae_decode,
through the reference-shaped API (ControlLDM / Diffusion / SpacedSampler of edtr_amd) on hand-written HIP kernels.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W          # one rank per GPU; batch axis sharded, no collective in the loop

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (dominant kernel = the MFMA implicit-GEMM, timed per launch
with HIP events on the launch stream) and, at N=1, "cpu_baseline" (the CPU oracle timed on the host cores on a
bounded sample of the same workload) plus "parity" (GPU vs oracle on that sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

USED_TIMESTEPS = [50, 100, 150, 200]
PEAK_TFLOPS = 2500.0       # dense bf16/fp16 MFMA peak, MI355X_MICROARCH.md chip table
# What a bare stream of v_mfma_f32_32x32x16_bf16 on random operands delivers at the clock the chip HOLDS under that load (1.69 GHz; the
# 2.5 PFLOP/s figure assumes 2.4 GHz): tools/exp/attn_issue_bound.hip, profiles/r06/attn_issue_bound.log.  Reported beside `frac` (which
# stays against the contract's peak) so that a reader can tell "far from the pipes' rate" from "far from the data-sheet clock".
SUSTAINED_TFLOPS = 1769.0
FLOP_PER_IMAGE = 7.925e12  # VAE-enc 1.1167 + 4 x 1.0734 + VAE-dec 2.5145 TFLOP per 512x512 image
# 50-step variant 57.30 TFLOP/image; tiled 1024^2 = 38.6 (9 windows x 4 steps) + 6.2 (tiled encoder) + 10.5 (untiled decoder)
FLOP_PER_IMAGE_BY_WORKLOAD = {"det512": FLOP_PER_IMAGE, "det512s50": 57.30e12, "seg1024tiled": 55.4e12}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60,
                    help="timed passes (default 60: >= 4 s of timed region at the headline workload, so that a few-percent kernel change "
                         "is visible on the driver's record; the 50-step / tiled workloads default to 8 / 16 passes)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step (BASELINE config 2: 8)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--config", default="sd21", choices=["sd21", "tiny"])
    ap.add_argument("--precision", default="fast", choices=["fast", "mixed", "high", "hybrid", "robust"],
                    help="fast = 16-bit activation storage in --dtype (headline); mixed = the fast parity mode (fp32 stream, fp16 "
                         "operands, 1-3 products per layer class: edtr_amd/precision.py); high = the robust parity mode (fp32 stream, "
                         "bf16 split-3 products everywhere).  Both parity modes meet the 1e-3 north-star tolerance")
    ap.add_argument("--parity-steps", type=int, default=24,
                    help="after the headline (fast) measurement, time this many passes of the parity mode (mixed) in the same run and "
                         "report them as \"parity_mode\" (0 = skip; skipped for N > 1 and for the non-default workloads)")
    ap.add_argument("--also", default="det512s50,seg1024tiled",
                    help="after the headline (default workload, N = 1): also time these BASELINE configurations briefly in the same run "
                         "and report them under \"other_workloads\" (\"none\" = skip)")
    ap.add_argument("--breakdown-json", default=None, help="write the per-launch-name time table of the roofline pass to this file")
    ap.add_argument("--dup", default=None, help="measurement aid: issue every idempotent launch whose name contains this string twice "
                                                "(marginal wall-clock cost of a kernel class inside the overlapped graphs)")
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)   # launcher self-test: rendezvous only, no model
    ap.add_argument("--workload", default="det512", choices=["det512", "seg1024tiled", "det512s50"],
                    help="det512 = BASELINE configs[1] (default); seg1024tiled = configs[3]: one 1024x1024 image, tiled VAE encoder "
                         "(256-px tiles), latent-tiled denoiser (64/32 latent tiles), untiled decoder (demo.py:99-124); det512s50 = configs[4] per GPU: batch 4 of 512x512, 50-step sampler from pure noise")
    ap.add_argument("--no-graph", action="store_true", help="replay launch lists eagerly instead of hipGraphs")
    ap.add_argument("--inflight", type=int, default=2, choices=[1, 2, 3, 4],
                    help="batches in flight: 2 = consecutive steps alternate between two HIP streams / buffer sets, so "
                         "the low-occupancy phases of one batch overlap the other (throughput mode)")
    ap.add_argument("--serial-lanes", action="store_true", help="capture ControlNet and the UNet encoder on one stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print a per-kernel-name time table to stderr")
    ap.add_argument("--swinir", action="store_true",
                    help="also time the SwinIR pre-restoration in front of the path (excluded from the metric, SURVEY.md §8d) and "
                         "report it separately as \"pre_restoration\" (the default run does: --also none skips it)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without torch.distributed.run: start the N ranks ourselves.  This process has made no GPU
        # call (importing torch does not initialise HIP) and makes none: it only waits and forwards rank 0's JSON line.
        raise SystemExit(launch_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    tiled = args.workload == "seg1024tiled"
    s50 = args.workload == "det512s50"
    steps_given = any(a == "--steps" or a.startswith("--steps=") for a in sys.argv[1:])
    if tiled:
        args.batch, args.size = 1, 1024        # (two images in flight like the other workloads: 10.57 -> 11.52 images/s, same device, round 3)
        args.no_cpu_baseline = True
        if not steps_given:
            args.steps = 16
    if s50:
        args.batch, args.size = int(os.environ.get("EDTR_S50_BATCH", "4")), 512       # (EDTR_S50_BATCH: experiments only — configs[4] is batch 4 per GPU)
        args.no_cpu_baseline = True
        if not steps_given:
            args.steps = 8
    # ---- the CPU-baseline child is spawned before this process makes any HIP call; it builds its fp32 weights, reports "ready"
    #      and idles.  The GPU side waits for that "ready" before its warm-up, so the two never compete for the host cores.
    cpu_handle = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_handle = start_cpu_baseline(args.config, args.size)
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    if args.dry_run:
        raise SystemExit(dry_run_rank(rank, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the EDTR MI355X path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or bool(os.environ.get("EDTR_BENCH_DIST"))   # EDTR_BENCH_DIST=1: exercise the RCCL path on one rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:       # EDTR_BENCH_DIST=1 from a plain shell: a one-rank rendezvous on the loopback
            for k_, v_ in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(_free_port()))):
                os.environ.setdefault(k_, v_)
        dist.init_process_group(backend="nccl", device_id=dev)   # nccl == RCCL on ROCm
    rccl_ranks = count_ranks(dist, dev) if dist is not None else 1
    if rccl_ranks != world:
        raise SystemExit(f"bench.py: RCCL sees {rccl_ranks} ranks, expected {world}")

    from edtr_amd import synth, workloads
    from edtr_amd.diffusion import Diffusion
    from edtr_amd.model import ControlLDM
    from edtr_amd.model.params import skip_init
    from edtr_amd.parallel import broadcast_packed, verify_packed_store
    from edtr_amd.sampler import SpacedSampler
    from edtr_amd.testing import rel_err, synthetic_state_dicts

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    cfg = synth.CONFIGS[args.config]()
    B, S = args.batch, args.size
    h = S // 8
    ctx_dim = cfg["unet_cfg"]["context_dim"]

    # ---- model: rank 0 materialises the synthetic checkpoint, the other ranks receive it by ONE bucketed RCCL
    #      broadcast over xGMI (start-up only; the denoise loop has no collective)
    t0 = time.time()
    with skip_init(), torch.device(dev):          # parameters are born on the device
        cldm = ControlLDM(**cfg)
    cldm.compute_dtype = dtype
    cldm.precision = args.precision
    if rank == 0:                                  # the "checkpoint" lives on rank 0 only (hashed on the device: bit-identical to the host)
        # EDTR_SYNTH_DEVICE=cpu: hash on the host (rocprofv3 --pmc FETCH_SIZE crashes inside torch's int64 elementwise kernels)
        sds = synthetic_state_dicts(cfg, None if os.environ.get("EDTR_SYNTH_DEVICE") == "cpu" else dev)
        cldm.unet.load_state_dict(sds["unet"], strict=True)
        cldm.load_controlnet_from_ckpt(sds["controlnet"])
        cldm.vae.load_state_dict(sds["vae"], strict=True)
    else:
        with torch.no_grad():
            for p_ in cldm.parameters():
                p_.zero_()
    cldm = cldm.eval().to(dev)
    log(f"[rank {rank}] model ready in {time.time() - t0:.1f}s")

    diffusion = Diffusion(linear_start=0.00085, linear_end=0.0120, timesteps=1000).to(dev)
    sampler = SpacedSampler(diffusion.betas)

    # ---- synthetic inputs for the GLOBAL batch, sliced per rank (results independent of the GPU count)
    GB = B * world
    inp = workloads.make_inputs(args.workload, ctx_dim, dev, B, S, rank, world, with_step_noises=s50)     # (the untimed parity pass injects them: rank 0 starts at image 0 = the golden's image)
    untiled_forward = cldm.forward

    def one_pass(inject=False):
        # inject=False: the sampler's per-step noise is drawn by torch.randn_like on the GPU INSIDE the pass, as the reference does
        # (utils/sampler.py:199) — the timed region does that work; inject=True (the parity pass after the timed region) feeds
        # the tensors the reference golden was made with
        img, z, _ = workloads.restore_pass(cldm, diffusion, sampler, inp, args.workload, untiled_forward, inject=inject)
        return img, z

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if cpu_handle is not None:
        wait_cpu_ready(cpu_handle)
    # ---- warm-up (builds the kernel programs on the first pass), optional hipGraph capture
    slots = list(range(args.inflight))
    streams = [torch.cuda.Stream() for _ in slots]
    t0 = time.time()
    for sl_ in slots:
        cldm.engine_slot = sl_
        img, z = one_pass()
    torch.cuda.synchronize()
    log(f"[rank {rank}] first pass (program build + weight packing) {time.time() - t0:.1f}s")
    bcast = None
    if use_dist:
        # ---- ONE bucketed RCCL broadcast of the PACKED weights over xGMI (start-up only; the denoise loop has no collective):
        #      every rank has built the same programs, the packed tensors are overwritten in place
        t0 = time.time()
        calls, nbytes = broadcast_packed(cldm, src=0)
        torch.cuda.synchronize()
        bcast = {"collectives": calls, "GiB": round(nbytes / 2**30, 3), "seconds": round(time.time() - t0, 2)}
        # every rank must now hold rank 0's packed store bit for bit: a receiver that re-packed from its placeholder parameters
        # would still "restore" images at full speed, so the throughput is refused unless the checksums agree on all ranks
        bcast["verified_on_all_ranks"] = verify_packed_store(cldm, src=0)
        log(f"[rank {rank}] packed weights broadcast over RCCL: {bcast}")
        if not bcast["verified_on_all_ranks"]:
            raise SystemExit("bench.py: a rank's packed weight store differs from rank 0's after the broadcast")
        for sl_ in slots:                          # results of the warm-up pass on the receiving ranks were computed from zeros
            cldm.engine_slot = sl_
            img, z = one_pass()
        torch.cuda.synchronize()
    if args.dup:
        n_dup = sum(e.step_prog.duplicate_launches(args.dup) for e in cldm._cldm_engines.values())
        n_dup += sum(e.prog.duplicate_launches(args.dup) for e in cldm._vae_engines.values())
        log(f"[rank {rank}] --dup {args.dup}: {n_dup} launches doubled (measurement aid: the throughput below is NOT a result)")

    def capture_all():
        if args.no_graph:
            return
        for e in cldm._cldm_engines.values():
            if e.step_prog.graph is None:
                e.step_prog.capture(parallel_lanes=not args.serial_lanes)
        for e in cldm._vae_engines.values():
            if e.prog.graph is None:
                e.prog.capture()

    capture_all()
    step_events = []

    def run_steps(n, record=False):
        out = None
        for i in range(n):
            k = i % args.inflight
            cldm.engine_slot = k
            with torch.cuda.stream(streams[k]):
                if record:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                out = one_pass()
                if record:
                    e1.record()
                    step_events.append((e0, e1))
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        return out

    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    img, z = run_steps(max(1, args.warmup) * args.inflight)
    barrier()
    t0 = time.perf_counter()
    img, z = run_steps(args.steps, record=True)
    barrier()
    elapsed = time.perf_counter() - t0
    # HIP events around every pass on its own stream: the latency of ONE pass while `inflight` passes overlap (not a throughput)
    lat = sorted(a.elapsed_time(b) for a, b in step_events)
    pass_latency_ms = lat[len(lat) // 2] if lat else None
    # completion-to-completion intervals of consecutive passes (HIP events, device clock): their median is the steady-state time
    # per pass without the host's start-up / drain effects that the wall clock includes
    ends = [b for _, b in step_events]
    nf = args.inflight      # passes i and i + nf complete on the same stream
    gaps = sorted(ends[i].elapsed_time(ends[i + nf]) / nf for i in range(len(ends) - nf)) if len(ends) > 2 * nf else []
    ms_per_step_median = gaps[len(gaps) // 2] if gaps else None
    # ---- parity pass (untimed): the same path once more with the reference golden's noise tensors injected
    cldm.engine_slot = 0
    img, z = one_pass(inject=True)
    torch.cuda.synchronize()
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = GB * args.steps / elapsed

    tol_key = args.precision if args.precision != "fast" else args.dtype
    std_shape = (B, S) == workloads.WORKLOADS[args.workload][:2]        # the FLOP count per image below is for this shape only
    result = {
        "metric": f"restored {S}x{S} images/sec @ {50 if s50 else 4} denoise steps",
        "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"high": "bf16 split-3 products over an fp32 stream (precision=high)",
                  "mixed": "fp16 1-3-part products over an fp32 stream (precision=mixed)",
                  "hybrid": "fp16 storage in the encoder and the denoiser, fp16 1-3-part products over an fp32 stream in the decoder (precision=hybrid)",
                  "robust": "fp16 three-part products over an fp32 stream in the denoiser, the mixed allocation in the VAE (precision=robust)"}.get(args.precision, args.dtype),
        "data": "synthetic",
        "config": {"workload": (f"EDTR-seg s4 ({args.config}), configs[3]: tiled vae_encode (256-px tiles) + q_sample(t=200) + 4 x latent-tiled "
                                f"(64/32) ControlNet+UNet + untiled vae_decode, batch {B}/GPU of {S}x{S}") if tiled else
                               (f"EDTR-det s50 ({args.config}), configs[4] per GPU: vae_encode + 50-step spaced sampler from pure noise "
                                f"(50 x (ControlNet+UNet)) + vae_decode, batch {B}/GPU of {S}x{S}") if s50 else
                               f"EDTR-det s4 ({args.config}): vae_encode + q_sample(t=200) + 4 x (ControlNet+UNet) + "
                               f"vae_decode, batch {B}/GPU of {S}x{S}", "global_batch": GB, "image_size": S,
                   "denoise_steps": 50 if s50 else 4, "parallelism": f"batch-sharded x{world}", "graphs": not args.no_graph,
                   "batches_in_flight": args.inflight, "precision": args.precision},
        "weight_broadcast": bcast, "rccl_ranks": rccl_ranks,
        "pass_latency_ms_median_hip_events": round(pass_latency_ms, 3) if pass_latency_ms is not None else None,
        "ms_per_step_median_hip_events": round(ms_per_step_median, 3) if ms_per_step_median is not None else None,
        "sampler_noise": "torch.randn_like on the GPU inside every timed pass (reference utils/sampler.py:199); the parity figures come "
                         "from one extra untimed pass with the golden's noise injected",
        "mfma_frac_whole_path": round(value * FLOP_PER_IMAGE_BY_WORKLOAD[args.workload] / (world * PEAK_TFLOPS * 1e12), 4)
        if (args.config == "sd21" and std_shape) else None,
    }
    if args.precision in ("mixed", "hybrid", "robust") and cldm._policy() is not None:
        result["config"]["precision_policy"] = cldm._policy().describe()
    if args.precision == "hybrid":
        from edtr_amd.model.cldm import hybrid_sections
        result["config"]["hybrid_sections"] = hybrid_sections()
    if args.dup:
        result["INVALID_measurement_aid"] = f"--dup {args.dup}"

    default_legs = (world == 1 and args.also != "none" and args.precision == "fast" and args.workload == "det512" and args.config == "sd21"
                    and std_shape and not args.dup and not args.no_graph)
    if rank == 0 and (args.swinir or default_legs):      # (on the default run too: the step in front of the path, on the driver's record)
        try:
            result["pre_restoration"] = swinir_leg(dev, dtype, B, S, args.steps)
        except Exception as e:       # reported separately: must never take the headline line down with it
            result["pre_restoration"] = {"error": repr(e)}
    if rank == 0 and not args.no_roofline:
        result.update(roofline_pass(cldm, args, ms_per_step))
    if (rank == 0 and world == 1 and args.also != "none" and args.precision == "fast" and args.workload == "det512" and args.config == "sd21"
            and std_shape and not args.dup and not args.no_graph):
        # ---- the other single-GPU BASELINE configurations, briefly, on the driver's record too
        result["other_workloads"] = {}
        for name in [w for w in args.also.split(",") if w in ("det512s50", "seg1024tiled")]:
            try:
                result["other_workloads"][name] = other_workload_leg(cldm, diffusion, sampler, name, dev, ctx_dim, args, capture_all, rel_err)
            except Exception as e:       # never take the headline down
                result["other_workloads"][name] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result.update(finish_cpu_baseline(cpu_handle, inp, img.cpu(), z.cpu(), S, rel_err, tol_key))
    if rank == 0 and args.config == "sd21":     # rank 0's shard starts at image 0 of the global batch: same images as the golden's
        result.update(golden_parity(args.workload, img, z, rel_err, tol_key))
    if (rank == 0 and world == 1 and args.parity_steps > 0 and args.precision == "fast" and args.workload == "det512"
            and args.config == "sd21" and std_shape and not args.dup):
        # ---- the modes that meet the north-star tolerance, timed in the SAME run (the headline above is the fast bf16 mode):
        #      "hybrid" = the fastest one on well-conditioned weights (round 6: fp16 encoder + denoiser, mixed decoder); "robust" = the
        #      cheapest one that holds 1e-3 on OUTLIER-BEARING weights too (round 6: three parts on every denoiser class, mixed VAE, q / k
        #      split; pinned against 1e-3 on the moderate set by tests/test_gpu_heavy.py); "high" = bf16 split-3 everywhere (a short leg).
        #      The all-sections "mixed" mode of rounds 3 - 5 is `--precision mixed` (89.9 images/s, 5.4e-4 / 6.1e-4 beside this run's figures).
        try:
            result["parity_mode"] = parity_mode_leg(cldm, args, one_pass, capture_all, run_steps, B, rel_err, mode="hybrid")
            result["parity_mode_robust"] = parity_mode_leg(cldm, args, one_pass, capture_all, run_steps, B, rel_err, mode="robust",
                                                           steps=max(4, args.parity_steps // 2))
            result["parity_mode_high"] = parity_mode_leg(cldm, args, one_pass, capture_all, run_steps, B, rel_err, mode="high",
                                                         steps=max(4, args.parity_steps // 3))
            pm = result["parity_mode"]
            pr = max((result[k] for k in ("parity_mode_robust", "parity_mode_high") if result[k].get("meets_north_star")),
                     key=lambda r: r["images_per_s"], default=result["parity_mode_high"])
            # how to read `value` against BASELINE.json's north star: `value` is the bf16-storage mode that
            # configs[1] names — its own image error is 1.2e-2 — and the throughput at which "parity within 1e-3" HOLDS is the figure below
            result["north_star"] = {
                "parity_tolerance": NORTH_STAR,
                "images_per_s_at_parity": pm.get("images_per_s") if pm.get("meets_north_star") else None,
                "ratio_to_value": round(pm["images_per_s"] / value, 3) if pm.get("meets_north_star") and value else None,
                "mode": f"precision=\"{pm.get('precision')}\" (parity_mode), same model / inputs / run",
                "value_mode_image_error": result.get("parity_vs_reference_golden", {}).get("rel_err_image_samples"),
                "images_per_s_at_parity_outlier_weights": pr.get("images_per_s") if pr.get("meets_north_star") else None,
                "ratio_to_value_outlier_weights": round(pr["images_per_s"] / value, 3) if pr.get("meets_north_star") and value else None,
                "mode_outlier_weights": f"precision=\"{pr.get('precision')}\"",
                "scope": "hybrid holds 1e-3 on well-conditioned weights (synthetic smooth set, tests/golden/full_det512.npz: the parity figures on this "
                         "line).  With outlier channels in the weights (moderate set: tests/golden/moderate.npz, the reference's outputs) hybrid / mixed "
                         "leave 1.6 - 3.4e-3 and the modes that hold 1e-3 on EVERY figure are precision=\"robust\" (6.3 - 7.8e-4) and \"high\" (<= 1.3e-4), "
                         "both asserted against 1e-3 itself in tests/test_gpu_heavy.py; their throughput on this workload is timed above "
                         "(profiles/r06/moderate_policies*.log)"}
        except Exception as e:       # never take the headline down
            result["parity_mode"] = {"error": repr(e)}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio; flush it first so that the JSON line is the LAST line on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(result), flush=True)


def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n: int) -> int:
    """Start `n` fresh child processes of this script, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on
    127.0.0.1), and forward rank 0's JSON line.  The parent never touches the GPU (a process that has initialised HIP must not be
    replaced or forked on this pool) and never re-execs.  All children are POLLED together: as soon as one exits non-zero the
    others are killed (a rank that dies before or inside the rendezvous / a collective would otherwise leave its peers waiting
    for their multi-minute timeouts) and that code is returned; the launcher also exits non-zero if the line does not report `n`
    ranks.  Rank 0's stdout goes to a temporary file, so a full pipe can never stall it."""
    import subprocess
    import tempfile
    port = _free_port()
    procs = []
    with tempfile.TemporaryFile(mode="w+") as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else sys.stderr))
        failed = None
        while failed is None and any(p.poll() is None for p in procs):
            for r, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    failed = (r, p.returncode)
                    break
            else:
                time.sleep(0.2)
        if failed is None:
            failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0), None)
        if failed is not None:
            for p in procs:                      # fail fast: the survivors would sit in the rendezvous / a collective
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            out0.seek(0)
            log(f"bench.py launcher: rank {failed[0]} exited with code {failed[1]}; the other ranks were stopped; rank 0 printed "
                f"(NOT forwarded as a result):\n{out0.read()}")
            return failed[1] or 1
        out0.seek(0)
        text = out0.read()
    lines = [ln for ln in text.splitlines() if ln.strip()]
    try:
        line = json.loads(lines[-1])
    except (IndexError, ValueError):
        log("bench.py launcher: rank 0 printed no JSON line")
        return 1
    if line.get("rccl_ranks") != n or line.get("n_gpus") != n:
        log(f"bench.py launcher: asked for {n} ranks, the result line reports rccl_ranks={line.get('rccl_ranks')} n_gpus={line.get('n_gpus')}")
        return 1
    print(lines[-1], flush=True)
    return 0


def count_ranks(dist, dev) -> int:
    """Ranks that actually take part in a collective (all-reduce of ones over the process group)."""
    one = torch.ones(1, dtype=torch.int32, device=dev)
    dist.all_reduce(one)
    return int(one.item())


def dry_run_rank(rank: int, world: int) -> int:
    """Launcher self-test (tests/test_dist_cpu.py): rendezvous + one all-reduce, no model, no GPU needed with
    EDTR_BENCH_BACKEND=gloo.  Rank 0 prints a JSON line shaped like the real one."""
    import torch.distributed as dist
    backend = os.environ.get("EDTR_BENCH_BACKEND", "nccl")
    if os.environ.get("EDTR_BENCH_DRY_FAIL_EARLY_RANK") == str(rank):  # the self-test's failure injection BEFORE the rendezvous:
        return 4                                                       # the surviving ranks would wait in init_process_group
    if world > 1 or os.environ.get("EDTR_BENCH_DIST"):
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if backend == "nccl" else torch.device("cpu")
        ranks = count_ranks(dist, dev)
        dist.barrier()
        dist.destroy_process_group()
    else:
        ranks = 1
    if os.environ.get("EDTR_BENCH_DRY_FAIL_RANK") == str(rank):      # the self-test's failure injection
        return 3
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": ranks, "backend": backend}), flush=True)
    return 0


def parity_mode_leg(cldm, args, one_pass, capture_all, run_steps, B, rel_err, mode="mixed", steps=None) -> dict:
    """Switch the SAME model to a parity mode — "hybrid" (fp16 storage in the encoder and the denoiser, the decoder on the mixed mode's
    fp32 stream), "mixed" (fp32 stream, fp16 operands, 1-3 products per layer class everywhere) or "high" (bf16 split-3 products
    everywhere: the mode that holds 1e-3 on outlier-bearing weights) — rebuild its programs, time `--parity-steps` passes the way the
    headline was timed, and check the result against the reference golden.  The previous mode's engines are dropped."""
    t0 = time.time()
    steps = steps or args.parity_steps
    cldm.precision = cldm.unet.precision = cldm.controlnet.precision = mode
    for sl_ in range(args.inflight):
        cldm.engine_slot = sl_
        one_pass()
    torch.cuda.synchronize()
    capture_all()
    run_steps(args.inflight)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    t0 = time.perf_counter()
    run_steps(steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    cldm.engine_slot = 0
    img, z = one_pass(inject=True)         # untimed parity pass: the golden's noise tensors
    torch.cuda.synchronize()
    pol = cldm._policy()
    out = {"precision": mode, "policy": pol.describe() if pol is not None else None, "images_per_s": round(B / ms * 1e3, 3), "ms_per_step": round(ms, 3),
           "steps": steps, "build_seconds": round(build_s, 1),
           "mfma_frac_whole_path": round(B / ms * 1e3 * FLOP_PER_IMAGE / (PEAK_TFLOPS * 1e12), 4),
           "note": "same model, same inputs, same run as the headline; fp32 activation stream + multi-part fp16 products "
                   "(algorithmic FLOP unchanged: the extra products are precision overhead, not counted)"}
    if mode == "hybrid":
        from edtr_amd.model.cldm import hybrid_sections
        out["sections"] = hybrid_sections()
    gp = golden_parity(args.workload, img, z, rel_err, mode).get("parity_vs_reference_golden")
    if gp:
        out.update(rel_err_latent=gp["rel_err_latent"], rel_err_image=gp["rel_err_image_samples"], fixture=gp["fixture"],
                   max_err_latent=gp["max_err_latent"], max_err_image=gp["max_err_image_samples"],
                   p9999_err_latent=gp["p9999_err_latent"], p9999_err_image=gp["p9999_err_image_samples"], max_norm_ok=gp["max_norm_ok"],
                   images=gp["images"], meets_north_star=bool(gp["rel_err_latent"] < NORTH_STAR and gp["rel_err_image_samples"] < NORTH_STAR), north_star=NORTH_STAR)
    return out


def other_workload_leg(cldm, diffusion, sampler, name, dev, ctx_dim, args, capture_all, rel_err) -> dict:
    """A short measurement of another BASELINE configuration in the headline's process (same model, fast mode): configs[4] per GPU
    (det512s50: batch 4, 50 steps from pure noise, fresh torch.randn_like per step) or configs[3] (seg1024tiled: one 1024x1024
    image, tiled VAE encoder + latent-tiled sampler).  Timed like the headline (hipGraph replay, barrier-to-barrier wall clock);
    `python bench.py --workload <name>` is the full-length form."""
    from edtr_amd import workloads
    B, S, _ = workloads.WORKLOADS[name]
    inflight = args.inflight
    steps = 16 if name == "seg1024tiled" else 6      # (the last pass of a leg runs without a partner in flight: short legs under-read by half a pass)
    inp = workloads.make_inputs(name, ctx_dim, dev, B, S, with_step_noises=(name == "det512s50"))     # (the parity pass injects them)
    untiled_forward = type(cldm).forward.__get__(cldm)      # (the tiled sampler monkey-patches cldm.forward and never restores it)

    def one_pass(inject=False):
        img, z, _ = workloads.restore_pass(cldm, diffusion, sampler, inp, name, untiled_forward, inject=inject)
        return img, z

    streams = [torch.cuda.Stream() for _ in range(inflight)]

    def run(n):
        out = None
        for i in range(n):
            k = i % inflight
            cldm.engine_slot = k
            with torch.cuda.stream(streams[k]):
                out = one_pass()
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        return out

    try:
        for k in range(inflight):          # program build per buffer slot
            cldm.engine_slot = k
            one_pass()
        torch.cuda.synchronize()
        capture_all()
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        run(inflight)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        cldm.engine_slot = 0
        img, z = one_pass(inject=True)     # untimed parity pass
        torch.cuda.synchronize()
    finally:                               # whatever happened: the legs after this one must see the untiled forward again
        cldm.forward = untiled_forward
        cldm.engine_slot = 0
    out = {"images_per_s": round(B / ms * 1e3, 4), "ms_per_step": round(ms, 3), "steps": steps, "batch": B, "image_size": S,
           "denoise_steps": workloads.WORKLOADS[name][2], "batches_in_flight": inflight,
           "mfma_frac_whole_path": round(B / ms * 1e3 * FLOP_PER_IMAGE_BY_WORKLOAD[name] / (PEAK_TFLOPS * 1e12), 4)}
    gp = golden_parity(name, img, z, rel_err, args.dtype).get("parity_vs_reference_golden")
    if gp:
        out.update(rel_err_latent=gp["rel_err_latent"], rel_err_image=gp["rel_err_image_samples"], max_err_latent=gp["max_err_latent"],
                   max_err_image=gp["max_err_image_samples"], parity_ok=gp["ok"])
    return out


def swinir_leg(dev, dtype, B, S, steps) -> dict:
    """SwinIR (configs/det/demo.yaml:2-18) on a batch of synthetic low-quality images: hipGraph replay, wall clock over
    `steps` forwards.  Not part of `value`."""
    from edtr_amd import synth
    from edtr_amd.model.swinir import SwinIR
    m = SwinIR(**synth.swinir_config())
    sd = m.state_dict()
    m.load_state_dict({k: (synth.synth_param("swinirfull." + k, tuple(v.shape)) if v.dtype.is_floating_point and not k.endswith("attn_mask") else v)
                       for k, v in sd.items()}, strict=True)
    m = m.eval().to(dev)
    m.compute_dtype = dtype
    x = synth.synth_input("bench:lq", (B, 3, S, S), 0.0, 1.0).to(dev)
    m(x)
    eng = next(iter(m._engines.values()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.prog.run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"network": "SwinIR 8x6 layers, 180 channels, window 8 (pixel-unshuffle 8, nearest+conv)", "ms_per_batch": round(ms, 3),
            "images_per_s": round(B / ms * 1e3, 2), "tflops": round(eng.prog.total_flops() / ms / 1e9, 1), "launches": len(eng.prog.recs)}


def roofline_pass(cldm, args, ms_per_step=None) -> dict:
    """Per-launch HIP-event timing of every program once (eager replay on the launch stream), aggregated by kernel."""
    agg = {}
    shapes = {}
    progs = []
    for key, e in cldm._cldm_engines.items():
        if key[-1] == 0:            # one buffer slot is enough: the slots run identical programs
            progs.append((e.step_prog, 50 if args.workload == "det512s50" else 4))
    for key, e in cldm._vae_engines.items():
        if key[-1] == 0:
            progs.append((e.prog, 1))
    for prog, mult in progs:
        g, prog.graph = prog.graph, None
        prog.run_timed()                      # warm
        rows = prog.run_timed()
        prog.graph = g
        for name, ms, flops, nbytes, tag in rows:
            kind = kernel_of(name)
            # EXECUTED multiply-adds: the sub-pixel form of the upsample convolutions runs 4 of the 9 algorithmic taps (edtr_hip.h,
            # upsample2x == 2); everything else executes what it is credited with (per-shape lines above the peak)
            fx = flops * (4.0 / 9.0) if " up2sp" in tag else flops
            if tag:
                sh = shapes.setdefault(tag, [0.0, 0.0, 0, 0.0])
                sh[0] += ms * mult
                sh[1] += flops * mult
                sh[2] += mult
                sh[3] += fx * mult
            a = agg.setdefault(kind, {"ms": 0.0, "flops": 0.0, "flops_exec": 0.0, "bytes": 0.0, "n": 0, "by_name": {}})
            a["ms"] += ms * mult
            a["flops"] += flops * mult
            a["flops_exec"] += fx * mult
            a["bytes"] += nbytes * mult
            a["n"] += mult
            if " sk" in tag:
                a["n_splitk"] = a.get("n_splitk", 0) + mult
            bn = a["by_name"].setdefault(name, [0.0, 0.0, 0])
            bn[0] += ms * mult
            bn[1] += fx * mult            # (executed)
            bn[2] += mult
    total_ms = sum(a["ms"] for a in agg.values())
    if args.breakdown:
        log(f"--- per-kernel breakdown of one pass (sum of launch durations {total_ms:.2f} ms) ---")
        for kind, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            tf = a["flops"] / (a["ms"] * 1e-3) / 1e12 if a["ms"] > 0 else 0.0
            log(f"{kind:28s} {a['ms']:9.3f} ms  {100 * a['ms'] / total_ms:5.1f}%  n={a['n']:5d}  {tf:8.1f} TFLOP/s")
            for name, (ms, fl, n) in sorted(a["by_name"].items(), key=lambda kv: -kv[1][0])[:12]:
                tfn = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                log(f"    {name:28s} {ms:9.3f} ms  n={n:5d}  {tfn:8.1f} TFLOP/s")
    if args.breakdown:
        log("--- igemm launches by shape (top 45 by time) ---")
        for tag, (ms, fl, n, fx) in sorted(shapes.items(), key=lambda kv: -kv[1][0])[:45]:
            log(f"    {tag:58s} {ms:8.3f} ms  n={n:4d}  {ms / n * 1e3:8.1f} us  {fx / (ms * 1e-3) / 1e12:7.1f} TFLOP/s executed"
                + (f"  ({fl / (ms * 1e-3) / 1e12:7.1f} algorithmic: 9 taps credited, 4 run)" if fx != fl else ""))
    ig = agg.get("igemm_kernel")
    at = agg.get("flash_attn64_kernel")
    out = {}

    def pmc_family(names):
        """(bytes per pass, note) of kernel families in the committed PMC passes, None when stale or absent"""
        try:
            from edtr_amd.build import source_hash
            with open(os.path.join(ROOT, PMC_TRAFFIC_FILE)) as f:
                pm = json.load(f)
            if pm.get("kernel_source_hash") != source_hash():
                return None, f"{PMC_TRAFFIC_FILE} was measured on kernel sources {pm.get('kernel_source_hash')}, this build is {source_hash()}: stale, not reported"
            fams = [pm["families"][n] for n in names if n in pm["families"]]
            if not fams:
                return None, None
            return (sum(f["hbm_side_bytes_per_pass"] for f in fams), sum(f["launches_per_pass"] for f in fams)), PMC_TRAFFIC_FILE
        except (OSError, KeyError, ValueError):
            return None, None

    if ig:
        ach = ig["flops"] / (ig["ms"] * 1e-3) / 1e12
        ach_x = ig["flops_exec"] / (ig["ms"] * 1e-3) / 1e12
        # HBM-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE cannot be read live); the
        # entry is used only if it was measured on a launch list of the same length as the one just timed
        traffic, traffic_src = None, None
        try:
            from edtr_amd.build import source_hash
            with open(os.path.join(ROOT, PMC_TRAFFIC_FILE)) as f:
                pm = json.load(f)
            if pm.get("kernel_source_hash") != source_hash():
                raise ValueError(f"{PMC_TRAFFIC_FILE} was measured on kernel sources {pm.get('kernel_source_hash')}, this build is "
                                 f"{source_hash()}: stale, not reported")
            fam = pm["families"]["igemm"]
            # the profiler counts kernels (main loops + split-K reducers), the timing above counts edtr_igemm calls: compare like
            # with like, then quote the bytes per edtr_igemm CALL, the unit of algorithmic_bytes_per_launch
            kernels = ig["n"] + ig.get("n_splitk", 0)
            if abs(fam["launches_per_pass"] - kernels) <= 0.02 * kernels:
                traffic = round(fam["hbm_side_bytes_per_pass"] / ig["n"])
                traffic_src = (f"{PMC_TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, fetch x2 gfx950 "
                               f"correction; kernel sources {pm.get('kernel_source_hash')}: {fam['launches_per_pass']:.0f} kernels per pass = "
                               f"{ig['n']} edtr_igemm calls + {ig.get('n_splitk', 0)} split-K reducers; bytes per call)")
            else:
                traffic_src = (f"{PMC_TRAFFIC_FILE} was measured on {fam['launches_per_pass']:.0f} kernels per pass, this run "
                               f"has {kernels}: stale, not reported")
        except (OSError, KeyError) :
            pass
        except ValueError as e:
            traffic_src = str(e)
        out["roofline"] = {"kernel": "edtr_igemm family (implicit-GEMM conv / linear, MFMA 32x32x16 and 16x16x32 tiles; incl. the fused feed-forward launch edtr_ffn and the row-resident K = 320 projections edtr_lin320)", "bound": "mfma",
                           "achieved": round(ach, 2), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_TFLOPS, 4),
                           "achieved_executed": round(ach_x, 2), "frac_executed": round(ach_x / PEAK_TFLOPS, 4),
                           "frac_of_sustained_mfma_rate": round(ach / SUSTAINED_TFLOPS, 4),
                           "sustained_mfma_rate": {"tflops": SUSTAINED_TFLOPS, "clock_ghz": 1.69,
                                                   "source": "profiles/r06/attn_issue_bound.log (bare MFMA stream, random bf16 operands, 256 workgroups x 4 waves)"},
                           "frac_note": "frac = ALGORITHMIC FLOP (2 M N K of the reference's operation) / time; frac_executed = the multiply-adds "
                                        "the kernels actually run / time: the sub-pixel upsample convolutions run 4 of their 9 algorithmic taps",
                           "traffic": traffic,
                           "algorithmic_bytes_per_launch": round(ig["bytes"] / ig["n"]),
                           "traffic_source": traffic_src,
                           "launches_per_pass": ig["n"], "avg_launch_ms": round(ig["ms"] / ig["n"], 4),
                           "share_of_pass": round(ig["ms"] / total_ms, 3),
                           "how": "achieved = sum of algorithmic FLOP / sum of per-launch durations, every launch alone on the launch "
                                  "stream between two HIP events in an eager replay AFTER the timed region (isolated-launch figure); "
                                  "inside the timed region two lanes and two batches overlap, so the family's share of the wall is "
                                  "smaller than ms_per_pass_isolated",
                           "ms_per_pass_isolated": round(ig["ms"], 3)}
        n_gnin = sum(v[2] for tag, v in shapes.items() if " gnin" in tag)
        if n_gnin:
            out["roofline"]["fused_groupnorm_inputs"] = {
                "launches_per_pass": n_gnin,
                "note": "these convolutions apply the GroupNorm + SiLU of their input while staging it (edtr_igemm a_gn): their duration "
                        "includes that arithmetic (the edtr_gn_apply launch in front of each is gone), their FLOP count does not, so "
                        "`frac` reads ~0.02 lower than with EDTR_GN_IN_CONV=0 while the path is 1.4 % faster (profiles/r04/gn_in_conv_ab.log)"}
        if ms_per_step:
            # self-consistent in-situ bound: the family's FLOP of one pass over the WALL time of one pass of the timed region
            # (as if nothing else ran): what the timed region certainly sustained, <= the isolated figure by construction
            lo = ig["flops"] / (ms_per_step * 1e-3) / 1e12
            out["roofline"]["in_timed_region"] = {"achieved_lower_bound": round(lo, 2), "frac_lower_bound": round(lo / PEAK_TFLOPS, 4),
                                                  "wall_ms_per_pass": round(ms_per_step, 3)}
    if at:
        ach = at["flops"] / (at["ms"] * 1e-3) / 1e12
        tr, tr_src = pmc_family(["flash_attn_smallk", "flash_attn_v1", "flash_attn_v3", "flash_attn_d512"])
        at_traffic = round(tr[0] / at["n"]) if tr and abs(tr[1] - at["n"]) <= 0.02 * at["n"] else None
        out["roofline_attention"] = {"kernel": "flash_attn64_kernel", "bound": "mfma", "achieved": round(ach, 2),
                                     "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS, 4),
                                     "frac_of_sustained_mfma_rate": round(ach / SUSTAINED_TFLOPS, 4),
                                     "issue_bound_ceiling": {"tflops": 1283.5, "pipe_occupancy": 0.78,
                                                             "source": "profiles/r06/attn_issue_bound.log: the tile loop's instruction mix (32 MFMA + 64 v_exp + 64 v_add + "
                                                                       "32 v_cvt_pk per 64-key tile) as a register-only stream, one wave per SIMD"},
                                     "achieved_executed": round(ach, 2), "frac_executed": round(ach / PEAK_TFLOPS, 4),
                                     "frac_note": "algorithmic = executed for attention up to key padding (the 77-key cross-attention multiplies 80 - 128 keys); "
                                                  "4 B H Nq Nk d FLOP per call",
                                     "traffic": at_traffic, "algorithmic_bytes_per_launch": round(at["bytes"] / at["n"]), "traffic_source": tr_src,
                                     "launches_per_pass": at["n"], "share_of_pass": round(at["ms"] / total_ms, 3)}
    out["kernel_time_ms_per_pass"] = {k: round(a["ms"], 3) for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    out["launches_per_pass"] = int(sum(a["n"] for a in agg.values()))
    if args.breakdown_json:
        table = {}
        for kind, a in agg.items():
            for name, (ms, fl, n) in a["by_name"].items():
                table[name] = {"kernel": kind, "ms": round(ms, 4), "flops": fl, "n": n}
        os.makedirs(os.path.dirname(os.path.abspath(args.breakdown_json)), exist_ok=True)
        with open(args.breakdown_json, "w") as f:
            json.dump({"by_name": table, "by_shape": {t: {"ms": round(v[0], 4), "flops": v[1], "n": v[2], "flops_executed": v[3]} for t, v in shapes.items()},
                       "total_ms": total_ms, "precision": args.precision, "workload": args.workload}, f, indent=1)
    return out


def kernel_of(name: str) -> str:
    if name.startswith("flash") or name.endswith(".flash"):        # (the d = 64 kernels and the VAE's d = 512 kernel: one attention family)
        return "flash_attn64_kernel"
    if name.endswith(".stats"):
        return "gn_stats_kernel"
    if name.endswith(".finalize") or name.endswith(".table"):     # (gn.table: the finalize that writes a (scale, shift) table for a fused GroupNorm input)
        return "gn_finalize_kernel"
    if name.startswith("gn_pool") or name.startswith("copy3d") or name.startswith("wavelet"):
        return name.split(".")[0] + "_kernel"
    if name.endswith(".apply"):
        return "gn_apply_kernel"
    for k in ("layernorm", "softmax_rows", "nchw_to_nhwc", "nhwc_to_nchw", "add", "timestep_embedding", "cast16",
              "sampler_update", "axpby"):
        if name.startswith(k):
            return k + "_kernel"
    return "igemm_kernel"


# relative-L2 tolerances against the fp32 reference at full size, each <= 1.5 x its measured value (bf16 6.2e-3 /
# 1.2e-2, fp16 8.6e-4 / 1.5e-3, mixed 5.4e-4 / 6.0e-4, high — with split attention operands, round 4 — 2.9e-5 / 1.2e-5 / 2.4e-5
# for z_pre / latent / image over the three workloads) so that a 2x regression of a mode's numerics fails the run; the north-star
# 1e-3 is what the parity modes (mixed, high) must additionally meet
TOLERANCE = {"bf16": {"z_pre": 1.75e-2, "latent": 9.3e-3, "image": 1.8e-2}, "fp16": {"z_pre": 2.2e-3, "latent": 1.3e-3, "image": 2.3e-3},
             "mixed": {"z_pre": 1e-3, "latent": 8.2e-4, "image": 9e-4}, "hybrid": {"z_pre": 2.2e-3, "latent": 1e-3, "image": 1e-3},
             "robust": {"z_pre": 1e-3, "latent": 3.8e-4, "image": 5.7e-4}, "high": {"z_pre": 4.4e-5, "latent": 1.8e-5, "image": 3.6e-5}}
NORTH_STAR = 1e-3
# max-norm bound as a multiple of the L2 tolerance: measured max / L2 ratios of the shipped modes are 0.9 - 1.3 (the peak of a latent / image is a few times its RMS)
# (profiles/r04/maxnorm_measured.log), a defect in one 16 x 16 tile of a 512 x 512 image with O(1) errors gives > 100
MAX_OVER_L2 = 3.0
PMC_TRAFFIC_FILE = os.path.join("profiles", "r06", "pmc_hbm_traffic.json")


def _cpu_worker(q_in, q_out, cfg_name, S, threads):
    """Child process: the oracle (CPU fp32 restatement pinned to the reference) on samples of the bench batch."""
    import torch as th
    th.set_num_threads(min(threads, 32))     # weight hashing is elementwise int64: more threads only add contention
    from edtr_amd import synth as sy
    from edtr_amd.testing import synthetic_state_dicts
    from oracle import edtr_oracle as O
    from oracle import flat_sd as flat_oracle_sd
    cfg = sy.CONFIGS[cfg_name]()
    sd = flat_oracle_sd(synthetic_state_dicts(cfg))
    th.set_num_threads(threads)
    q_out.put(("ready", None))
    while True:
        job = q_in.get()          # numpy arrays (pickled by value, no shared-memory handles); None = stop
        if job is None:
            return
        pre_res, c_txt, noises = job
        pre_res, c_txt, noises = th.from_numpy(pre_res), th.from_numpy(c_txt), [th.from_numpy(n) for n in noises]
        with th.no_grad():
            t0 = time.perf_counter()
            img, tr = O.restore(sd, cfg, O.make_betas(), pre_res, c_txt, noises, USED_TIMESTEPS, 200, return_trace=True)
            dt = time.perf_counter() - t0
        q_out.put(("done", (dt, img.numpy(), tr["z"].numpy())))


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def start_cpu_baseline(cfg_name, S):
    """Spawn the oracle process early (before this process touches the GPU): it builds its fp32 weights while the GPU side
    builds its own, then idles until the timed GPU region is over (so neither measurement perturbs the other)."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    # BASELINE.md §3 asks for os.cpu_count() threads; on the GPU box's 2 x 64-core EPYC 9575F (256 hardware threads) the fp32
    # oracle gets SLOWER beyond 32 threads (18.3 s at 32, 29.9 s at 64, 50.7 s at 128 for one image:
    # profiles/r02/cpu_oracle_threads.log), so the baseline runs at the thread count that is fastest for it, min(cores, 32);
    # EDTR_CPU_THREADS overrides
    threads = int(os.environ.get("EDTR_CPU_THREADS", str(min(cores, 32))))
    ctx = mp.get_context("spawn")
    q_in, q_out = ctx.Queue(), ctx.Queue()
    proc = ctx.Process(target=_cpu_worker, args=(q_in, q_out, cfg_name, S, threads), daemon=True)
    proc.start()
    return [proc, q_in, q_out, threads, cores, {"ready": False}]


def wait_cpu_ready(handle, budget_s=600.0) -> None:
    """Block until the oracle child has built its weights (so it is idle during the GPU warm-up and the timed region)."""
    proc, q_in, q_out, threads, cores, state = handle
    try:
        q_out.get(timeout=budget_s)
        state["ready"] = True
    except Exception:
        log("cpu baseline child did not become ready in time")


def finish_cpu_baseline(handle, inp, img, z, S, rel_err, dtype_name, budget_s=420.0) -> dict:
    """CPU points of BASELINE.md §3 on a bounded sample: the FIRST and the LAST image of the batch as two B=1 runs (both also
    give the live GPU-vs-oracle parity), then — if the B=1 runs were fast enough to leave room in the budget — the whole
    batch as one B=8 run."""
    proc, q_in, q_out, threads, cores, state = handle
    B = inp.pre_res.shape[0]
    pre, ctx, noises = inp.pre_res.cpu(), inp.c_txt[:1].cpu(), [n.cpu() for n in inp.noises]
    runs = {}
    t_begin = time.perf_counter()

    def job(sel):
        q_in.put((pre[sel].numpy().copy(), ctx.numpy().copy(), [n[sel].numpy().copy() for n in noises]))
        _, res = q_out.get(timeout=max(10.0, budget_s - (time.perf_counter() - t_begin)))
        return res

    try:
        if not state["ready"]:
            q_out.get(timeout=budget_s)                   # "ready"
        runs["first"] = (slice(0, 1), job(slice(0, 1)))
        if B > 1:
            runs["last"] = (slice(B - 1, B), job(slice(B - 1, B)))
            # the whole batch as ONE oracle call (BASELINE.md §3's B = 8 point) takes minutes: opt-in (EDTR_CPU_B8=1; the record
            # is profiles/r02/cpu_oracle_threads.log), the default run stays within ~40 s of CPU work
            if os.environ.get("EDTR_CPU_B8") == "1" and runs["first"][1][0] * B * 0.6 < budget_s - (time.perf_counter() - t_begin):
                runs["batch"] = (slice(0, B), job(slice(0, B)))
        q_in.put(None)
    except Exception:
        pass
    finally:
        if proc.is_alive():
            proc.kill()
        proc.join(timeout=10)
    cpu = _cpu_model()
    if "first" not in runs:
        return {"cpu_baseline": {"value": None, "unit": "images/s", "cores": threads, "kind": "port",
                                 "sample": f"1 image {S}x{S}, 4 steps: exceeded the {budget_s:.0f}s budget ({cpu})"}}
    tol = TOLERANCE[dtype_name]
    parity, ok = {}, True
    for name in ("first", "last"):
        if name in runs:
            sel, (dt, ref_img, ref_z) = runs[name]
            from edtr_amd.testing import err_stats
            sz, si = err_stats(z[sel], ref_z), err_stats(img[sel], ref_img)
            ez, ei = sz["l2"], si["l2"]
            good = bool(ez == ez and ei == ei and ez < tol["latent"] and ei < tol["image"]           # NaN-safe
                        and sz["max"] < MAX_OVER_L2 * tol["latent"] and si["max"] < MAX_OVER_L2 * tol["image"])
            ok = ok and good
            parity[f"{name}_image"] = {"index": sel.start, "rel_err_latent_vs_oracle": float(f"{ez:.3e}"),
                                       "rel_err_image_vs_oracle": float(f"{ei:.3e}"), "max_err_latent_vs_oracle": float(f"{sz['max']:.3e}"),
                                       "max_err_image_vs_oracle": float(f"{si['max']:.3e}"), "ok": good}
    if "batch" in runs:
        sel, (dt8, ref_img, ref_z) = runs["batch"]
        worst = max(max(rel_err(z[k:k + 1], ref_z[k:k + 1]), rel_err(img[k:k + 1], ref_img[k:k + 1])) for k in range(B))
        parity["whole_batch_worst_image_rel_err"] = float(f"{worst:.3e}")
        ok = ok and bool(worst == worst and worst < tol["image"])
    if not ok:
        log(f"!!! PARITY FAILURE: GPU vs CPU oracle {parity} (tolerance {tol}) — the throughput above is NOT a valid result")
    dt1 = runs["first"][1][0]
    b1 = 1.0 / dt1
    out = {"value": round(b1, 5), "unit": "images/s", "cores": threads, "threads_all": cores, "kind": "port", "cpu_model": cpu,
           "sample": f"B=1: image 0 of the batch, {S}x{S}, 4 steps, fp32 oracle on torch CPU kernels, {threads} threads of "
                     f"{cores} host cores ({dt1:.1f} s)", "b1_images_per_s": round(b1, 5)}
    if "batch" in runs:
        dt8 = runs["batch"][1][0]
        out["b8_images_per_s"] = round(B / dt8, 5)
        out["b8_measured_in_run"] = True
        out["value"] = round(max(b1, B / dt8), 5)
        out["sample"] += f"; B={B}: the whole batch in one oracle call ({dt8:.1f} s); value = the faster of the two"
    else:
        # SURVEY §8(d) asks for B = 1 and B = 8: the B = 8 point is the COMMITTED measurement (not re-measured in this run, and marked so)
        out["b8_images_per_s"] = 0.0794
        out["b8_measured_in_run"] = False
        out["b8_note"] = (f"BASELINE.md §3's B={B} point was measured ONCE on this host class and committed (profiles/r04/cpu_oracle_b8.log, "
                          "tests/cpu_oracle_b8.py): the bench batch as 8 concurrent B=1 oracle processes restores in 100.7 s at 8 x 16 threads "
                          "(0.079 images/s) and in 237 s at 8 x 32 threads (0.034) — the fp32 oracle is memory-bound on the host, so the batch "
                          "point is within 1.4 x of the B=1 figure above; it is not re-measured in every run (4 - 10 minutes of CPU work; "
                          "EDTR_CPU_B8=1 runs the whole batch as one oracle call)")
    return {"cpu_baseline": out, "parity": dict(parity, ok=ok, tolerance=tol, against="CPU oracle (fp32, pinned to the reference)")}


def golden_parity(workload, img, z, rel_err, dtype_name) -> dict:
    """Live check of the timed workload's result against the REFERENCE's own output on the same inputs
    (tests/golden/full_*.npz, tools/make_goldens.py gen_full): latents in full, images at stride-4 samples."""
    name = {"det512": "full_det512.npz", "seg1024tiled": "full_seg1024.npz", "det512s50": "full_s50.npz"}[workload]
    path = os.path.join(ROOT, "tests", "golden", name) if name else None
    if not path or not os.path.exists(path):
        return {}
    g = np.load(path)
    sel = [int(k) for k in g["images"]] if "images" in g.files else [0]
    if max(sel) >= z.shape[0]:
        return {}
    from edtr_amd.testing import err_stats
    zc, ic = z.cpu()[sel], img.cpu()[sel][:, :, 1::4, 2::4]
    sz, si = err_stats(zc, g["z"]), err_stats(ic, g["img_samples"].astype(np.float32))
    ez, ei = sz["l2"], si["l2"]
    tol = TOLERANCE[dtype_name]
    # max-norm gate (BASELINE.md §3 "max relative error"): worst element / the signal's peak must stay below MAX_OVER_L2 x the L2
    # tolerance — a localised defect (one wrong halo column, one bad tile seam) moves the max by orders of magnitude and the L2 norm hardly
    max_ok = bool(sz["max"] < MAX_OVER_L2 * tol["latent"] and si["max"] < MAX_OVER_L2 * tol["image"])
    ok = bool(ez == ez and ei == ei and ez < tol["latent"] and ei < tol["image"] and max_ok)
    if not ok:
        log(f"!!! PARITY FAILURE vs the reference golden {name}: latent {sz}, image {si} (tolerance {tol}, max-norm bound {MAX_OVER_L2} x)")
    r3 = lambda v: float(f"{v:.3e}")
    return {"parity_vs_reference_golden": {"fixture": f"tests/golden/{name}", "images": sel, "rel_err_latent": r3(ez),
                                           "rel_err_image_samples": r3(ei), "max_err_latent": r3(sz["max"]), "max_err_image_samples": r3(si["max"]),
                                           "p9999_err_latent": r3(sz["p9999"]), "p9999_err_image_samples": r3(si["p9999"]),
                                           "max_norm": "max|a - b| / max|b| (p9999: the 99.99th percentile of |a - b| / max|b|)",
                                           "max_norm_bound": {"latent": MAX_OVER_L2 * tol["latent"], "image": MAX_OVER_L2 * tol["image"]},
                                           "max_norm_ok": max_ok, "ok": ok, "tolerance": tol}}


if __name__ == "__main__":
    main()
