#!/usr/bin/env python
"""Headline benchmark: denoising-steps/sec, DeepCAD face-LDM, batch=512 per GPU (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch: SurfZNet eps-prediction on [512, 60, 48] latents
(+ face bboxes [512,60,6], key-padding mask, timestep) through bg_denoiser_fwd, the on-device draw of the
ancestral noise, and the fused DDPM update (bg_cfg_ddpm_step) -- exactly what sample.py:144-153 does per
iteration.  Synthetic inputs and random-init weights (no datasets / checkpoints offline), all resident in HBM
before the timed region.  bf16 operands, fp32 accumulation / residual / LayerNorm / softmax.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, the batch is sharded by rank (512 samples per rank: weak scaling), no data-path
collective; the finished latents are collected with ONE all_gather (RCCL) inside the timed region.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, N_FACE = 512, 60
MFMA_PEAK_TFLOPS = 2500.0        # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def algorithmic_flops_per_sample_eval(n_tok, c_io):
    """F(N) = N*(12*7,864,320 + C_io) + 36,864*N^2."""
    return n_tok * (12 * 7_864_320 + c_io) + 36_864 * n_tok * n_tok


def make_inputs(B, device, seed):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(B, N_FACE, 48, generator=g)
    pos = torch.randn(B, N_FACE, 6, generator=g).clamp(-3, 3)
    mask = torch.ones(B, N_FACE, dtype=torch.bool)
    nvalid = torch.randint(8, N_FACE + 1, (B,), generator=g)          # valid faces per sample ~ U{8..60}
    for b in range(B):
        mask[b, : int(nvalid[b])] = False
    return z.to(device), pos.to(device), mask.to(device)


CPU_SAMPLE_B = 128     # bounded sample of the workload for the CPU leg: a quarter of the batch


def cpu_baseline(steps=2, warmup=1):
    """The CPU oracle (a 'port': plain-math restatement of the reference, fp32, torch CPU threads) timed on a
    BOUNDED SAMPLE of the same workload: the first 128 of the 512 samples (60 tokens each), `steps` denoising steps
    after `warmup`.  Every op of the path is per-sample, so a full 512-batch step costs 4x the sample's time; `value`
    is reported in the bench's unit (steps/s at batch 512) with that factor applied."""
    from oracle import denoisers as orc
    from oracle.schedulers import OracleDDPM
    sd = orc.seeded_state_dict("SurfZNet", 0)
    z, pos, mask = make_inputs(B_PER_GPU, "cpu", 1234)
    z, pos, mask = z[:CPU_SAMPLE_B], pos[:CPU_SAMPLE_B], mask[:CPU_SAMPLE_B]
    sch = OracleDDPM(clip_sample=True, clip_sample_range=3)
    sch.set_timesteps(1000)
    ts = sch.timesteps[-250:]
    g = torch.Generator().manual_seed(7)
    times = []
    with torch.no_grad():
        for i in range(warmup + steps):
            t0 = time.perf_counter()
            t = ts[i]
            eps = orc.surfz_forward(sd, z, t.reshape(-1), pos, mask)
            z = sch.step(eps, t, z, noise=torch.randn(z.shape, generator=g))
            if i >= warmup:
                times.append(time.perf_counter() - t0)
    per = sum(times) / len(times) * (B_PER_GPU / CPU_SAMPLE_B)
    return {"value": round(1.0 / per, 4), "unit": "denoising-steps/s (batch=512)", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{CPU_SAMPLE_B} of the 512 samples x 60 tokens, {steps} timed steps after {warmup} "
            f"warm-up (x{B_PER_GPU // CPU_SAMPLE_B} to a full batch: the path is per-sample), oracle/denoisers.py + "
            "oracle/schedulers.py fp32 on torch CPU threads", "s_per_step_batch512": round(per, 3)}


def pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
    written by tools/pmc_summary.py: (2 x FETCH_SIZE + WRITE_SIZE) KiB averaged over that kernel's launches -- the x2
    is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md).  None when no measurement is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel.split("(")[0], {}).get("bytes_per_launch")
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import brepgen_amd as bga
    from brepgen_amd import _lib
    from brepgen_amd.sampling import gather_latents

    torch.manual_seed(0)
    net = bga.SurfZNet(False).to(dev).eval()
    net.compute_dtype = torch.bfloat16
    net.cache_conditioning = False     # every step recomputes p_embed(surfPos) like the reference does (network.py:1182)
    z, pos, mask = make_inputs(B_PER_GPU, dev, 1234 + rank)
    sch = bga.DDPMScheduler(num_train_timesteps=1000, beta_schedule="linear", prediction_type="epsilon",
                            beta_start=0.0001, beta_end=0.02, clip_sample=True, clip_sample_range=3)
    sch.set_timesteps(1000)
    ts_cpu = sch.timesteps[-250:]
    ts_dev = ts_cpu.to(dev)

    def run_steps(k, x, offset=0):
        with torch.no_grad():
            for i in range(k):
                j = (offset + i) % 250
                eps = net(x, ts_dev[j:j + 1], pos, mask, None)
                noise = torch.randn_like(x)                          # upstream draws it on the device, per step
                x = sch.step(eps, ts_cpu[j], x, noise=noise).prev_sample
        return x

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    x = run_steps(args.warmup, z)
    if dist is not None:
        gather_latents({"surfZ": x}, dist)                           # warm the communicator up outside the clock
    barrier()
    t0 = time.perf_counter()
    x = run_steps(args.steps, x, args.warmup)
    out = gather_latents({"surfZ": x}, dist) if dist is not None else {"surfZ": x}
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    finite = bool(torch.isfinite(out["surfZ"]).all())

    roofline = None
    breakdown = None
    if not args.no_roofline:
        # second pass of the same K steps with a hipEvent pair around every kernel launch (on the launch stream)
        with _lib.profile() as prof:
            run_steps(args.steps, x, args.warmup)
        rows = {r["kernel"]: r for r in prof.rows}
        breakdown = {k: {"launches": r["launches"], "avg_us": round(1e3 * r["total_ms"] / r["launches"], 2),
                         "total_ms_per_step": round(r["total_ms"] / args.steps, 4),
                         "tflops": round(r["flops"] / r["total_ms"] / 1e9, 1) if r["flops"] else None,
                         "gbs": round(r["bytes"] / r["total_ms"] / 1e6, 1)} for k, r in rows.items()}
        dom = max(rows.values(), key=lambda r: r["total_ms"])
        ach = dom["flops"] / dom["total_ms"] / 1e9                    # TFLOP/s = flops per launch / avg duration
        roofline = {"kernel": dom["kernel"], "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic(dom["kernel"]),
                    "launches_per_step": dom["launches"] // args.steps,
                    "avg_launch_us": round(1e3 * dom["total_ms"] / dom["launches"], 2),
                    "flops_per_launch": dom["flops"] / dom["launches"]}

    if rank == 0:
        steps_per_s = world * args.steps / elapsed
        f_step = B_PER_GPU * algorithmic_flops_per_sample_eval(N_FACE, 3.70e6)
        line = {
            "metric": "denoising-steps/sec (whole node), DeepCAD face-LDM, batch=512",
            "value": round(steps_per_s, 3), "unit": "denoising-steps/s (batch=512 per step)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: DeepCAD face-LDM, SurfZNet eps-eval [512,60,48]+bbox+mask "
                                   "+ DDPM update (timesteps[-250:], clip 3), bf16 operands / fp32 accumulate, per GPU",
                       "batch_per_gpu": B_PER_GPU, "tokens_per_sample": N_FACE,
                       "sample_steps_per_s": round(steps_per_s * B_PER_GPU, 1),
                       "algorithmic_tflop_per_step": round(f_step / 1e12, 3),
                       "model_tflops_per_gpu": round(f_step * args.steps / elapsed / 1e12, 1),
                       "finite": finite, "parallelism": f"batch-sharded x{world}, 1 all_gather of latents",
                       "formulation": "norm1/norm2 folded into the QKV/FFN1 GEMM epilogues, residual stream as (hi, lo) "
                                      "16-bit planes, fused input embeds; conditioning cache off (every embed recomputed)"},
            "roofline": roofline, "kernels": breakdown,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
