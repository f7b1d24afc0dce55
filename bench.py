#!/usr/bin/env python
"""Headline benchmark: denoising-steps/sec, DeepCAD face-LDM, batch=512 per GPU (BASELINE.json configs[1]).

The face LDM of the reference is three loops (sample.py:126-202), and one "step" here is one iteration of them -- an
eps-prediction through bg_denoiser_fwd plus the scheduler update the reference runs in that loop:
  A  SurfPosNet on [512, 30, 6] boxes        + PNDM update (sample.py:128-137)      158 of the 617 iterations
  B  SurfPosNet on [512, 60, 6] boxes        + DDPM update, ancestral noise drawn on the device (sample.py:144-153)   250
  C  SurfZNet  on [512, 60, 48] latents + boxes + key-padding mask (valid faces ~ U{8..60}) + PNDM update (sample.py:191-202)   209
The K timed steps are split over the three loops in those proportions (A, B, C back to back inside ONE timed region), so
`value` is the step-weighted face-LDM rate; the per-loop rates are reported beside it (`extra.face_ldm_legs`).
Synthetic inputs and random-init weights (no datasets / checkpoints offline), all resident in HBM before the timed
region.  bf16 operands, fp32 accumulation / residual / LayerNorm / softmax.

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
# iterations of the three face-LDM loops per sample batch (eval_config.yaml: 200-step PNDM schedule cut at 158 evaluations,
# the last 250 of 1000 DDPM steps, the full 209-evaluation PNDM schedule), and the per-token I/O FLOPs of SURVEY 8(d)
LEGS = (("A", "SurfPosNet [512,30,6] + PNDM", 158), ("B", "SurfPosNet [512,60,6] + DDPM", 250), ("C", "SurfZNet [512,60,48]+bbox+mask + PNDM", 209))
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
    make_inputs.nvalid = nvalid                                       # host-side copy: FLOP accounting only
    return z.to(device), pos.to(device), mask.to(device)


CPU_SAMPLE_B = 128     # bounded sample of the workload for the CPU leg: a quarter of the batch


def _physical_cores():
    """Distinct (package, core) pairs of /proc/cpuinfo; falls back to the logical count."""
    try:
        pairs, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                pairs.add((phys, line.split(":")[1].strip()))
        return len(pairs) or os.cpu_count()
    except OSError:
        return os.cpu_count()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def split_steps(k):
    """K face-LDM iterations -> (kA, kB, kC) in the proportions 158 : 250 : 209 of the three loops (each at least 1 when K >= 3)."""
    tot = sum(w for _, _, w in LEGS)
    ka, kb = round(k * LEGS[0][2] / tot), round(k * LEGS[1][2] / tot)
    if k >= 3:
        ka, kb = max(1, ka), max(1, kb)
    kc = k - ka - kb
    if kc < 1 and k >= 3:
        kb, kc = kb - (1 - kc), 1
    return ka, kb, kc


def cpu_baseline(steps=2, warmup=1):
    """The reference's CPU PyTorch path, timed on this box's host cores on a BOUNDED SAMPLE of the same workload.

    What runs: `oracle/ref_formulation.py` -- the reference's own formulation of SurfPosNet / SurfZNet (stock
    nn.TransformerEncoder(norm_first, 12 x 768/12/1024) fed seq-first + Linear-LayerNorm-SiLU-Linear embeds, exactly the
    library modules network.py:1080-1200 composes; pinned to the reference's outputs by tests/test_oracle_golden.py), fp32,
    eval / no_grad, with the restated PNDM / DDPM updates -- the three loops of the headline, `steps` timed iterations each
    after `warmup`, combined with the same 158 : 250 : 209 weights.  /root/reference itself cannot travel to the GPU box,
    hence kind "port".  Sample: the first 128 of the 512 samples; every op of the path is per-sample, so a full 512-batch
    step costs 4x the sample's time and `value` is reported in the bench's unit with that factor applied.
    Threads: torch intra-op threads = PHYSICAL cores (SMT siblings only add contention to GEMM-bound work); a second
    setting (32) is calibrated on a warm-up step of loop C and the faster one is used -- both are reported."""
    from oracle import denoisers as orc
    from oracle import ref_formulation as rf
    from oracle.schedulers import OracleDDPM, OraclePNDM
    pos_net = rf.build("SurfPosNet", orc.seeded_state_dict("SurfPosNet", 0))
    z_net = rf.build("SurfZNet", orc.seeded_state_dict("SurfZNet", 0))
    z, pos, mask = make_inputs(B_PER_GPU, "cpu", 1234)
    z, pos, mask = z[:CPU_SAMPLE_B], pos[:CPU_SAMPLE_B], mask[:CPU_SAMPLE_B]
    g = torch.Generator().manual_seed(7)
    ddpm = OracleDDPM(clip_sample=True, clip_sample_range=3)
    ddpm.set_timesteps(1000)
    dts = ddpm.timesteps[-250:]

    def leg_a():
        sch = OraclePNDM()
        sch.set_timesteps(200)
        x = pos[:, :30].clone()
        for t in sch.timesteps:
            x = sch.step(pos_net(x, t.reshape(-1), None), t, x)
            yield

    def leg_b():
        x = pos.clone()
        for t in dts:
            x = ddpm.step(pos_net(x, t.reshape(-1), None), t, x, noise=torch.randn(x.shape, generator=g))
            yield

    def leg_c():
        sch = OraclePNDM()
        sch.set_timesteps(200)
        x = z.clone()
        for t in sch.timesteps:
            x = sch.step(z_net(x, t.reshape(-1), pos, mask, None), t, x)
            yield

    def clock(gen, n):
        t0 = time.perf_counter()
        for _ in range(n):
            next(gen)
        return (time.perf_counter() - t0) / n

    phys = _physical_cores()
    calib = {}
    with torch.no_grad():
        for n in sorted({phys, min(32, phys)}, reverse=True):
            torch.set_num_threads(n)
            it = leg_c()
            next(it)                                                # page in / build the thread pool
            calib[n] = clock(it, 1)
        threads = min(calib, key=calib.get)
        torch.set_num_threads(threads)
        per_leg = {}
        for (key, _, _), fn in zip(LEGS, (leg_a, leg_b, leg_c)):
            it = fn()
            clock(it, warmup)
            per_leg[key] = clock(it, steps)
    tot = sum(w for _, _, w in LEGS)
    per = sum(per_leg[k] * w for k, _, w in LEGS) / tot * (B_PER_GPU / CPU_SAMPLE_B)
    return {"value": round(1.0 / per, 4), "unit": "denoising-steps/s (batch=512)", "cores": threads, "kind": "port",
            "sample": f"{CPU_SAMPLE_B} of the 512 samples, {steps} timed iterations of each of the three face-LDM loops after "
                      f"{warmup} warm-up, weighted 158:250:209 (x{B_PER_GPU // CPU_SAMPLE_B} to a full batch: the path is "
                      "per-sample); the reference's formulation (nn.TransformerEncoder seq-first, oracle/ref_formulation.py) + "
                      "oracle/schedulers.py, fp32, torch CPU",
            "s_per_step_batch512": round(per, 3),
            "s_per_leg_step_batch512": {k: round(v * B_PER_GPU / CPU_SAMPLE_B, 3) for k, v in per_leg.items()},
            "threads": threads, "physical_cores": phys, "logical_cpus": os.cpu_count(), "cpu_model": _cpu_model(),
            "calibration_s_per_sample_step": {str(k): round(v, 3) for k, v in calib.items()}}


def _sensors(dev):
    """hwmon files of THIS device: package power (W), shader clock (MHz), power cap (W) -> read(key) in the sensor's unit / 1e6."""
    import glob
    pr = torch.cuda.get_device_properties(dev)
    want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0)) if hasattr(pr, "pci_bus_id") else None
    sens = {}
    for card in sorted(glob.glob("/sys/class/drm/card*")):
        if "-" in os.path.basename(card) or (want and want not in os.path.realpath(os.path.join(card, "device"))):
            continue
        for hw in glob.glob(os.path.join(card, "device", "hwmon", "hwmon*")):
            for key, names in (("W", ("power1_average", "power1_input")), ("MHz", ("freq1_input",)), ("cap_W", ("power1_cap",))):
                for n in names:
                    if key not in sens and os.path.exists(os.path.join(hw, n)):
                        sens[key] = os.path.join(hw, n)

    def read(key):
        try:
            return int(open(sens[key]).read()) / 1e6
        except (KeyError, OSError, ValueError):
            return None
    return read


def face_ldm_full_pass_extra(ldm, dev, barrier):
    """ONE real face LDM as sample.py:126-202 runs it -- 158 + 250 + 209 iterations back to back (the headline's timed region is
    a short, step-weighted sample of it) -- with the package power and shader clock sampled while it runs: what the part SUSTAINS
    over ~2 s of this step mix, not over 60 ms."""
    import threading
    read = _sensors(dev)
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            samples.append((read("W"), read("MHz")))
            time.sleep(0.02)
    ldm.run(1, 1, 1)
    barrier()
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.perf_counter()
    ldm.run(158, 250, 209)
    barrier()
    d = time.perf_counter() - t0
    stop.set()
    th.join()
    tail = samples[len(samples) // 2:]
    med = lambda v: (sorted(v)[len(v) // 2] if v else None)
    return {"iterations": {"A": 158, "B": 250, "C": 209}, "seconds": round(d, 4), "ms_per_step": round(1e3 * d / 617, 4),
            "steps_per_s_per_gpu": round(617 / d, 2), "samples_per_s_per_gpu": round(B_PER_GPU / d, 1),
            "power_W_second_half": med([p for p, _ in tail if p is not None]),
            "shader_clock_MHz_second_half": med([f for _, f in tail if f is not None])}


def sustained_clock_extra(dev, secs=0.6):
    """Reported BESIDE the headline: what the part sustains under the MFMA-bound launch of the step (the fused QKV + attention
    launch of SurfPosNet at 512 x 60) -- microseconds per launch, package power and shader clock (hwmon of THIS device), with the
    bench's random operands and with all-zero operands (the same instruction stream, no bits toggling).  The roofline's MFMA peak
    assumes 2.4 GHz; with random 16-bit operands the 1400 W cap leaves less."""
    import threading
    from brepgen_amd import _lib
    lib = _lib.load()
    read = _sensors(dev)

    B, N = B_PER_GPU, N_FACE
    M = B * N
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, 768, generator=g) * 2
    grp = x.reshape(M, 12, 64)
    stats = torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2).contiguous().to(dev)
    a = x.to(torch.bfloat16).to(dev)
    w = (torch.randn(2304, 768, generator=g) * 0.04).to(torch.bfloat16).to(dev)
    b = torch.randn(2304, generator=g).to(dev)
    cs = w.float().sum(1).contiguous()
    out = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    res = {"power_cap_W": read("cap_W")}
    flop = None
    for name, aa, ww in (("random_operands", a, w), ("zero_operands", torch.zeros_like(a), torch.zeros_like(w))):
        def fn():
            _lib.check(lib.bg_qkv_attn_fwd(aa.data_ptr(), ww.data_ptr(), b.data_ptr(), cs.data_ptr(), stats.data_ptr(), None,
                                           out.data_ptr(), None, B, N, _lib.BG_BF16, 1e-5, st), "bg_qkv_attn_fwd")
        samples, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                samples.append((read("W"), read("MHz")))
                time.sleep(0.03)
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        if flop is None:                     # the library's own FLOP count of this launch (opt-in profiler: GEMM + attention)
            with _lib.profile(16) as prof:
                fn()
            flop = sum(r["flops"] for r in prof.rows)
            res["launch"] = "bg_qkv_attn_fwd, 512 x 60 tokens, bf16 (%.1f GFLOP)" % (flop / 1e9)
        th = threading.Thread(target=poll)
        th.start()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0, n = time.perf_counter(), 0
        e0.record()
        while time.perf_counter() - t0 < secs:
            for _ in range(50):
                fn()
            n += 50
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        tail = samples[len(samples) // 2:]
        med = lambda v: (sorted(v)[len(v) // 2] if v else None)
        us = e0.elapsed_time(e1) / n * 1e3
        res[name] = {"us_per_launch": round(us, 1), "tflops": round(flop / us / 1e6, 1),
                     "power_W": med([p for p, _ in tail if p is not None]), "shader_clock_MHz": med([f for _, f in tail if f is not None])}
    return res


def edge_net_extra(dev, evals=2):
    """Reported BESIDE the headline (never part of `value`): one eps-evaluation of the edge nets at the shapes of
    BASELINE configs[2] / configs[3] / configs[4] (the last one guided: conditional + unconditional rows, fp16) -- where
    98 % of the cascade's FLOPs are -- with their own roofline.
    Masks per SURVEY 8(d): valid faces per sample ~ U{8..S}, valid edges per valid face ~ U{3..E}."""
    import brepgen_amd as bga
    from brepgen_amd import _lib
    out = []
    for name, cls, B, S, E, c_tok, cf, dt in (
            ("EdgeZNet cfg3 DeepCAD [256,60,30,18] bf16", bga.EdgeZNet, 256, 60, 30, 4.78e6, False, torch.bfloat16),
            ("EdgePosNet cfg4 ABC [512,100,40,6] bf16 (one rank's 512 samples)", bga.EdgePosNet, 512, 100, 40, 2.38e6, False, torch.bfloat16),
            ("EdgeZNet cfg5 Furniture guided [2x256,60,40,18] fp16 (one rank's 256 samples: conditional + unconditional rows "
             "in one eval)", bga.EdgeZNet, 256, 60, 40, 4.78e6, True, torch.float16)):
        g = torch.Generator().manual_seed(99)
        torch.manual_seed(1)
        net = cls(cf).to(dev).eval()
        net.compute_dtype = dt
        nf = torch.randint(8, S + 1, (B,), generator=g)
        smask = torch.arange(S)[None] >= nf[:, None]                                     # [B,S] True = padded face
        pos = torch.randn(B, S, 6, generator=g).clamp(-3, 3).to(dev)
        sz = torch.randn(B, S, 48, generator=g).to(dev)
        t = torch.tensor([249], device=dev)
        if cls is bga.EdgeZNet:
            ne = torch.randint(3, E + 1, (B, S), generator=g)
            emask = (torch.arange(E)[None, None] >= ne[:, :, None]) | smask[:, :, None]     # [B,S,E]
            ntok = (~emask).sum((1, 2)).double()
            args = (torch.randn(B, S, E, 18, generator=g).to(dev), t, torch.randn(B, S, E, 6, generator=g).clamp(-3, 3).to(dev),
                    pos, sz, emask.to(dev), None)
        else:
            ntok = (nf * E).double()
            args = (torch.randn(B, S, E, 6, generator=g).clamp(-3, 3).to(dev), t, pos, sz, smask.to(dev), None)
        if cf:                                    # sample.py:273-279: inputs repeated, labels [class] * B + [uncond] * B
            rep = lambda v: v.repeat(2, *([1] * (v.dim() - 1))).contiguous() if torch.is_tensor(v) and v.dim() > 1 else v
            args = tuple(rep(a) for a in args[:-1]) + (torch.tensor([6] * B + [0] * B, dtype=torch.int64, device=dev).reshape(-1, 1),)
            ntok = ntok.repeat(2)
        rows_b = 2 * B if cf else B
        N = S * E
        f_dense = rows_b * (algorithmic_flops_per_sample_eval(N, c_tok) + S * 2.44e6)
        f_exec = float(sum(algorithmic_flops_per_sample_eval(float(n), c_tok) for n in ntok)) + rows_b * S * 2.44e6
        net.profile_hints = (float(ntok.sum()), float((ntok * ntok).sum()))
        row = {"workload": name, "tokens_per_sample": N, "valid_tokens_mean": round(float(ntok.mean()), 1),
               "algorithmic_tflop_per_eval": round(f_dense / 1e12, 2), "executed_tflop_per_eval": round(f_exec / 1e12, 2)}
        with torch.no_grad():
            for mode in ("varlen", "dense"):
                net.varlen = mode == "varlen"
                for _ in range(3):                # (the third call with one mask tensor runs with its counted hints, as every later step of a loop does)
                    net(*args)
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(evals):
                    net(*args)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / evals
                fl = f_exec if net.varlen else f_dense
                row[mode] = {"ms_per_eval": round(dt * 1e3, 2), "executed_tflops": round(fl / dt / 1e12, 1),
                             "frac_of_mfma_peak": round(fl / dt / 1e12 / MFMA_PEAK_TFLOPS, 4)}
            net.varlen = True
            net.n_split = 1                               # per-kernel numbers: launches serialised
            with _lib.profile() as prof:
                net(*args)
            net.n_split = "auto"
            row["kernels_varlen"] = {r["kernel"]: {"launches": r["launches"], "total_ms": round(r["total_ms"], 3),
                                                   "tflops": round(r["flops"] / r["total_ms"] / 1e9, 1) if r["flops"] else None,
                                                   "frac_of_mfma_peak": round(r["flops"] / r["total_ms"] / 1e9 / MFMA_PEAK_TFLOPS, 4) if r["flops"] else None}
                                     for r in prof.rows if r["total_ms"] > 0.05}
        out.append(row)
        del net, args
        torch.cuda.empty_cache()
    return out


class FaceLDM:
    """The three loops of the face LDM on the HIP path (one rank's 512 samples), each an endless step generator."""

    def __init__(self, dev, rank, dense=False, split=0, dtype=torch.bfloat16):
        import brepgen_amd as bga
        from brepgen_amd.sampling import device_randn
        torch.manual_seed(0)
        self.dev, self.rank, self._randn = dev, rank, device_randn
        self.pos_net = bga.SurfPosNet(False).to(dev).eval()
        self.z_net = bga.SurfZNet(False).to(dev).eval()
        for n in (self.pos_net, self.z_net):
            n.compute_dtype = dtype
            if split:
                n.n_split = split
        self.z_net.cache_conditioning = False     # every step recomputes p_embed(surfPos) like the reference does (network.py:1182)
        self.z_net.varlen = not dense             # variable-length execution: only the valid faces (U{8..60} of 60) run through the net
        z, pos, mask = make_inputs(B_PER_GPU, dev, 1234 + rank)
        self.nvalid = make_inputs.nvalid.double()
        if self.z_net.varlen:                     # the opt-in profiler books EXECUTED rows / attention pairs (host-side knowledge)
            self.z_net.profile_hints = (float(self.nvalid.sum()), float((self.nvalid * self.nvalid).sum()))
        self.pos, self.mask = pos, mask
        kw = dict(num_train_timesteps=1000, beta_schedule="linear", prediction_type="epsilon", beta_start=0.0001, beta_end=0.02)
        self.pndm_a, self.pndm_c = bga.PNDMScheduler(**kw), bga.PNDMScheduler(**kw)
        self.ddpm = bga.DDPMScheduler(clip_sample=True, clip_sample_range=3, **kw)
        self.ddpm.set_timesteps(1000)
        self.x = {"A": pos[:, :30].clone(), "B": pos.clone(), "C": z}
        self.gens = {"A": self._leg_a(), "B": self._leg_b(), "C": self._leg_c()}

    def set_split(self, ns):
        self.pos_net.n_split = self.z_net.n_split = ns

    def n_split(self, net):
        ns = net.n_split
        return (2 if B_PER_GPU * N_FACE >= 16384 else 1) if ns == "auto" else int(ns)

    def _leg_a(self):                             # sample.py:128-137
        while True:
            self.pndm_a.set_timesteps(200)
            ts = self.pndm_a.timesteps[:LEGS[0][2]]
            tsd = ts.to(self.dev)
            for i in range(len(ts)):
                eps = self.pos_net(self.x["A"], tsd[i:i + 1], None)
                self.x["A"] = self.pndm_a.step(eps, ts[i], self.x["A"]).prev_sample
                yield

    def _leg_b(self):                             # sample.py:144-153 (noise drawn on the device, keyed on the global sample index)
        ts = self.ddpm.timesteps[-LEGS[1][2]:]
        tsd = ts.to(self.dev)
        draw = 0
        while True:
            for i in range(len(ts)):
                eps = self.pos_net(self.x["B"], tsd[i:i + 1], None)
                draw += 1
                noise = self._randn(tuple(self.x["B"].shape), 20240917, draw, self.rank * B_PER_GPU, self.dev)
                self.x["B"] = self.ddpm.step(eps, ts[i], self.x["B"], noise=noise).prev_sample
                yield

    def _leg_c(self):                             # sample.py:191-202
        while True:
            self.pndm_c.set_timesteps(200)
            ts = self.pndm_c.timesteps
            tsd = ts.to(self.dev)
            for i in range(len(ts)):
                eps = self.z_net(self.x["C"], tsd[i:i + 1], self.pos, self.mask, None)
                self.x["C"] = self.pndm_c.step(eps, ts[i], self.x["C"]).prev_sample
                yield

    def run(self, ka, kb, kc):
        with torch.no_grad():
            for key, k in (("A", ka), ("B", kb), ("C", kc)):
                g = self.gens[key]
                for _ in range(k):
                    next(g)

    def flops(self, ka, kb, kc):
        """(algorithmic, executed) FLOPs of ka + kb + kc iterations (F(N) = N (12 * 7,864,320 + C_io) + 36,864 N^2)."""
        fa = B_PER_GPU * algorithmic_flops_per_sample_eval(30, 2.38e6)
        fb = B_PER_GPU * algorithmic_flops_per_sample_eval(N_FACE, 2.38e6)
        fc = B_PER_GPU * algorithmic_flops_per_sample_eval(N_FACE, 3.70e6)
        # executed: F(n_b) summed over the samples (n_b valid faces each); the padded conditioning embed p_embed(surfPos)
        # (2.44 MFLOP per face) still runs on all 60 faces
        fc_exec = float(sum(algorithmic_flops_per_sample_eval(float(n), 3.70e6 - 2.44e6) for n in self.nvalid)) + \
            B_PER_GPU * N_FACE * 2.44e6 if self.z_net.varlen else fc
        return ka * fa + kb * fb + kc * fc, ka * fa + kb * fb + kc * fc_exec


def cascade_extra():
    """BASELINE configs[2] end to end (tools/cascade_bench.py: the whole DeepCAD cascade + VAE decode, batch 256, bf16)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import cascade_bench
    return cascade_bench.run(256)


def rank_local_extra(name, k=None):
    """BASELINE configs[3] / configs[4] as one rank runs them (tools/rank_local_bench.py): the four cascade loops IN FULL
    (408 + 209 + 408 + 209 iterations) at the rank-local batch, measured in this run (~90 s + ~75 s)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import rank_local_bench
    return rank_local_bench.run(name, k)


def torch_eager_extra(dev, steps=10, warmup=3):
    """The like-for-like "reference PyTorch on this GPU" figure: the reference's formulation of
    SurfZNet -- stock torch.nn blocks composed as network.py:1133-1200 composes them (tools/torch_eager_baseline.py: EagerSurfZ; no
    oracle, no reference import, random-init weights) -- through torch-ROCm EAGER on this device, fp32 and under torch.autocast (fp16:
    what sample.py:121 runs; bf16: the bench's dtype): one eps-evaluation at the headline's loop-C shape (512 x 60, key-padding mask)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from torch_eager_baseline import EagerSurfZ
    torch.manual_seed(0)
    net = EagerSurfZ().to(dev).eval()
    z, pos, mask = (t.to(dev) for t in make_inputs(B_PER_GPU, "cpu", 1234))
    t = torch.full((B_PER_GPU,), 249, device=dev, dtype=torch.long)
    out = {"workload": "SurfZNet eps-evaluation, 512 x 60 tokens + key-padding mask, dense (as the reference runs it), torch-ROCm eager of "
                       "stock nn.TransformerEncoder (tools/torch_eager_baseline.py)", "torch": torch.__version__}
    for name, dt in (("autocast_bf16", torch.bfloat16), ("autocast_fp16", torch.float16), ("fp32", None)):
        def fn():
            with torch.no_grad():
                if dt is None:
                    return net(z, t, pos, mask)
                with torch.autocast("cuda", dtype=dt):
                    return net(z, t, pos, mask)
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eps = fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out[name] = {"ms_per_eval": round(ms, 3), "evals_per_s": round(1e3 / ms, 2), "finite": bool(torch.isfinite(eps[~mask]).all())}
    del net
    torch.cuda.empty_cache()
    return out


def cpu_stage_baselines(threads):
    """CPU cost of the stages of BASELINE configs[2] / [3] / [4]: the reference's formulation
    (oracle/ref_formulation.py, fp32, dense -- every padded position, as the reference computes it) of each eps-net at a REDUCED
    batch, ONE timed evaluation after one warm-up, scaled linearly in the batch (every op of the path is per-sample) and
    multiplied by the loop's iteration count (sample.py:128-282; x 2 evaluations per iteration with classifier-free guidance).
    A projection, stated as such: the only measured quantities are `s_per_eval` at `batch_timed`."""
    from oracle import denoisers as orc
    from oracle import ref_formulation as rf
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(11)
    R = lambda *sh: torch.randn(*sh, generator=g)
    t = torch.tensor([249])

    def clock(fn):
        with torch.no_grad():
            fn()
            t0 = time.perf_counter()
            fn()
        return time.perf_counter() - t0

    nets = {n: rf.build(n, orc.seeded_state_dict(n, 0)) for n in ("SurfPosNet", "SurfZNet", "EdgePosNet", "EdgeZNet")}

    def stage_times(S, E, b_face, b_edge, s_first):
        m_face = torch.zeros(b_face, S, dtype=torch.bool)
        m_edge = torch.zeros(b_edge, S, dtype=torch.bool)
        em = torch.zeros(b_edge, S, E, dtype=torch.bool)
        return {
            "surfPos_first158": clock(lambda: nets["SurfPosNet"](R(b_face, s_first, 6), t, None)) / b_face,
            "surfPos": clock(lambda: nets["SurfPosNet"](R(b_face, S, 6), t, None)) / b_face,
            "surfZ": clock(lambda: nets["SurfZNet"](R(b_face, S, 48), t, R(b_face, S, 6), m_face, None)) / b_face,
            "edgePos": clock(lambda: nets["EdgePosNet"](R(b_edge, S, E, 6), t, R(b_edge, S, 6), R(b_edge, S, 48), m_edge, None)) / b_edge,
            "edgeZV": clock(lambda: nets["EdgeZNet"](R(b_edge, S, E, 18), t, R(b_edge, S, E, 6), R(b_edge, S, 6), R(b_edge, S, 48), em, None)) / b_edge,
        }

    out = {"threads": threads, "formulation": "oracle/ref_formulation.py (stock nn.TransformerEncoder, seq-first), fp32, dense, torch CPU",
           "projection": "loop seconds = s_per_eval_per_sample x batch x iterations (x 2 evaluations with guidance); without guidance the "
                         "first 158 surfPos iterations run on half the faces (the late doubling, sample.py:139-142)"}
    for name, S, E, B, evals, what in (("cfg3", 60, 30, 256, 1, "DeepCAD cascade, batch 256 (2 x 30 faces x 30 edges = 1800 edge tokens)"),
                                       ("cfg4", 100, 40, 512, 1, "ABC, one rank's 512 samples (2 x 50 faces x 40 edges = 4000 edge tokens)"),
                                       ("cfg5", 60, 40, 256, 2, "furniture, one rank's 256 samples, classifier-free guidance (60 faces x 40 edges = 2400 edge tokens)")):
        st = stage_times(S, E, 4, 1, S // 2 if evals == 1 else S)
        loops = {"surfPos": B * evals * (158 * st["surfPos_first158"] + 250 * st["surfPos"]), "surfZ": B * evals * 209 * st["surfZ"],
                 "edgePos": B * evals * 408 * st["edgePos"], "edgeZV": B * evals * 209 * st["edgeZV"]}
        out[name] = {"workload": what, "batch_timed": {"face_nets": 4, "edge_nets": 1},
                     "s_per_eval_per_sample": {k: round(v, 4) for k, v in st.items()},
                     "projected_loop_s": {k: round(v, 1) for k, v in loops.items()}, "projected_loops_s": round(sum(loops.values()), 1),
                     "projected_samples_per_s": round(B / sum(loops.values()), 5)}
    return out


def pmc_traffic(kernel, launches_per_step):
    """Fabric bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written
    by tools/pmc_summary.py from `bench.py --steps 20 --warmup 5 --split 1` under --pmc FETCH_SIZE / WRITE_SIZE: (2 x FETCH_SIZE
    + WRITE_SIZE) KiB averaged over that kernel's launches -- the x2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md).
    A stored measurement is only comparable with this run's algorithmic bytes when it was taken on the same step mix: the file
    records the kernel's launches per step of ITS run, and a mismatch of more than 5 % with this run's returns None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
            row = table.get(kernel) or table.get(kernel.split("(")[0], {})
    except (OSError, ValueError):
        return None, None
    lps = row.get("launches_per_step")
    if not row.get("bytes_per_launch") or not lps or abs(lps - launches_per_step) > 0.05 * launches_per_step:
        return None, None
    return row["bytes_per_launch"], row.get("measured_on")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the measurements reported beside the headline")
    ap.add_argument("--dry-run", action="store_true", help="launch + rendezvous + JSON only, on CPU (no compute)")
    ap.add_argument("--dense", action="store_true", help="run every padded position of loop C like the reference (no compaction)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group even at world size 1 "
                                                              "(exercises the collective path on a 1-GPU box)")
    ap.add_argument("--split", type=int, default=0, help="sample groups run concurrently on forked streams (0 = the "
                                                         "module's default: 2 at this batch size; 1 = off)")
    args = ap.parse_args()

    ap_world = os.environ.get("WORLD_SIZE")
    if ap_world is None and args.gpus > 1:
        # launched bare (`python bench.py --gpus N`): become the launcher -- one rank per GPU under torch.distributed.run
        if not args.dry_run and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but this node exposes {torch.cuda.device_count()} GPU(s)")
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(ap_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}), "
                         "or run `python bench.py --gpus N` bare and let it spawn the ranks")
    if args.dry_run:
        # launcher / rendezvous / JSON plumbing only (CPU, gloo): what tests/test_bench_cpu.py exercises without a GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "denoising-steps/sec (whole node), DeepCAD face-LDM, batch=512", "value": None,
                              "unit": "denoising-steps/s (batch=512 per step)", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "dry_run": True, "ranks_seen": int(tt)}), flush=True)
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from brepgen_amd import _lib
    from brepgen_amd.sampling import gather_latents

    ldm = FaceLDM(dev, rank, dense=args.dense, split=args.split)
    n_split = ldm.n_split(ldm.z_net)
    ka, kb, kc = split_steps(args.steps)
    wa, wb, wc = split_steps(max(args.warmup, 3))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def latents():
        return {"surfPos": ldm.x["B"], "surfZ": ldm.x["C"]}

    ldm.run(wa, wb, wc)
    if dist is not None:
        gather_latents(latents(), dist, single_rank_collective=args.force_dist)      # warm the communicator up outside the clock
    barrier()
    t0 = time.perf_counter()
    ldm.run(ka, kb, kc)
    out = gather_latents(latents(), dist, single_rank_collective=args.force_dist) if dist is not None else latents()
    barrier()
    elapsed = my_elapsed = time.perf_counter() - t0
    ranks_seen, per_rank_ms = 1, [round(1e3 * my_elapsed / args.steps, 4)]
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
        ones = torch.ones(1, device=dev, dtype=torch.float64)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)                    # how many ranks really took part
        ranks_seen = int(ones.item())
        every = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(every, torch.tensor([my_elapsed], device=dev, dtype=torch.float64))
        per_rank_ms = [round(1e3 * float(v) / args.steps, 4) for v in every]
    finite = all(bool(torch.isfinite(v).all()) for v in out.values())
    flops_cache = ldm.flops(ka, kb, kc)

    def clock(ka_, kb_, kc_):
        ldm.run(1 if ka_ else 0, 1 if kb_ else 0, 1 if kc_ else 0)
        barrier()
        t1 = time.perf_counter()
        ldm.run(ka_, kb_, kc_)
        barrier()
        return time.perf_counter() - t1

    extra = {}
    if not args.no_extra:
        # each loop on its own (20 iterations), never part of `value`
        legs = {}
        for (key, what, w), ks in zip(LEGS, ((20, 0, 0), (0, 20, 0), (0, 0, 20))):
            d = clock(*ks)
            fa, fe = ldm.flops(*ks)
            legs[key] = {"workload": what + ", bf16", "iterations_of_617": w, "ms_per_step": round(1e3 * d / 20, 4),
                         "steps_per_s_per_gpu": round(20 / d, 2), "executed_tflops": round(fe / d / 1e12, 1)}
        extra["face_ldm_legs"] = legs
        if n_split > 1:
            # the same K steps with the launches of a step serialised on one stream (n_split = 1)
            ldm.set_split(1)
            d = clock(ka, kb, kc)
            extra["single_stream_execution"] = {"ms_per_step": round(1e3 * d / args.steps, 4), "steps_per_s_per_gpu": round(args.steps / d, 3)}
            ldm.set_split(args.split if args.split else "auto")
        if ldm.z_net.varlen:
            # loop C with dense execution (every padded position computed, as the reference does)
            ldm.z_net.varlen, hints = False, ldm.z_net.profile_hints
            ldm.z_net.profile_hints = None
            d = clock(0, 0, 20)
            extra["dense_execution_loop_C"] = {"ms_per_step": round(1e3 * d / 20, 4), "steps_per_s_per_gpu": round(20 / d, 3)}
            ldm.z_net.varlen, ldm.z_net.profile_hints = True, hints
            # the composite on the DENSE definition of a step (every padded position computed, as the reference runs
            # loop C): the per-loop times above, weighted 158 : 250 : 209
            tot_w = sum(w for _, _, w in LEGS)
            ms_dense = (LEGS[0][2] * legs["A"]["ms_per_step"] + LEGS[1][2] * legs["B"]["ms_per_step"] +
                        LEGS[2][2] * extra["dense_execution_loop_C"]["ms_per_step"]) / tot_w
            extra["dense_definition_composite"] = {"ms_per_step": round(ms_dense, 4), "steps_per_s_per_gpu": round(1e3 / ms_dense, 2),
                                                   "from": "face_ldm_legs A, B + dense_execution_loop_C (20 iterations each)"}
        # the same three loops with fp16 operands -- the autocast dtype the reference itself runs (sample.py:121) and the drop-in
        # modules' default inside torch.autocast('cuda')
        ldm16 = FaceLDM(dev, rank, dense=args.dense, split=args.split, dtype=torch.float16)
        ldm16.run(1, 1, 1)
        barrier()
        t1 = time.perf_counter()
        ldm16.run(*split_steps(20))
        barrier()
        d = time.perf_counter() - t1
        extra["fp16_operands_composite"] = {"ms_per_step": round(1e3 * d / 20, 4), "steps_per_s_per_gpu": round(20 / d, 2),
                                            "steps_per_loop": dict(zip("ABC", split_steps(20)))}
        del ldm16

    roofline = None
    breakdown = None
    if not args.no_roofline:
        # second pass of the same K steps with a hipEvent pair around every kernel launch (on the launch stream).  The
        # launches are serialised for it (n_split = 1): with two sample groups in flight every launch shares the CUs with
        # a launch of the other group, and its duration then says nothing about the kernel.
        ldm.set_split(1)
        ldm.run(1, 1, 1)
        with _lib.profile() as prof:
            ldm.run(ka, kb, kc)
        ldm.set_split(args.split if args.split else "auto")
        rows = {r["kernel"]: r for r in prof.rows}
        breakdown = {k: {"launches": r["launches"], "avg_us": round(1e3 * r["total_ms"] / r["launches"], 2),
                         "total_ms_per_step": round(r["total_ms"] / args.steps, 4),
                         "tflops": round(r["flops"] / r["total_ms"] / 1e9, 1) if r["flops"] else None,
                         "gbs": round(r["bytes"] / r["total_ms"] / 1e6, 1)} for k, r in rows.items()}
        dom = max(rows.values(), key=lambda r: r["total_ms"])
        # which roofline bounds the dominant kernel: its arithmetic intensity (algorithmic FLOPs / algorithmic bytes of its
        # launches) against the ridge MFMA peak / HBM peak = 312 FLOP/B.  QKV / FFN1 (567 / 427 FLOP/B) are MFMA-bound; the
        # residual-stream GEMMs (out-proj / FFN2 with the split residual: 153 / 192 FLOP/B -- 8 B of residual traffic per output
        # element) and every elementwise kernel are HBM-bound and are judged as GB/s.
        ridge = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        intensity = dom["flops"] / dom["bytes"] if dom["bytes"] else float("inf")
        traffic, traffic_src = pmc_traffic(dom["kernel"], dom["launches"] / args.steps)
        if intensity >= ridge:
            ach = dom["flops"] / dom["total_ms"] / 1e9                # TFLOP/s = flops per launch / avg duration
            bound, peak, unit = "mfma", MFMA_PEAK_TFLOPS, "TFLOP/s"
        else:
            ach = dom["bytes"] / dom["total_ms"] / 1e6                # GB/s = algorithmic bytes per launch / avg duration
            bound, peak, unit = "hbm", HBM_PEAK_GBS, "GB/s"
        roofline = {"kernel": dom["kernel"], "bound": bound, "achieved": round(ach, 1), "peak": peak,
                    "unit": unit, "frac": round(ach / peak, 4), "traffic": traffic,
                    "traffic_over_algorithmic_bytes": round(traffic / (dom["bytes"] / dom["launches"]), 3) if traffic else None,
                    "traffic_measured_on": traffic_src,
                    "arithmetic_intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
                    "launches_per_step": round(dom["launches"] / args.steps, 2),
                    "avg_launch_us": round(1e3 * dom["total_ms"] / dom["launches"], 2),
                    "flops_per_launch": dom["flops"] / dom["launches"],
                    "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
                    "tflops": round(dom["flops"] / dom["total_ms"] / 1e9, 1), "gbs": round(dom["bytes"] / dom["total_ms"] / 1e6, 1),
                    "flops": "executed (valid rows only) over the launches of the three loops",
                    "measured_with": "launches serialised (n_split = 1); the timed region runs n_split = %d" % n_split}
        # the two largest kernels side by side (they trade places from box to box: ~1.3 ms per step each)
        roofline["by_kernel"] = {
            k: {"ms_per_step": round(r["total_ms"] / args.steps, 4),
                "bound": "mfma" if (r["bytes"] and r["flops"] / r["bytes"] >= ridge) else "hbm",
                "frac_of_mfma_peak": round(r["flops"] / r["total_ms"] / 1e9 / MFMA_PEAK_TFLOPS, 4),
                "frac_of_hbm_peak": round(r["bytes"] / r["total_ms"] / 1e6 / HBM_PEAK_GBS, 4)}
            for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["total_ms"])[:4] if r["flops"]}
        # every GEMM kernel of the step together (the 256 x 256 kernel takes the row panels that fill whole rounds, the
        # 128 x 128 kernel the rest and the residual-stream GEMMs): executed FLOPs / their summed durations
        gem = [r for k, r in rows.items() if k.startswith("gemm16") or k.startswith("qkv_attn")]   # (the fused launch: QKV GEMM + 5 % attention FLOPs)
        if gem:
            fl, ms = sum(r["flops"] for r in gem), sum(r["total_ms"] for r in gem)
            roofline["all_16bit_gemm_kernels"] = {"tflops": round(fl / ms / 1e9, 1), "frac": round(fl / ms / 1e9 / MFMA_PEAK_TFLOPS, 4),
                                                  "ms_per_step": round(ms / args.steps, 4)}

    if world == 1 and rank == 0 and not args.no_extra:
        try:
            extra["face_ldm_full_pass"] = face_ldm_full_pass_extra(ldm, dev, barrier)
        except Exception as e:
            extra["face_ldm_full_pass"] = {"error": repr(e)}
        ldm = None
        torch.cuda.empty_cache()
        try:
            extra["torch_eager_reference_formulation"] = torch_eager_extra(dev)
        except Exception as e:
            extra["torch_eager_reference_formulation"] = {"error": repr(e)}
        extra["edge_nets"] = edge_net_extra(dev)
        try:
            extra["sustained_clock"] = sustained_clock_extra(dev)
        except Exception as e:
            extra["sustained_clock"] = {"error": repr(e)}
        try:
            extra["cascade_cfg3"] = cascade_extra()
        except Exception as e:                                       # reported, never fatal for the headline
            extra["cascade_cfg3"] = {"error": repr(e)}
        for name in ("cfg4", "cfg5"):
            try:
                extra["rank_local_" + name] = rank_local_extra(name)
            except Exception as e:
                extra["rank_local_" + name] = {"error": repr(e)}

    if rank == 0:
        steps_per_s = world * args.steps / elapsed
        f_alg, f_exec = flops_cache
        line = {
            "metric": "denoising-steps/sec (whole node), DeepCAD face-LDM, batch=512",
            "value": round(steps_per_s, 3), "unit": "denoising-steps/s (batch=512 per step)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "value_dense": (round(world * extra["dense_definition_composite"]["steps_per_s_per_gpu"], 3)
                            if "dense_definition_composite" in extra else None),
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: DeepCAD face-LDM = the three loops of sample.py:126-202, step-weighted: "
                                   f"{ka} x [SurfPosNet [512,30,6] + PNDM], {kb} x [SurfPosNet [512,60,6] + DDPM, device noise], "
                                   f"{kc} x [SurfZNet [512,60,48]+bbox+mask + PNDM] (158 : 250 : 209), bf16 operands / fp32 accumulate, per GPU",
                       "batch_per_gpu": B_PER_GPU, "tokens_per_sample": N_FACE,
                       "steps_per_loop": {"A": ka, "B": kb, "C": kc},
                       "sample_steps_per_s": round(steps_per_s * B_PER_GPU, 1),
                       "algorithmic_tflop_per_step": round(f_alg / args.steps / 1e12, 3),
                       "executed_tflop_per_step": round(f_exec / args.steps / 1e12, 3),
                       "varlen": not args.dense, "n_split": n_split,
                       "valid_faces_per_sample_mean": round(float(make_inputs.nvalid.double().mean()), 2),
                       "dense_equivalent_tflops_per_gpu": round(f_alg / elapsed / 1e12, 1),
                       "executed_tflops_per_gpu": round(f_exec / elapsed / 1e12, 1),
                       "finite": finite, "ranks_seen": ranks_seen, "per_rank_ms_per_step": per_rank_ms,
                       "parallelism": f"batch-sharded x{world}, 1 all_gather of latents",
                       "formulation": ("%d sample group(s) per eps-evaluation on forked streams inside the one C call; " % n_split)
                                      + ("variable-length execution of loop C (valid faces compacted on the device; eps = 0 at padded "
                                         "positions, valid positions as the dense path), " if not args.dense else "dense execution, ")
                                      + "norm1/norm2 folded into the QKV/FFN1 GEMM epilogues, residual stream as (hi, lo) 16-bit planes, "
                                        "fused input embeds; conditioning cache off (every embed recomputed)"},
            "roofline": roofline, "kernels": breakdown, "extra": extra,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            if not args.no_extra:
                # (GPU side of the same configurations: extra.cascade_cfg3 / rank_local_cfg4 / rank_local_cfg5, measured above)
                try:
                    st = cpu_stage_baselines(line["cpu_baseline"]["threads"])
                    for name, gpu_key in (("cfg3", "cascade_cfg3"), ("cfg4", "rank_local_cfg4"), ("cfg5", "rank_local_cfg5")):
                        gpu = extra.get(gpu_key, {})
                        gpu_s = gpu.get("cascade_s") or gpu.get("loops_s")
                        if gpu_s and name in st:
                            st[name]["gpu_loops_s_this_run"] = gpu_s
                            st[name]["gpu_over_cpu_projection"] = round(st[name]["projected_loops_s"] / gpu_s, 1)
                    line["cpu_baseline"]["cascade_stages"] = st
                except Exception as e:
                    line["cpu_baseline"]["cascade_stages"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
