"""Like-for-like GPU baseline: what the reference's own formulation costs on the MI355X through torch-ROCm eager.

The reference denoiser (network.py:1133-1200) is stock torch.nn: four Linear-LN-SiLU-Linear MLPs around
`nn.TransformerEncoder(nn.TransformerEncoderLayer(768, 12, norm_first=True, dim_feedforward=1024), 12, LayerNorm)`,
run seq-first under `torch.cuda.amp.autocast()` (sample.py:121).  This tool builds that stack from torch.nn with random
weights (no oracle, no reference import), runs BASELINE configs[1] (B=512, N=60 + key-padding mask) with the same
timing protocol as bench.py, and also times the four per-layer GEMM shapes through `F.linear` (hipBLASLt / rocBLAS) so
the hand-written persistent GEMM can be judged against the vendor library on the same shapes.

    python tools/torch_eager_baseline.py [--steps 20] > gpurun_out/eager.json
"""
import argparse
import json
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def mlp(k, n):
    return nn.Sequential(nn.Linear(k, 768), nn.LayerNorm(768), nn.SiLU(), nn.Linear(768, n))


class EagerSurfZ(nn.Module):
    def __init__(self):
        super().__init__()
        layer = nn.TransformerEncoderLayer(d_model=768, nhead=12, norm_first=True, dim_feedforward=1024, dropout=0.1)
        self.net = nn.TransformerEncoder(layer, 12, nn.LayerNorm(768))
        self.z_embed, self.p_embed, self.time_embed, self.fc_out = mlp(48, 768), mlp(6, 768), mlp(768, 768), mlp(768, 48)

    def forward(self, z, t, pos, mask):
        f = torch.exp(-math.log(10000.0) * torch.arange(384, device=z.device, dtype=torch.float32) / 384)
        a = t[:, None].float() * f[None]
        temb = self.time_embed(torch.cat([torch.cos(a), torch.sin(a)], -1)).unsqueeze(1)
        tok = self.z_embed(z) + self.p_embed(pos) + temb
        out = self.net(src=tok.permute(1, 0, 2), src_key_padding_mask=mask).transpose(0, 1)
        return self.fc_out(out)


def ddpm_step(eps, t, x, acp, clip=3.0):
    """The diffusers-0.27 update written as the same chain of small torch ops."""
    prev = t - 1
    a_t, a_p = acp[t], (acp[prev] if prev >= 0 else torch.ones((), device=x.device))
    b_t, b_p = 1 - a_t, 1 - a_p
    cur_a = a_t / a_p
    x0 = ((x - b_t.sqrt() * eps) / a_t.sqrt()).clamp(-clip, clip)
    mu = (a_p.sqrt() * (1 - cur_a) / b_t) * x0 + (cur_a.sqrt() * b_p / b_t) * x
    if t > 0:
        mu = mu + (b_p / b_t * (1 - cur_a)).clamp(min=1e-20).sqrt() * torch.randn_like(eps, dtype=torch.float32)
    return mu


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, N = 512, 60
    net = EagerSurfZ().to(dev).eval()
    z = torch.randn(B, N, 48, device=dev)
    pos = torch.randn(B, N, 6, device=dev).clamp(-3, 3)
    nvalid = torch.randint(8, N + 1, (B,), device=dev)
    mask = torch.arange(N, device=dev)[None] >= nvalid[:, None]
    acp = torch.cumprod(1 - torch.linspace(1e-4, 0.02, 1000, device=dev), 0)
    out = {"workload": "SurfZNet B=512 N=60 + DDPM step, torch-ROCm eager (nn.TransformerEncoder)", "torch": torch.__version__,
           "device": torch.cuda.get_device_name(0), "steps": a.steps, "warmup": a.warmup, "nets": {}, "gemm": {}}
    state = {"x": z.clone(), "i": 0}

    def step(dtype):
        t = 249 - (state["i"] % 250)
        state["i"] += 1
        ts = torch.full((B,), t, device=dev, dtype=torch.long)
        with torch.no_grad():
            if dtype is None:
                eps = net(state["x"], ts, pos, mask)
            else:
                with torch.autocast("cuda", dtype=dtype):
                    eps = net(state["x"], ts, pos, mask)
            state["x"] = ddpm_step(eps.float(), t, state["x"], acp)

    for name, dt in (("autocast_fp16", torch.float16), ("autocast_bf16", torch.bfloat16), ("fp32", None)):
        state.update(x=z.clone(), i=0)
        ms = timed(lambda: step(dt), a.steps, a.warmup)
        out["nets"][name] = {"ms_per_step": round(ms, 3), "steps_per_s": round(1e3 / ms, 2),
                             "finite": bool(torch.isfinite(state["x"]).all())}

    M = B * N
    for name, (n, k) in {"qkv": (2304, 768), "out_proj": (768, 768), "ffn1": (1024, 768), "ffn2": (768, 1024)}.items():
        for dt in (torch.bfloat16, torch.float16):
            x = torch.randn(M, k, device=dev, dtype=dt)
            w = torch.randn(n, k, device=dev, dtype=dt)
            b = torch.randn(n, device=dev, dtype=dt)
            ms = timed(lambda: F.linear(x, w, b), 50, 10)
            out["gemm"][f"{name}_{str(dt)[6:]}"] = {"us": round(ms * 1e3, 1), "tflops": round(2 * M * n * k / ms / 1e9, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
