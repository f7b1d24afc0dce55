#!/usr/bin/env python
"""What clock and power does the part sustain under the MFMA-bound kernels?  Runs the fused QKV + attention kernel and the
256 x 256 LayerNorm-fold GEMM back to back for a few seconds each -- random operands, then zeros (the same instruction stream, no
bits toggling) -- while a thread samples the shader clock and the package power (hwmon sysfs; rocm-smi as a fallback).

    python tools/clock_power_probe.py [seconds per case]"""
import glob
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import hip_ops as ops

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0


def find_sensors():
    """hwmon files of the card torch's device 0 is (matched by PCI address: the host exposes every card of the node in sysfs)."""
    pr = torch.cuda.get_device_properties(0)
    want = None
    if hasattr(pr, "pci_bus_id"):
        want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
    out = {"pci": want}
    for card in sorted(glob.glob("/sys/class/drm/card*")):
        if "-" in os.path.basename(card):
            continue
        real = os.path.realpath(os.path.join(card, "device"))
        if want and want not in real:
            continue
        for hw in glob.glob(os.path.join(card, "device", "hwmon", "hwmon*")):
            for name in ("power1_average", "power1_input"):
                p = os.path.join(hw, name)
                if os.path.exists(p) and "power" not in out:
                    out["power"] = p
            p = os.path.join(hw, "freq1_input")
            if os.path.exists(p) and "sclk" not in out:
                out["sclk"] = p
            p = os.path.join(hw, "power1_cap")
            if os.path.exists(p):
                try:
                    out["cap_W"] = int(open(p).read()) / 1e6
                except (OSError, ValueError):
                    pass
    return out


SENS = find_sensors()


def read_sensors():
    r = {}
    try:
        if "power" in SENS:
            r["W"] = int(open(SENS["power"]).read()) / 1e6
        if "sclk" in SENS:
            r["MHz"] = int(open(SENS["sclk"]).read()) / 1e6
    except (OSError, ValueError):
        pass
    if not r:
        try:
            t = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            r["raw"] = t.strip().replace("\n", " | ")[:300]
        except (OSError, subprocess.SubprocessError):
            pass
    return r


def sampled(fn, secs):
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            samples.append(read_sensors())
            time.sleep(0.05 if SENS else 0.5)

    th = threading.Thread(target=poll)
    fn(); torch.cuda.synchronize()
    th.start()
    t0, n = time.perf_counter(), 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    tail = samples[len(samples) // 2:]                            # the second half: the clock has settled
    w = [s["W"] for s in tail if "W" in s]
    f = [s["MHz"] for s in tail if "MHz" in s]
    raw = [s["raw"] for s in tail if "raw" in s]
    return us, (statistics.median(w) if w else None), (statistics.median(f) if f else None), (raw[-1] if raw else None)


print("sensors:", SENS or "none in sysfs (rocm-smi fallback)", flush=True)
B, N, dt = 512, 60, torch.bfloat16
g = torch.Generator().manual_seed(0)
M = B * N
x = torch.randn(M, 768, generator=g) * 2
grp = x.reshape(M, 12, 64)
stats = torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2).contiguous().cuda()
a = x.to(dt).cuda()
w = (torch.randn(2304, 768, generator=g) * 0.04).to(dt).cuda()
b = torch.randn(2304, generator=g).cuda()
cs = w.float().sum(1).contiguous()
print("idle:", read_sensors(), flush=True)
K2 = 1024
hi, lo = a.clone(), (x - a.float().cpu()).to(dt).cuda()
a2 = (torch.randn(M, K2, generator=g) * 0.5).to(dt).cuda()
w2 = (torch.randn(768, K2, generator=g) * 0.04).to(dt).cuda()
b2 = torch.randn(768, generator=g).cuda()
for name, zero in (("random operands", False), ("zero operands", True), ("random operands", False)):
    aa, ww = (torch.zeros_like(a), torch.zeros_like(w)) if zero else (a, w)
    a2z, w2z = (torch.zeros_like(a2), torch.zeros_like(w2)) if zero else (a2, w2)
    h, l = (torch.zeros_like(hi), torch.zeros_like(lo)) if zero else (hi.clone(), lo.clone())
    for kname, fl, fn in (("fused QKV + attention", 2.0 * M * 768 * 2304, lambda: ops.qkv_attention(aa, ww, b, cs, stats, B, N)),
                          ("256 x 256 GEMM, LayerNorm fold (QKV)", 2.0 * M * 768 * 2304, lambda: ops.linear_ex(aa, ww, b, stats_in=stats, colsum=cs)),
                          ("FFN2, split residual in place", 2.0 * M * 768 * K2,
                           lambda: ops.linear_ex(a2z, w2z, b2, split_out=True, res=(h, l), want_stats=True, inplace=True))):
        us, watts, mhz, raw = sampled(fn, SECS)
        print(f"{kname:38s} {name:16s}: {us:7.1f} us per launch ({fl / us / 1e6:5.0f} TF)  power {watts} W  shader clock {mhz} MHz  {raw or ''}", flush=True)
