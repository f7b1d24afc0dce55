"""CPU-side checks of the C-ABI boundary and the host logic (no kernel is launched)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import brepgen_amd as bga
from brepgen_amd import _lib
from oracle import denoisers as orc
from oracle.schedulers import OracleDDPM, OraclePNDM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "brepgen_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/brepgen_hip.h but not exported"
    assert set(names) == set(_lib.EXPORTS)
    assert lib.bg_abi_version() == _lib.ABI_VERSION


def test_argument_errors_are_negative_and_explained():
    lib = _lib.load()
    rc = lib.bg_gemm_bias_act_fwd(None, 0, None, None, None, 0, 1, 1, 1, 1, 0, 0, 0, None, 0, 1, None)
    assert rc == -1 and b"null" in lib.bg_last_error()
    rc = lib.bg_cfg_ddpm_step(None, None, 0.0, None, None, None, 0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, None)
    assert rc == -1
    w, i = _lib.DenoiserWeights(), _lib.DenoiserInputs()
    w.net = 7
    assert lib.bg_denoiser_fwd(C.byref(w), C.byref(i), None, None, 0, None) < 0


def test_gemm_ex_rejects_inconsistent_epilogue_options():
    """bg_gemm_ex_fwd validates the split-residual / LayerNorm-fold combinations before anything is launched."""
    lib = _lib.load()
    fake = 0x10000                                   # aligned, never dereferenced: validation fails first

    def desc(**kw):
        d = _lib.GemmDesc()
        d.a, d.lda, d.w, d.bias, d.out, d.ldc = fake, 768, fake, fake, fake, 768
        d.M, d.N, d.N_pad, d.K = 256, 768, 768, 768
        d.ab_dtype, d.out_dtype, d.act = _lib.BG_BF16, _lib.BG_BF16, 0
        d.add_div = d.add2_div = 1
        d.ln_eps = 1e-5
        for k, v in kw.items():
            setattr(d, k, v)
        return d

    bad = [desc(stats_in=fake),                                  # fold without column sums
           desc(stats_in=fake, colsum=fake, out_lo=fake),        # fold and split output together
           desc(stats_in=fake, colsum=fake, out_dtype=_lib.BG_F32),
           desc(res_hi=fake, res_lo=fake, ld_res=768),           # split residual needs a split output
           desc(out_lo=fake, res_hi=fake, ld_res=768),           # hi plane without lo plane
           desc(stats_out=fake),                                 # row statistics only with a split output
           desc(out_lo=fake, out_dtype=_lib.BG_F32),
           desc(out_lo=fake, N=700)]                             # N != N_pad
    for d in bad:
        rc = lib.bg_gemm_ex_fwd(C.byref(d), None)
        assert rc < 0, rc
        assert lib.bg_last_error()
    d = desc(out_lo=fake, ab_dtype=_lib.BG_F32, out_dtype=_lib.BG_F32)
    assert lib.bg_gemm_ex_fwd(C.byref(d), None) == -4            # BG_E_DTYPE: 16-bit operands only
    assert lib.bg_embed_ln_silu_fwd(fake, 5, 4, 5, fake, fake, fake, fake, fake, _lib.BG_BF16, 1e-5, None) == -2   # k = 5


def test_fused_qkv_attention_validates_before_launching():
    """bg_qkv_attn_fwd covers unmasked batches of equally long sequences of an EVEN length <= 64 with 16-bit operands and the
    LayerNorm-fold operands present; everything else is an argument error (the two-launch path serves it), never a launch."""
    lib = _lib.load()
    fake = 0x10000
    f = lambda **kw: lib.bg_qkv_attn_fwd(kw.get("x", fake), fake, kw.get("bias", fake), kw.get("colsum", fake), kw.get("stats", fake),
                                         None, fake, None, kw.get("B", 4), kw.get("N", 60), kw.get("dtype", _lib.BG_BF16), 1e-5, None)
    for kw in (dict(N=33), dict(N=66), dict(N=0), dict(dtype=_lib.BG_F32), dict(stats=None), dict(colsum=None), dict(bias=None),
               dict(stats=fake + 8)):
        assert f(**kw) == -2 and lib.bg_last_error(), kw                 # BG_E_SHAPE
    assert f(x=None) == -1                                                # BG_E_ARG
    assert f(x=fake + 2) == -5                                            # BG_E_ALIGN


def test_workspace_bytes_scales_with_tokens():
    lib = _lib.load()
    a = lib.bg_workspace_bytes(_lib.BG_SURFZ, 512, 60, 1, _lib.BG_BF16)
    b = lib.bg_workspace_bytes(_lib.BG_SURFZ, 1024, 60, 1, _lib.BG_BF16)
    assert 0 < a < b < 2.2 * a
    # X fp32 + H bf16 + QKV bf16 = 3072 + 1536 + 4608 bytes per token, plus small buffers
    assert a >= 512 * 60 * 9216
    assert lib.bg_workspace_bytes(_lib.BG_EDGEZ, 2, 60, 30, _lib.BG_F32) > lib.bg_workspace_bytes(
        _lib.BG_EDGEZ, 2, 60, 30, _lib.BG_BF16)
    assert lib.bg_workspace_bytes(_lib.BG_SURFZ, 0, 60, 1, _lib.BG_BF16) == 0


@pytest.mark.parametrize("net", ["SurfPosNet", "SurfZNet", "EdgePosNet", "EdgeZNet"])
@pytest.mark.parametrize("use_cf", [False, True])
def test_state_dict_keys_match_reference_layout(net, use_cf):
    m = getattr(bga, net)(use_cf)
    spec = orc.state_dict_spec(net, use_cf)
    sd = m.state_dict()
    assert set(sd) == set(spec)
    for k, shape in spec.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    m.load_state_dict(orc.seeded_state_dict(net, 3, use_cf), strict=True)


def test_no_cpu_fallback():
    m = bga.SurfPosNet(False)
    with pytest.raises(_lib.BrepgenHipError):
        m(torch.zeros(1, 4, 6), torch.tensor([5]), None)
    s = bga.DDPMScheduler()
    with pytest.raises(_lib.BrepgenHipError):
        s.step(torch.zeros(2, 3), 5, torch.zeros(2, 3))


def test_training_and_pipeline_helpers_have_no_cpu_fallback_either():
    from brepgen_amd import training
    with pytest.raises(_lib.BrepgenHipError):
        training.masked_mse(torch.zeros(2, 3, 6), torch.zeros(2, 3, 6))
    enc = bga.AutoencoderKL1DFastEncode(in_channels=3, out_channels=3, down_block_types=["DownBlock1D"] * 3,
                                        up_block_types=["UpBlock1D"] * 3, block_out_channels=[128, 256, 512],
                                        layers_per_block=2, act_fn="silu", latent_channels=3, norm_num_groups=32, sample_size=512)
    with pytest.raises(_lib.BrepgenHipError):
        enc.encode_tokens(torch.zeros(1, 2, 32, 3))


def test_scheduler_host_logic_matches_oracle():
    d, o = bga.DDPMScheduler(clip_sample=True, clip_sample_range=3), OracleDDPM(clip_sample_range=3)
    for n in (1000, 50):
        d.set_timesteps(n)
        o.set_timesteps(n)
        assert d.timesteps.tolist() == o.timesteps.tolist()
        for t in (int(d.timesteps[0]), int(d.timesteps[len(d.timesteps) // 2]), 0):
            c, r = d._coefficients(t), o.coefficients(t)
            assert c["x0_coeff"] == r["x0_coeff"] and c["xt_coeff"] == r["xt_coeff"] and c["sigma"] == r["sigma"]
            assert c["sqrt_alpha_prod"] == r["sqrt_alpha_prod_t"] and c["sqrt_beta_prod"] == r["sqrt_beta_prod_t"]
    p, q = bga.PNDMScheduler(), OraclePNDM()
    p.set_timesteps(200)
    q.set_timesteps(200)
    assert p.timesteps.tolist() == q.timesteps.tolist() and len(p.timesteps) == 209
    assert p._prev_coeffs(980, 975) == tuple(float(v) for v in q.prev_sample_coeffs(980, 975))
    assert p._prev_coeffs(0, -5) == tuple(float(v) for v in q.prev_sample_coeffs(0, -5))
    assert d.config.num_train_timesteps == 1000 and len(d) == 1000


def test_randn_tensor_seed_semantics():
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    a = bga.randn_tensor((4, 30, 6), generator=g1)
    b = torch.randn(4, 30, 6, generator=g2)
    assert torch.equal(a, b) and a.device.type == "cpu"
    gs = [torch.Generator().manual_seed(i) for i in range(3)]
    c = bga.randn_tensor((3, 5), generator=gs)
    assert torch.equal(c[1:2], torch.randn(1, 5, generator=torch.Generator().manual_seed(1)))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "brepgen_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
    # tools/ (micro-benchmarks, profiling helpers) stays clear of the checker as well; bench.py may use it in its
    # cpu_baseline leg only, __graft_entry__ in smoke() only
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith(".py"):
            src = open(os.path.join(ROOT, "tools", f)).read()
            assert not re.search(r"^\s*(from|import)\s+(oracle|parity_cases)", src, flags=re.M), f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    for m in re.finditer(r"^(\s*)(from|import)\s+oracle", bench, flags=re.M):
        assert len(m.group(1)) > 0 and "def cpu_baseline" in bench[:m.start()], "oracle import outside cpu_baseline()"


def test_p256_kernels_do_not_spill():
    """csrc/gemm_p256.hip keeps LDS-DMA in flight across its whole K loop with hand-counted vmcnt waits; a register spill would
    put scratch loads (vector-memory operations hipcc follows with vmcnt(0)) into that stream -- the pipeline would drain at
    every reload, and a spilled destination of the split epilogue's inline-asm residual loads would be read before it lands.
    The build must therefore stay spill-free: 0 bytes of scratch for the plain / LayerNorm-fold kernels, at most three
    loop-invariant dwords for the split-residual ones (one before round 6's -fno-slp-vectorize build; the reloads sit in the epilogue,
    and the scan below checks that none of them is a destination of the asm loads)."""
    import re
    import subprocess
    from brepgen_amd import build as b
    src = os.path.join(b.CSRC, "gemm_p256.hip")
    r = subprocess.run([b._hipcc(), *b.FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    scratch = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    assert len(names) == len(scratch) and len(names) >= 8              # {bf16, fp16} x {plain, fold, split, split + stats}
    for n, sc in zip(names, scratch):
        if "gemm16_p256_kernel" not in n:
            continue
        split = "ELi3E" in n                                     # MODE = P_SPLIT in the mangled name
        assert sc <= (16 if split else 0), (n, sc)
    # ... and the one dword the split-residual kernels may spill must not be a destination of their inline-asm residual loads (those
    # are invisible to hipcc's wait insertion: a spilled or reloaded destination would be read before the counted wait covers it).
    # In the -S output the asm loads sit between ;;#ASMSTART / ;;#ASMEND markers; every scratch access names its register.
    s_out = subprocess.run([b._hipcc(), *b.FLAGS, "-S", "--cuda-device-only", "-c", src, "-o", "-"], capture_output=True, text=True)
    assert s_out.returncode == 0, s_out.stderr[-2000:]
    for fn in re.split(r"\n(?=_ZN2bg18gemm16_p256_kernel)", s_out.stdout):
        if not fn.startswith("_ZN2bg18gemm16_p256_kernel"):
            continue
        asm_dst = set()
        for blk in re.findall(r";;#ASMSTART(.*?);;#ASMEND", fn, flags=re.S):
            for lo, hi in re.findall(r"global_load_dwordx4 v\[(\d+):(\d+)\]", blk):
                asm_dst.update(range(int(lo), int(hi) + 1))
        scr = set()
        for lo, hi in re.findall(r"scratch_(?:store|load)_dword\w* (?:off, )?v\[?(\d+)(?::(\d+))?", fn):
            scr.update(range(int(lo), int(hi or lo) + 1))
        assert not (asm_dst & scr), (fn[:60], sorted(asm_dst & scr))


@pytest.mark.parametrize("src", ["gemm_split.hip", "qkv_attn.hip", "ffn_fused.hip"])
def test_hand_scheduled_kernels_do_not_spill(src):
    """csrc/gemm_split.hip runs at the 256-VGPR limit of two waves per SIMD (two accumulator sets + the slab in flight),
    csrc/qkv_attn.hip keeps 96 accumulators + 80 fragment registers + the fold temporaries live in its K loop; a spill would put
    scratch traffic (and hipcc's vmcnt(0) after every reload) into K-steps whose waits are placed by hand; csrc/ffn_fused.hip streams
    its weights into registers several k-slices ahead of their use (a reload's vmcnt(0) would wait for all of them)."""
    import re
    import subprocess
    from brepgen_amd import build as b
    r = subprocess.run([b._hipcc(), *b.FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(b.CSRC, src),
                        "-o", os.devnull], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    scratch = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    spills = [int(v) for v in re.findall(r"VGPRs Spill: (\d+)", r.stderr)]
    assert len(scratch) >= 2 and all(v == 0 for v in scratch) and all(v == 0 for v in spills), (scratch, spills)


def test_gemm_partition_rule_between_the_256_and_128_kernels():
    """bg_gemm_p256_rows is the rule both GEMM kernels evaluate on the device-side row count (csrc/bg_common.h p256_rows):
    the 256 x 256 kernel takes whole 256-row panels from the front, never more than the launch has, whole tile rounds
    unless the last round is well filled; the known answers are the face-LDM shapes of DESIGN.md section 4."""
    lib = _lib.load()
    f = lib.bg_gemm_p256_rows
    assert f(-1, 2304, 0, 0) == -1 and f(100, 100, 0, 0) == -1
    for n_cols in (768, 1024, 2304):
        nt = n_cols // 256
        for split in (0, 1):
            for conc in (0, 1):
                for rows in list(range(0, 1200, 37)) + [8640, 15360, 17280, 17293, 30720, 30797, 61440, 138752, 1 << 20]:
                    p = f(rows, n_cols, split, conc)
                    panels = (rows + 255) // 256
                    assert p % 256 == 0 and 0 <= p <= panels * 256, (rows, n_cols, split, conc, p)
                    if conc:
                        assert p in (0, panels * 256)                      # all or nothing next to sibling launches
                    elif 0 < p < panels * 256:
                        assert (p // 256 * nt) <= (panels * nt) // 256 * 256   # a partial take never starts a round it cannot fill
    # QKV of the compacted 512 x 60 face batch (17 280 rows, 9 column tiles): 68 panels = 612 tiles -> two full rounds (56 panels)
    # on the 256 kernel, 12 panels on the 128 kernel; the dense batch (30 720 rows = 1080 tiles, last round 56 tiles): 4 rounds
    assert f(17280, 2304, 0, 0) == 56 * 256 and f(30720, 2304, 0, 0) == (1024 // 9) * 256
    assert f(17280, 2304, 0, 1) == 68 * 256 and f(8640, 2304, 0, 1) == 34 * 256      # concurrent sample groups: >= 300 tiles -> alone
    assert f(7680, 2304, 0, 1) == 0                                           # 270 tiles: a round and a sliver -> the 128 kernel
    # out-proj / FFN2 (3 column tiles, rounds of 255 tiles): 204 tiles (the compacted face batch) fill one round well enough for the
    # 256 kernel, 360 tiles (1.4 rounds) stay on the pipelined 128 x 128 kernel, 720 tiles (2.8 rounds, the last one 210 tiles full) go to the 256 kernel whole, the edge nets (1626 tiles = 6 rounds
    # + 96) give it its six full rounds = 510 panels
    assert f(17280, 768, 1, 0) == 68 * 256 and f(30720, 768, 1, 0) == 0 and f(61440, 768, 1, 0) == 240 * 256
    assert f(138752, 768, 1, 0) == 510 * 256 and f(138752, 768, 1, 1) == 542 * 256 and f(40960, 768, 1, 1) == 0
    assert f(100, 768, 0, 0) == 0 and f(0, 2304, 0, 0) == 0                 # a few tiles: the 128 kernel (finer tiles fill more CUs)


def test_host_side_row_plan_helpers():
    """network._slot_rows restates compact.hip: pair_slots_kernel on the host (the exact slot-packed row count the launcher plans with);
    network._group_ranges restates bg_denoiser_fwd's contiguous sample groups (csrc/denoiser.hip: split_range)."""
    import random
    from brepgen_amd.network import _HipDenoiser as H

    def slots_reference(n):                                       # the kernel's walk, written out (tests/test_gpu_round4.py checks it on the device)
        order = sorted(range(len(n)), key=lambda i: (n[i], i))
        i, j, k = 0, len(n) - 1, 0
        while i < len(n) and n[order[i]] == 0:
            i += 1
        while i <= j:
            if i < j and n[order[i]] + n[order[j]] <= 64:
                i += 1
            j -= 1
            k += 1
        return 64 * k

    rng = random.Random(7)
    for _ in range(300):
        n = [rng.randint(0, 64) for _ in range(rng.randint(1, 200))]
        assert H._slot_rows(n) == slots_reference(n)
    assert H._slot_rows([0, 0]) == 0 and H._slot_rows([64]) == 64 and H._slot_rows([32, 32, 32]) == 128
    for B in (1, 2, 3, 5, 16, 511, 512, 513):
        for ns in (0, 1, 2, 3, 4, 7):
            groups = list(H._group_ranges(B, ns))
            want = min(ns, 4) if (min(ns, 4) >= 2 and B >= min(ns, 4)) else 1
            assert len(groups) == want and groups[0][0] == 0 and groups[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(groups, groups[1:]))
            sizes = [hi - lo for lo, hi in groups]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_driver_entry_build_runs():
    """__graft_entry__.build() is what the driver runs on CPU every round: it must agree with the header's ABI version (a stale literal
    there once survived an ABI bump) and resolve every export."""
    import re
    import __graft_entry__ as entry
    from brepgen_amd import _lib
    entry.build()
    header = open(os.path.join(ROOT, "include", "brepgen_hip.h")).read()
    assert int(re.search(r"#define BG_ABI_VERSION (\d+)", header).group(1)) == _lib.ABI_VERSION


def test_ctypes_structs_match_the_header(tmp_path):
    """Every struct of include/brepgen_hip.h that brepgen_amd/_lib.py mirrors: sizeof and the offset of every field, as gcc lays the
    header out, against ctypes -- the check a maintainer of another binding would run after an ABI bump."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this box")
    pairs = {"bg_mlp_weights": _lib.MlpWeights, "bg_layer_weights": _lib.LayerWeights, "bg_denoiser_weights": _lib.DenoiserWeights,
             "bg_denoiser_inputs": _lib.DenoiserInputs, "bg_gemm_desc": _lib.GemmDesc, "bg_conv_desc": _lib.ConvDesc,
             "bg_vae_op": _lib.VaeOp, "bg_profile_row": _lib.ProfileRow}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "brepgen_hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'    printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, *_ in cls._fields_:
            lines.append(f'    printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['    return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    r = subprocess.run([gcc, "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr                            # (a field _lib.py names that the header does not have fails HERE)
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(out[cname]) == C.sizeof(cls), (cname, out[cname], C.sizeof(cls))
        for fname, *_ in cls._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)


def test_scheduler_step_takes_a_device_timestep_without_reading_it():
    """sample.py iterates `scheduler.timesteps` -- on the device if the caller moved them there -- and hands each 0-d tensor to
    step(): `int()` of a device tensor is a synchronisation per step.  The schedule is host state, so the product takes a device
    timestep from it by position.  A `meta` tensor stands in for the device tensor here: reading it would raise."""
    import torch
    from brepgen_amd.schedulers import DDPMScheduler, PNDMScheduler
    d = DDPMScheduler(num_train_timesteps=1000)
    d.set_timesteps(1000)
    dev_t = torch.empty((), dtype=torch.int64, device="meta")
    assert [d._timestep(dev_t) for _ in range(3)] == [999, 998, 997]
    assert d._timestep(torch.tensor(500)) == 500 and d._timestep(dev_t) == 499      # a host timestep re-seats the cursor
    assert d._timestep(7) == 7 and d._timestep(dev_t) == 6
    d.set_timesteps(50)
    assert [d._timestep(dev_t) for _ in range(2)] == [980, 960]
    for _ in range(48):
        d._timestep(dev_t)
    assert d._timestep(dev_t) == 980                                                 # wraps for the next sampling run
    p = PNDMScheduler(num_train_timesteps=1000)
    p.set_timesteps(200)
    assert p._host_ts[:5] == [int(v) for v in p.timesteps[:5]] and len(p._host_ts) == 209


def test_allgather_entry_rejects_a_null_communicator():
    """bg_allgather (the path's one collective for hosts without torch.distributed) is exported and checks its arguments before it
    touches RCCL: no communicator -> BG_E_ARG; zero bytes -> nothing to do."""
    from brepgen_amd import _lib
    lib = _lib.load()
    assert lib.bg_allgather(None, None, 16, None, None) == -1
    assert b"communicator" in lib.bg_last_error()
    assert lib.bg_allgather(None, None, 0, 1, None) == 0


def test_no_kernel_contains_op_sel_modified_packed_fp32():
    """gfx950 hazard found in round 6 (brepgen_amd/build.py: FLAGS): packed-fp32 VALU instructions with op_sel modifiers compute wrong
    values in lanes 48-63 while another wave on the SIMD mixes MFMA with LDS-DMA -- which is what every 16-bit GEMM / attention kernel
    of this library does, on a second stream whenever sample groups (n_split) or the two VAE decodes run concurrently.  The build
    therefore must not contain such instructions: scan the ISA of every source."""
    import re
    import subprocess
    from brepgen_amd import build as b
    assert "-fno-slp-vectorize" in b.FLAGS
    bad = {}
    for src in b.SOURCES:
        r = subprocess.run([b._hipcc(), *b.FLAGS, *b.PER_FILE_FLAGS.get(src, []), "-S", "--cuda-device-only", "-c", os.path.join(b.CSRC, src), "-o", "-"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        hits = [l.strip() for l in r.stdout.splitlines() if re.search(r"\bv_pk_\w+_f32\b.*\bop_sel", l)]
        if hits:
            bad[src] = hits[:3]
    assert not bad, bad
