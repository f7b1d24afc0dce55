"""-m gpu: round-4 additions -- the software-pipelined split-residual GEMM (csrc/gemm_split.hip) against the kernels it replaces
(bit for bit: out of place, in place, ragged row counts, device-side row counts inside the denoiser), and full-size parity cases
the round-3 review asked for (EdgeZNet at the ABC shape, guided EdgePosNet).  Measured numbers -> gpurun_out/parity_r04.json."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

F32, F16, BF16 = torch.float32, torch.float16, torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pc():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import parity_cases
    return parity_cases


def _record(key, value):
    path = os.path.join(ROOT, "gpurun_out", "parity_r04.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        d = {}
    d[key] = value
    with open(path, "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)


@pytest.fixture
def tune():
    """bg_tune_set with automatic reset of the keys this file touches (8: phase-group delay, 10: 256-kernel mode, 12: split-kernel
    choice, 13: fused QKV + attention, 15: small-launch threshold)."""
    from brepgen_amd import _lib
    lib = _lib.load()
    yield lib.bg_tune_set
    for k in (8, 10, 12, 13, 15):
        lib.bg_tune_set(k, 0)


def _split_case(M, K, dt, seed=0):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(M, 768) * 2
    hi = x.to(dt)
    lo = (x - hi.float()).to(dt)
    a = (rn(M, K) * 0.5).to(dt).cuda()
    w, b = (rn(768, K) * 0.04).to(dt).cuda(), rn(768).cuda()
    return a, w, b, hi.cuda(), lo.cuda()


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("K", [768, 1024])
@pytest.mark.parametrize("M", [1409, 128 * 11, 4999, 17294, 30720 + 78])
def test_split_pipe_gemm_is_bit_identical(pc, tune, dt, K, M):
    """out-proj (K = 768) and FFN2 (K = 1024) with split residual + row statistics: the pipelined kernel (key 12 = 0) against the
    256 + 128 hybrid (key 12 = 1) and the 128 x 128 kernel alone (key 12 = 1, key 10 = 2): hi, lo and the statistics, out of place
    and in place.  M covers one tile per workgroup (1409: 12 panels), many tiles per workgroup, ragged last panels.  Repeated: a race
    between the K loop and the epilogue it carries would not necessarily show the first time."""
    import hip_ops as ops
    a, w, b, hi, lo = _split_case(M, K, dt)

    def run(inplace):
        if inplace:
            h, l = hi.clone(), lo.clone()
            r = ops.linear_ex(a, w, b, split_out=True, res=(h, l), want_stats=True, inplace=True)
        else:
            r = ops.linear_ex(a, w, b, split_out=True, res=(hi, lo), want_stats=True)
        torch.cuda.synchronize()
        return r["out"].clone(), r["lo"].clone(), r["stats"].clone()

    tune(15, -1)                                                  # (M = 1409 is 72 tiles: keep it off the small-launch path)
    tune(12, 1)
    tune(10, 2)
    ref = run(False)
    tune(10, 0)
    hyb = run(False)
    assert all(torch.equal(x, y) for x, y in zip(ref, hyb))
    tune(12, 0)
    for rep in range(3):
        for inplace in (False, True):
            got = run(inplace)
            for name, x, y in zip(("hi", "lo", "stats"), ref, got):
                assert torch.equal(x, y), (name, M, K, dt, inplace, rep, int((x != y).sum()))


@pytest.mark.parametrize("n_split", [1, 2])
def test_split_pipe_inside_the_denoiser_with_device_side_row_counts(pc, tune, n_split):
    """SurfZNet at the headline shape (512 x 60, ragged mask -> the compacted row count only exists on the device), dense and
    variable-length, one and two sample groups in flight: eps with the pipelined residual-stream GEMMs == eps with the kernels of
    round 3, bit for bit."""
    for varlen in (True, False):
        m, _ = pc.build_net("SurfZNet", 5, False, BF16, varlen=varlen)
        m.n_split = n_split
        args = [a.cuda() if torch.is_tensor(a) else a for a in pc.synth_inputs("SurfZNet", 512, 60, 1, False)]
        with torch.no_grad():
            tune(12, 1)
            ref = m(*args).clone()
            tune(12, 0)
            for _ in range(2):
                got = m(*args)
                torch.cuda.synchronize()
                assert torch.isfinite(ref).all() and torch.equal(ref, got), (varlen, n_split)


def test_split_pipe_edge_net_shape_fp16(pc, tune):
    """EdgeZNet, 8 x 60 x 40 with a ragged edge mask, fp16 (the furniture cascade's dtype): ~13 tiles per workgroup."""
    m, _ = pc.build_net("EdgeZNet", 9, False, F16, varlen=True)
    args = [a.cuda() if torch.is_tensor(a) else a for a in pc.synth_inputs("EdgeZNet", 8, 60, 40, False)]
    with torch.no_grad():
        tune(12, 1)
        ref = m(*args).clone()
        tune(12, 0)
        got = m(*args)
    assert torch.isfinite(ref).all() and torch.equal(ref, got)


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("M", [16 * 60, 960 + 37, 128])
def test_small_launch_tiles_are_bit_identical(pc, tune, dt, M):
    """The reference's shipped batch (16 samples x 60 faces = 960 tokens): launches of < 160 tiles of 128 x 128 run on 64 x 64 tiles
    (key 15 = -1 switches that off).  Every epilogue the encoder layers use -- plain, LayerNorm fold (+ ReLU), split residual +
    statistics -- must not depend on the tile shape: a sample's bits are the same at batch 16 and at batch 512."""
    import hip_ops as ops
    g = torch.Generator().manual_seed(3)
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(M, 768) * 2
    hi = x.to(dt)
    lo = (x - hi.float()).to(dt).cuda()
    hi = hi.cuda()
    grp = x.reshape(M, 12, 64)
    stats = torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2).contiguous().cuda()
    cases = {}
    for name, N, act in (("qkv", 2304, 0), ("ffn1", 1024, 1)):
        w, b = (rn(N, 768) * 0.04).to(dt).cuda(), rn(N).cuda()
        cs = w.float().sum(1).contiguous()
        cases[name + " plain"] = lambda w=w, b=b, act=act: ops.linear(hi, w, b, out_dtype=dt, act=act)
        cases[name + " fold"] = lambda w=w, b=b, act=act, cs=cs: ops.linear_ex(hi, w, b, act=act, stats_in=stats, colsum=cs)["out"]
    for name, K in (("outproj", 768), ("ffn2", 1024)):
        a = (rn(M, K) * 0.5).to(dt).cuda()
        w, b = (rn(768, K) * 0.04).to(dt).cuda(), rn(768).cuda()

        def split(a=a, w=w, b=b):
            r = ops.linear_ex(a, w, b, split_out=True, res=(hi, lo), want_stats=True)
            return torch.cat([r["out"].float().flatten(), r["lo"].float().flatten(), r["stats"].flatten()])
        cases[name + " split"] = split
    for name, fn in cases.items():
        tune(15, -1)
        ref = fn().clone()
        tune(15, 0)
        for _ in range(2):
            got = fn()
            torch.cuda.synchronize()
            assert torch.equal(ref, got), (name, M, dt)


def test_p256_split_phase_groups_are_bit_identical(pc, tune):
    """out-proj / FFN2 at the edge nets' row count on the 256 x 256 kernel alone (key 10 = 1), with the second phase group starting
    late (key 8 = 24 x 1024 cycles, the default), much later (200) and not at all (-1): the delay moves when a tile is computed, never
    what it computes."""
    import hip_ops as ops
    for K in (768, 1024):
        a, w, b, hi, lo = _split_case(138752 + 5, K, BF16, seed=K)
        outs = []
        for mode, stag in ((2, 0), (1, -1), (1, 0), (1, 200), (0, 0)):
            tune(12, 1 if mode == 2 else 0)
            tune(10, mode)
            tune(8, stag)
            h, l = hi.clone(), lo.clone()
            r = ops.linear_ex(a, w, b, split_out=True, res=(h, l), want_stats=True, inplace=True)
            torch.cuda.synchronize()
            outs.append((r["out"].clone(), r["lo"].clone(), r["stats"].clone()))
        for o in outs[1:]:
            assert all(torch.equal(x, y) for x, y in zip(outs[0], o)), K


# ---- QKV + attention in one launch (csrc/qkv_attn.hip) ------------------------------------------------------------------
def _qkv_case(B, N, dt, seed):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    M = B * N
    x = rn(M, 768) * 2
    grp = x.reshape(M, 12, 64)
    stats = torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2).contiguous().cuda()
    w = rn(2304, 768) * 0.04
    w[:768] *= 0.125                                              # (the weight packer folds 1 / sqrt(d_head) into the q rows)
    w = w.to(dt).cuda()
    return x.to(dt).cuda(), w, rn(2304).cuda(), w.float().sum(1).contiguous(), stats


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("B,N", [(1, 60), (4, 60), (7, 60), (131, 60), (512, 60), (3, 30), (8, 30), (13, 30), (512, 30), (5, 64), (9, 32), (6, 2), (6, 34), (300, 48)])
def test_fused_qkv_attention_is_bit_identical_to_the_two_launches(pc, dt, B, N):
    """LayerNorm-fold QKV GEMM + attention as ONE launch (the face LDM's SurfPosNet: 30 / 60 tokens, no mask) against
    bg_gemm_ex_fwd + bg_attn_fwd: the q|k|v image the fused kernel keeps in LDS and the attention output, bit for bit -- partial
    last sample groups (B not a multiple of 4 / 8), every slot count (N = 2 .. 64, both slot sizes), several tiles per workgroup
    (B = 512: 6 rounds).  Repeated: a race between the ring and the epilogue images would not necessarily show the first time."""
    import hip_ops as ops
    a, w, b, cs, stats = _qkv_case(B, N, dt, B * 100 + N)
    qkv = ops.linear_ex(a, w, b, stats_in=stats, colsum=cs)["out"]
    ref = ops.attention(qkv, None, B, N)
    for _ in range(3):
        out, img = ops.qkv_attention(a, w, b, cs, stats, B, N, want_qkv=True)
        out2 = ops.qkv_attention(a, w, b, cs, stats, B, N)
        torch.cuda.synchronize()
        assert torch.equal(img, qkv), (B, N, int((img.float() != qkv.float()).sum()))
        assert torch.equal(out, ref) and torch.equal(out2, ref), (B, N, int((out.float() != ref.float()).sum()))
    assert torch.isfinite(ref.float()).all()


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("B,N", [(9, 60), (512, 60), (16, 30), (64, 64), (10, 34)])
def test_fused_qkv_attention_with_a_key_padding_mask(pc, dt, B, N):
    """The dense execution of a masked net (SurfZNet with varlen off): ragged key-padding masks, one sample with every key padded
    (its outputs are 0 in both paths), one with none."""
    import hip_ops as ops
    a, w, b, cs, stats = _qkv_case(B, N, dt, B * 100 + N + 7)
    g = torch.Generator().manual_seed(B + N)
    nv = torch.randint(1, N + 1, (B,), generator=g)
    nv[0], nv[-1] = 0, N
    kp = (torch.arange(N)[None] >= nv[:, None]).cuda()
    qkv = ops.linear_ex(a, w, b, stats_in=stats, colsum=cs)["out"]
    ref = ops.attention(qkv, kp, B, N)
    for _ in range(3):
        out, img = ops.qkv_attention(a, w, b, cs, stats, B, N, want_qkv=True, key_pad=kp)
        out2 = ops.qkv_attention(a, w, b, cs, stats, B, N, key_pad=kp)
        torch.cuda.synchronize()
        assert torch.equal(img, qkv)
        assert torch.equal(out, ref) and torch.equal(out2, ref), (B, N, int((out.float() != ref.float()).sum()))
    assert torch.isfinite(ref.float()).all() and float(ref[:N].float().abs().max()) == 0.0


def test_fused_qkv_attention_rejects_what_it_does_not_cover(pc):
    import hip_ops as ops
    a, w, b, cs, stats = _qkv_case(4, 33, BF16, 1)                 # odd N: the statistics travel two rows per element
    with pytest.raises(RuntimeError):
        ops.qkv_attention(a, w, b, cs, stats, 4, 33)


@pytest.mark.parametrize("n_split", [1, 2])
def test_fused_qkv_attention_inside_the_densely_executed_masked_net(pc, tune, n_split):
    """SurfZNet with variable-length execution off (every padded position computed, as the reference does): the fused launch takes
    the key-padding mask; eps == eps of GEMM + attention, bit for bit."""
    m, _ = pc.build_net("SurfZNet", 5, False, BF16, varlen=False)
    m.n_split = n_split
    args = [a.cuda() if torch.is_tensor(a) else a for a in pc.synth_inputs("SurfZNet", 512, 60, 1, False)]
    with torch.no_grad():
        tune(13, 1)
        ref = m(*args).clone()
        tune(13, 0)
        for _ in range(2):
            got = m(*args)
            torch.cuda.synchronize()
            assert torch.isfinite(ref).all() and torch.equal(ref, got), n_split


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("B,S,n_split", [(512, 60, 1), (512, 60, 2), (512, 30, 1), (300, 60, 2)])
def test_fused_qkv_attention_inside_the_denoiser(pc, tune, dt, B, S, n_split):
    """SurfPosNet at the face LDM's shapes: eps with the fused launch == eps with GEMM + attention (key 13 = 1), bit for bit,
    one and two sample groups in flight."""
    m, _ = pc.build_net("SurfPosNet", 11, False, dt)
    m.n_split = n_split
    args = [a.cuda() if torch.is_tensor(a) else a for a in pc.synth_inputs("SurfPosNet", B, S, 1, False)]
    with torch.no_grad():
        tune(13, 1)
        ref = m(*args).clone()
        tune(13, 0)
        for _ in range(2):
            got = m(*args)
            torch.cuda.synchronize()
            assert torch.isfinite(ref).all() and torch.equal(ref, got), (B, S, n_split)


# ---- ragged batches through the fused launch: slot-packed compaction (csrc/compact.hip) + the PAIR variant of qkv_attn.hip ------
def _pair_slots_reference(n):
    """The pairing of compact.hip: pair_slots_kernel, restated: ascending (length, index); the shortest unpaired sample joins the
    longest one while their sum fits 64; samples without a valid token own no rows."""
    B = len(n)
    order = sorted(range(B), key=lambda i: (n[i], i))
    i, j, slots, start = 0, B - 1, [], [0] * B
    while i < B and n[order[i]] == 0:
        i += 1
    while i <= j:
        a, na, b, nb = order[j], n[order[j]], -1, 0
        if i < j and n[order[i]] + na <= 64:
            b, nb = order[i], n[order[i]]
            start[b] = 64 * len(slots) + na
            i += 1
        start[a] = 64 * len(slots)
        slots.append((a, na, b, nb))
        j -= 1
    return slots, start


def _ragged_lengths(B, N, seed, lo=0):
    g = torch.Generator().manual_seed(seed)
    n = torch.randint(lo, N + 1, (B,), generator=g)
    n[0], n[-1] = N, max(lo, 1)
    return n


# (B > 1024: the multi-pass `base += 1024` loops, the cross-wave histogram prefix and the 64-rank batches of the walk; 8192 = the bound)
@pytest.mark.parametrize("B,N", [(512, 60), (37, 64), (5, 8), (300, 30), (1025, 60), (4096, 33), (8192, 60)])
def test_slot_packed_compaction_matches_its_restatement(pc, B, N):
    from brepgen_amd import _lib
    n = _ragged_lengths(B, N, B + N)
    mask = (torch.arange(N)[None] >= n[:, None])
    perm_mask = mask.clone()
    for b in range(B):                                            # valid tokens anywhere in the row, not only at its front
        perm_mask[b] = mask[b][torch.randperm(N, generator=torch.Generator().manual_seed(b))]
    m_dev = perm_mask.to(torch.uint8).cuda()
    offs = torch.zeros(B + 1, dtype=torch.int32, device="cuda")
    src = torch.full((64 * B,), -1, dtype=torch.int32, device="cuda")
    sd = torch.zeros(2 * B, dtype=torch.int32, device="cuda")
    sa = torch.zeros(B, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    _lib.check(_lib.load().bg_compact_rows_paired(m_dev.data_ptr(), B, N, offs.data_ptr(), src.data_ptr(), sd.data_ptr(), sa.data_ptr(),
                                                  cnt.data_ptr(), _lib.stream()), "bg_compact_rows_paired")
    torch.cuda.synchronize()
    slots, start = _pair_slots_reference(n.tolist())
    assert int(offs[B]) == 64 * len(slots)
    assert cnt.cpu().tolist() == n.tolist()
    sdc, sac, offc, srcc = sd.cpu().tolist(), sa.cpu().tolist(), offs.cpu().tolist(), src.cpu().tolist()
    pm = perm_mask.tolist()                                       # (plain lists: B = 8192 would otherwise index a tensor half a million times)
    for k, (a, na, b, nb) in enumerate(slots):
        assert (sdc[2 * k], sdc[2 * k + 1], sac[k]) == (na, nb, a), k
        valid_a = [a * N + i for i in range(N) if not pm[a][i]]
        rows = valid_a + ([b * N + i for i in range(N) if not pm[b][i]] if b >= 0 else [])
        rows += [valid_a[0]] * (64 - len(rows))                   # clones of the slot's first token
        assert srcc[64 * k:64 * k + 64] == rows, k
    for b in range(B):
        if n[b] > 0:
            assert offc[b] == start[b], b
    fill = sum(n.tolist()) / max(1, 64 * len(slots))
    _record(f"slot_packing_fill_B{B}_N{N}", round(fill, 4))


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("B,N,lo", [(512, 60, 8), (64, 64, 1), (9, 60, 1), (130, 30, 0)])
def test_fused_qkv_attention_on_a_slot_packed_ragged_batch(pc, dt, B, N, lo):
    """The rows of a ragged batch, dense-packed through bg_gemm_ex_fwd (LayerNorm fold) + bg_attn_varlen_fwd, against the same samples
    slot-packed (two per 64-row slot where they fit) through the fused launch: every valid token's output bit for bit, and the clone
    rows of a slot equal to the row they clone."""
    import hip_ops as ops
    from brepgen_amd import _lib
    lib = _lib.load()
    n = _ragged_lengths(B, N, 3 * B + N, lo).tolist()
    slots, start = _pair_slots_reference(n)
    off = [0]
    for v in n:
        off.append(off[-1] + v)
    Md, Ms = off[-1], 64 * len(slots)
    g = torch.Generator().manual_seed(B * 7 + N)
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(Md, 768) * 2                                           # the valid tokens, dense-packed
    w = rn(2304, 768) * 0.04
    w[:768] *= 0.125
    w, b = w.to(dt).cuda(), rn(2304).cuda()
    cs = w.float().sum(1).contiguous()

    def stats_of(rows):
        grp = rows.reshape(rows.shape[0], 12, 64)
        return torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2).contiguous().cuda()

    # dense packing: two launches
    Mpad_d = Md + (Md & 1)                                        # (the fold launches take an even row count)
    xd = torch.cat([x, x[-1:]]) if Md & 1 else x
    qkv = ops.linear_ex(xd.to(dt).cuda(), w, b, stats_in=stats_of(xd), colsum=cs)["out"][:Md].contiguous()
    ref = torch.empty(Md, 768, device="cuda", dtype=dt)
    offs_d = torch.tensor(off, dtype=torch.int32, device="cuda")
    _lib.check(lib.bg_attn_varlen_fwd(qkv.data_ptr(), None, ref.data_ptr(), B, N, _lib.BG_F16 if dt == F16 else _lib.BG_BF16,
                                      offs_d.data_ptr(), _lib.stream()), "bg_attn_varlen_fwd")
    # slot packing
    src = []
    for (a, na, bb, nb) in slots:
        rows = list(range(off[a], off[a] + na)) + (list(range(off[bb], off[bb] + nb)) if bb >= 0 else [])
        src += rows + [off[a]] * (64 - len(rows))
    src_t = torch.tensor(src, dtype=torch.long)
    xs = x[src_t]
    cap = Ms + 128                                                # row capacity beyond the rows present (the kernel reads the count on the device)
    xs_dev = torch.zeros(cap, 768, dtype=dt, device="cuda")
    xs_dev[:Ms] = xs.to(dt).cuda()
    st = torch.zeros(12, cap, 2, device="cuda")
    st[:, :Ms] = stats_of(xs)
    sd = torch.tensor([v for (_, na, _, nb) in slots for v in (na, nb)] + [0, 0] * 2, dtype=torch.int32, device="cuda")
    m_dev = torch.tensor([Ms], dtype=torch.int32, device="cuda")
    for rep in range(3):
        out = torch.zeros(cap, 768, device="cuda", dtype=dt)
        img = torch.zeros(cap, 2304, device="cuda", dtype=dt)
        _lib.check(lib.bg_qkv_attn_paired_fwd(xs_dev.data_ptr(), w.data_ptr(), b.data_ptr(), cs.data_ptr(), st.data_ptr(), out.data_ptr(),
                                              img.data_ptr(), m_dev.data_ptr(), sd.data_ptr(), len(slots) + 2, cap,
                                              _lib.BG_F16 if dt == F16 else _lib.BG_BF16, 1e-5, _lib.stream()), "bg_qkv_attn_paired_fwd")
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all()
        assert torch.equal(img[:Ms], qkv[src_t.cuda()]), int((img[:Ms].float() != qkv[src_t.cuda()].float()).sum())
        got, want = out[:Ms], ref[src_t.cuda()]
        bad = (got.float() != want.float()).any(1)
        assert not bad.any(), (B, N, int(bad.sum()), bad.nonzero()[:8].flatten().tolist())
        assert float(out[Ms:].float().abs().max()) == 0.0          # nothing written past the rows present


@pytest.mark.parametrize("n_split", [1, 2])
@pytest.mark.parametrize("dt", [BF16, F16])
def test_slot_packed_execution_inside_the_denoiser(pc, tune, dt, n_split):
    """SurfZNet at the headline shape (512 x 60, ragged mask), variable-length execution: slot-packed rows + the fused launch against
    the dense packing + GEMM + attention (key 13 = 1): eps bit for bit, zeros at the padded positions."""
    m, _ = pc.build_net("SurfZNet", 5, False, dt, varlen=True)
    m.n_split = n_split
    args = [a.cuda() if torch.is_tensor(a) else a for a in pc.synth_inputs("SurfZNet", 512, 60, 1, False)]
    with torch.no_grad():
        tune(13, 1)
        ref = m(*args).clone()
        tune(13, 0)
        for _ in range(2):
            got = m(*args)
            torch.cuda.synchronize()
            assert torch.isfinite(ref).all() and torch.equal(ref, got), n_split
    assert float(ref[args[3]].abs().max()) == 0.0
