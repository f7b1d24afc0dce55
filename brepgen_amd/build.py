"""Build libbrepgen_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake: plain `hipcc -c` per kernel
file (in parallel) and one link.  The .so is git-ignored but travels to the GPU box with the repo snapshot.

    python -m brepgen_amd.build [--force] [--save-temps]
"""
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libbrepgen_hip.so")
SOURCES = ["elementwise.hip", "embed.hip", "gemm_f32.hip", "gemm_16bit.hip", "gemm_p256.hip", "gemm_split.hip", "qkv_attn.hip", "ffn_fused.hip", "out_tail.hip", "attn.hip", "vae.hip", "dedup.hip", "chamfer.hip", "rng.hip", "compact.hip", "vae_exec.hip", "collective.hip", "denoiser.hip"]
PER_FILE_FLAGS = {}
# -fno-slp-vectorize: hipcc's SLP vectoriser turns adjacent scalar fp32 arithmetic into PACKED fp32 VALU instructions with op_sel broadcast
# modifiers (v_pk_add_f32 ... op_sel_hi:[1,0], v_pk_mul_f32 ... op_sel:[0,1]).  On gfx950 (ROCm 7.2) a wave executing those returns WRONG values
# in its lanes 48-63 while another wave on the same SIMD -- of ANY kernel, e.g. this library's GEMMs on a second stream -- mixes MFMA with LDS-DMA
# (global_load_lds): found in round 6 (GroupNorm passes of a VAE decode corrupted by a concurrent VAE decode), reproduced stand-alone by
# tools/pk_f32_mfma_hazard_probe.hip (profiles/r06/pk_f32_mfma_lds_dma_hazard_probe.log).  Without the vectoriser no kernel of the library
# contains such an instruction (tests/test_abi_cpu.py scans the ISA); packed f32 is an anti-lever beside MFMAs anyway (CDNA guide).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libbrepgen_hip.so cannot be built")
    return exe


def _source_digest():
    """Digest of the kernel sources + flags alone (no compiler version): what a box WITHOUT hipcc can still re-derive, to notice a
    prebuilt library that is older than the csrc next to it."""
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + ["../../include/brepgen_hip.h"]
    for f in files:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _digest():
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + ["../../include/brepgen_hip.h"]   # sources only
    for f in files:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update((" ".join(FLAGS) + repr(sorted(PER_FILE_FLAGS.items()))).encode())
    # the compiler is part of the build: the kernels that run at the 256-VGPR limit (gemm_p256.hip, gemm_split.hip) are checked for
    # spills with THIS hipcc (tests/test_abi_cpu.py); another version has to rebuild -- and re-run that check
    # (no hipcc on this box: the digest then covers the sources only -- see build())
    try:
        h.update(subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout.encode())
    except (OSError, RuntimeError):
        return None
    return h.hexdigest()


def _compile(src, extra):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    cmd = [_hipcc(), *FLAGS, *PER_FILE_FLAGS.get(src, []), *extra, "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj, r.stderr


def build(force=False, save_temps=False, verbose=True):
    """Compile every HIP source for gfx950 and link the shared library.  Returns the library path."""
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "digest.txt")
    dig = _digest()
    if dig is None:
        # a box without a compiler: the prebuilt library that travelled with the tree is what there is (its digest cannot be
        # re-derived without hipcc's version string); without it there is nothing to load -- fail loudly
        if os.path.exists(LIB) and not force:
            src_stamp = os.path.join(OBJ, "digest_sources.txt")
            if os.path.exists(src_stamp) and open(src_stamp).read() != _source_digest():
                # same ABI number, different kernels: running them silently would be a stale build; there is no compiler to fix it here
                raise RuntimeError("libbrepgen_hip.so was built from other sources than brepgen_amd/csrc now holds (digest_sources.txt "
                                   "differs) and there is no hipcc on this box to rebuild it: build on a box with ROCm first")
            return LIB
        _hipcc()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    extra = ["-save-temps"] if save_temps else []
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(lambda s: _compile(s, extra), SOURCES))
    for _, warn in results:
        if verbose and warn.strip():
            sys.stderr.write(warn)
    objs = [o for o, _ in results]
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", LIB],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    with open(os.path.join(OBJ, "digest_sources.txt"), "w") as f:
        f.write(_source_digest())
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
    print(path)
