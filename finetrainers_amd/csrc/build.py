"""Build libftmi355.so (gfx950) in-tree with hipcc.  `python -m finetrainers_amd.csrc.build`"""

from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gemm.hip", "attention.hip", "rowwise.hip", "cast.hip", "cogvideox.hip", "ltx_dit.hip", "cog_dit.hip", "hy_dit.hip", "wan.hip", "wan_dit.hip", "collective.hip", "api.hip"]
HEADERS = ["common.hip.h", "kernels.h", os.path.join("..", "..", "include", "ftmi355.h"), "attention_pl.hip.h"]
HEADERS += sorted(f for f in os.listdir(HERE) if f.startswith("attn_pl_") and f.endswith(".inc"))  # generated statement lists (tools/gen_attn_pl.py)
LIB = os.path.join(HERE, "..", "libftmi355.so")
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",  # the eager reference graph never fuses a*b+c; keep rounding points identical
    "-munsafe-fp-atomics",  # hardware fp32 atomic add for the split-M weight-gradient reduction
    "-Wno-unused-result",
]
# per-file extras.  attention: keep the MFMA accumulators in the (unified) VGPR file -- the online-softmax rescale and the
# dS products read/modify them with VALU every tile, and the accumulator-file round trip (v_accvgpr_read/write) was 25-35 %
# of the loop's instructions in a VALU-bound kernel.
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
# FTMI_EXPERIMENTAL=1: also compile the research K loops / timing experiments of tools/experimental/gemm_experimental.hip.h (tools/bench_gemm.py,
# tools/ab_variants.sh).  Never set for the product library.
if os.environ.get("FTMI_EXPERIMENTAL", "0") not in ("", "0"):
    FLAGS.append("-DFTMI_EXPERIMENTAL")
    _EXP = os.path.join("..", "..", "tools", "experimental")  # research sources live outside csrc/: the product directory holds only what libftmi355.so ships
    SOURCES.insert(1, os.path.join(_EXP, "gemm_skinny.hip"))  # the 64 x 128-tile LoRA down-projection kernel (measured slower than the shipped one: profiles/r03_skinny_experiments.txt)
    SOURCES.insert(1, os.path.join(_EXP, "gemm_sk.hip"))      # the persistent stream-K GEMM (5-25 % slower than the shipped kernels: profiles/r03_gemm_streamk.txt)
    HEADERS.append(os.path.join(_EXP, "gemm_experimental.hip.h"))
    HEADERS += [os.path.join(_EXP, f) for f in sorted(os.listdir(os.path.join(HERE, _EXP))) if f.startswith("attention_experimental_") or f.startswith("attn_pl_fwd_") or f.startswith("attn_pl_dq128_")]


# FTMI_TRACE=1: the same sources with the in-kernel phase stamps of the tiled NT GEMMs compiled in (gemm.hip NT_STAMP; tools/nt_trace.py) -- a SEPARATE
# library (libftmi355_trace.so, objects under build_trace/), never the product one.  Load it with FTMI_LIB=<path>.
_TRACE = os.environ.get("FTMI_TRACE", "0") not in ("", "0")
if _TRACE:
    FLAGS.append("-DFTMI_TRACE")
    LIB = os.path.join(HERE, "..", "libftmi355_trace.so")


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths, extra: str = "") -> str:
    """Content hash of the inputs of one build product.  File times are not used: a snapshot copied to another machine (gpurun) keeps
    contents, not necessarily times, and a shipped object newer than its edited source must not pass for fresh."""
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def _stale(target: str, digest: str) -> bool:
    stamp = target + ".sha256"
    return not (os.path.exists(target) and os.path.exists(stamp) and open(stamp).read() == digest)


def _stamp(target: str, digest: str) -> None:
    with open(target + ".sha256", "w") as f:
        f.write(digest)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build_trace" if _TRACE else "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]

    def compile_one(src: str) -> str:
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, os.path.basename(src) + ".o")
        flags = FLAGS + EXTRA_FLAGS.get(src, []) + ["-I", HERE]
        dg = _digest([s] + hdrs, " ".join(flags))  # a change of flags (e.g. FTMI_EXPERIMENTAL on / off) invalidates the object
        if force or _stale(o, dg):
            cmd = [hipcc] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            _stamp(o, dg)
        return o

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    lib = os.path.abspath(LIB)
    dg = _digest(objs)
    if force or _stale(lib, dg):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        _stamp(lib, dg)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
