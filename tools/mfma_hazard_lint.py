"""Static check of the hand-placed GEMM / attention kernels for ONE hazard hipcc cannot see: an accumulator register written by a v_mfma that sits inside an inline-asm
statement is invisible to the compiler's hazard recogniser, so a copy hipcc itself inserts behind the loop (live-range split of an accumulator tile: v_accvgpr_mov_b32 /
v_accvgpr_read_b32 at the exit edge) can read the register before the matrix pipe has written it -- gfx950 has no interlock there, the ISA asks for passes + 3 wait states
(7 behind a 16 x 16 x 32 bf16 MFMA, 11 behind a 32 x 32 x 16).  Found in round 6 as one wrong accumulator register (a220 <- a224 five instructions behind the last MFMA of a K loop).  The kernels keep the
distance by construction (acc_fence16, the settle() statement on the loop's exit path); this script proves it on the built code object:

    python tools/mfma_hazard_lint.py [finetrainers_amd/csrc/build/gemm.hip.o ...]      exit status 1 and a listing if any read comes too early

A "read" is any non-MFMA instruction with an AGPR source; distance is counted in issued instructions (s_nop N = N + 1), linearly through the listing (branches are not
followed: a loop's back edge only shortens real distances where the body is shorter than the threshold, which no body here is)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
# wait states between an MFMA and a non-MFMA read of its result: passes + 3 (LLVM's GCNHazardRecogniser for gfx940+).  gfx950: 16 x 16 x 32 bf16 = 4 passes (16 cycles),
# 32 x 32 x 16 = 8 -- the failure that prompted this script fits: a copy 6 wait states behind a 16 x 16 x 32 MFMA read the old value, the next one at 7 the new.  One spare.
NEED = {"v_mfma_f32_16x16x32": 8, "v_mfma_f32_32x32x16": 12, "v_mfma_f32_32x32x8": 12, "v_mfma_f32_16x16x16": 8}
# kernels whose MFMAs are written as inline asm (everywhere else hipcc sees the builtin and inserts the wait states itself)
ASM_KERNELS = ("gemm_nt16_kernel", "gemm_nt16_fused_kernel", "_pl_kernel", "gemm_nt_kernelILi256ELi256ELi64ELi2ELi2E", "gemm_nt_kernelILi256ELi256ELi64ELi2ELi4E")
AREG = re.compile(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b")


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section=.hip_fatbin=" + fat, obj, "/dev/null"], check=True, stderr=subprocess.DEVNULL)
    targets = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", "--input=" + fat], check=True, capture_output=True, text=True).stdout.split()
    t = [x for x in targets if "gfx950" in x][0]
    co = os.path.join(tmp, "co.o")
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=" + t, "--output=" + co], check=True)
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout


def aregs(text):
    out = []
    for m in AREG.finditer(text):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def lint(listing, name_filter=None):
    problems = []
    kernel, pos, last = None, 0, {}
    for line in listing.splitlines():
        if line.endswith(">:") and "<" in line:
            kernel, pos, last = line.split("<", 1)[1][:-2], 0, {}
            continue
        if kernel is None or not line.startswith("\t"):
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        op = ins.split()[0]
        if op == "s_nop":
            pos += int(ins.split()[1]) + 1
            continue
        pos += 1
        if op.startswith("v_mfma"):
            need = next((v for k, v in NEED.items() if op.startswith(k)), 20)
            dst = ins[len(op):].split(",")[0]
            for r in aregs(dst):
                last[r] = (pos, need, ins)
            continue
        if " a" not in ins and ",a" not in ins:
            continue
        operands = ins[len(op):]
        srcs = operands.split(",", 1)[1] if (op.startswith("v_accvgpr") or op.startswith("v_")) and "," in operands else operands
        for r in aregs(srcs):
            if r in last and pos - last[r][0] < last[r][1]:
                if name_filter is None or name_filter in kernel:
                    problems.append((kernel, ins, pos - last[r][0], last[r][1], last[r][2]))
    return problems


def main():
    objs = sys.argv[1:] or [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "finetrainers_amd", "csrc", "build", f) for f in ("gemm.hip.o", "attention.hip.o")]
    bad = 0
    for o in objs:
        pr = [x for x in lint(disassemble(o)) if any(k in x[0] for k in ASM_KERNELS)]
        print(f"{o}: {len(pr)} early accumulator reads")
        for k, ins, d, need, mf in pr[:40]:
            print(f"  {subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()[:110]}\n      {ins}   <- {d} wait states after   {mf}   (needs {need})")
        bad += len(pr)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
