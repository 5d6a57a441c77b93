"""Static check of the built gfx950 code objects for the SECOND hazard class that round 6 met in kernels whose loads are inline-asm statements:
a register that is the destination of a load STILL IN FLIGHT is written or read by another instruction.

hipcc tracks the completion of its own loads (it inserts the s_waitcnt itself); a load inside an asm statement completes when the kernel's own counted wait says so,
and the compiler knows nothing about that: if the asm statement's output is dead (a fragment read "for the next stage" behind the last stage) or is copied right
behind the statement (a live-range split under register pressure), the register is handed to the next value while the LDS / memory read has not returned -- the
returning data then overwrites the new value, or the copy reads the old one.  Seen as a memory fault (the reused register was a load's address).  This script replays
every kernel's instruction stream with the two in-order counters (lgkmcnt: LDS reads + scalar loads; vmcnt: vector memory) and flags, inside a basic block,

    * a write to a VGPR / AGPR that is the destination of an outstanding load,
    * a read of such a register by anything but the load itself.

    python tools/inflight_reg_lint.py [objects]          exit status 1 and a listing if anything is flagged

Linear scan; the outstanding set is dropped at every branch and at s_endpgm (a hazard that needs two blocks to happen is out of its reach, a false alarm from
a path that is never taken is not produced).  vmcnt is modelled in issue order, stores included (gfx9 counts them there)."""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mfma_hazard_lint import disassemble  # noqa: E402

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
LDS_READ = ("ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append")
VM_LOAD = ("buffer_load", "global_load", "flat_load", "scratch_load", "buffer_atomic", "global_atomic", "flat_atomic")
VM_STORE = ("buffer_store", "global_store", "flat_store", "scratch_store")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def split_operands(ins):
    op, _, rest = ins.partition(" ")
    ops = [o.strip() for o in rest.split(",")] if rest.strip() else []
    return op, ops


def lint(listing, kernels=None):
    problems = []
    kernel, lgkm, vm = None, [], []  # queues of (set of dst regs, instruction), oldest first

    def outstanding():
        s = {}
        for q in (lgkm, vm):
            for dst, ins in q:
                for r in dst:
                    s[r] = ins
        return s

    for line in listing.splitlines():
        if line.endswith(">:") and "<" in line:
            kernel, lgkm, vm = line.split("<", 1)[1][:-2], [], []
            continue
        if kernel is None or not line.startswith("\t"):
            continue
        if kernels is not None and not any(k in kernel for k in kernels):
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        op, ops = split_operands(ins)
        if op.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_swappc")):
            lgkm, vm = [], []
            continue
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m:
                vm = vm[max(0, len(vm) - int(m.group(1))):]
            m = re.search(r"lgkmcnt\((\d+)\)", ins)
            if m:
                n = int(m.group(1))
                # scalar loads may return out of order: only lgkmcnt(0) retires a queue that holds any
                if n == 0 or not any(i.startswith("s_load") or i.startswith("s_buffer_load") for _, i in lgkm):
                    lgkm = lgkm[max(0, len(lgkm) - n):]
            if not re.search(r"vmcnt|lgkmcnt|expcnt", ins):  # a raw immediate: treat as a full wait
                lgkm, vm = [], []
            continue
        out = outstanding()
        is_lds_read = op.startswith(LDS_READ)
        is_vm_load = op.startswith(VM_LOAD)
        is_vm_store = op.startswith(VM_STORE)
        lds_dma = is_vm_load and (" lds" in ins or "_lds_" in op)  # direct-to-LDS forms: no register destination, the first operand is the address
        # destination operands (registers the instruction writes) and source operands
        if is_lds_read or (is_vm_load and not lds_dma and not op.startswith(("buffer_atomic", "global_atomic", "flat_atomic"))):
            dst, src = regs(ops[0]) if ops else set(), set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
        elif is_vm_store or lds_dma or op.startswith(("ds_write", "ds_store", "s_")) or op in ("s_nop", "s_barrier"):
            dst, src = set(), set().union(*[regs(o) for o in ops]) if ops else set()
        elif op.startswith("v_mfma") or op.startswith("v_smfmac"):
            dst, src = regs(ops[0]), set().union(*[regs(o) for o in ops[1:]])
        elif op.startswith("v_cmp") and not op.startswith("v_cmpx"):
            dst, src = set(), set().union(*[regs(o) for o in ops]) if ops else set()
        elif op.startswith(("v_", "ds_")):
            dst, src = regs(ops[0]) if ops else set(), set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
        else:
            dst, src = set(), set()
        for r in sorted(dst | src):
            if r in out:
                problems.append((kernel, ins, "writes" if r in dst else "reads", f"{r[0]}{r[1]}", out[r]))
                break
        if is_lds_read:
            lgkm.append((dst, ins))
        elif op.startswith(("s_load", "s_buffer_load")):
            lgkm.append((set(), ins))
        elif op.startswith(("ds_write", "ds_store")):
            lgkm.append((set(), ins))
        elif is_vm_load:
            vm.append((set() if lds_dma else dst, ins))
        elif is_vm_store:
            vm.append((set(), ins))
    return problems


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    objs = sys.argv[1:] or [os.path.join(root, "finetrainers_amd", "csrc", "build", f) for f in ("gemm.hip.o", "attention.hip.o")]
    bad = 0
    for o in objs:
        pr = lint(disassemble(o))
        print(f"{o}: {len(pr)} uses of a register whose load is still in flight")
        seen = {}
        for k, ins, what, r, load in pr:
            seen.setdefault(k, []).append((ins, what, r, load))
        for k, items in list(seen.items())[:30]:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:120]
            ins, what, r, load = items[0]
            print(f"  {name}: {len(items)}x, first: `{ins}` {what} {r} while `{load}` is outstanding")
        bad += len(pr)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
