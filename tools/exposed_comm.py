"""Exposed communication of one rank from a rocprofv3 --kernel-trace CSV: the time RCCL kernels run while NO compute kernel of the library
(ftmi::*) or of torch runs on the same GPU -- the part of the gradient / parameter exchange the step actually waits for.  Also prints the RCCL
total and how much of it sat under compute.  Usage: python tools/exposed_comm.py <*_kernel_trace.csv> [n_steps]
(step boundaries = optimiser kernels, as in tools/step_trace.py; without n_steps the whole trace is used)."""
import csv
import sys


def is_comm(name):
    n = name.lower()
    return "nccl" in n or "rccl" in n


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def subtract(a, b):
    """total length of the intervals a (disjoint, sorted) not covered by b (disjoint, sorted)"""
    tot, j = 0, 0
    for s, e in a:
        cur = s
        while j < len(b) and b[j][1] <= cur:
            j += 1
        k = j
        while k < len(b) and b[k][0] < e:
            if b[k][0] > cur:
                tot += b[k][0] - cur
            cur = max(cur, b[k][1])
            k += 1
        if cur < e:
            tot += e - cur
    return tot


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Agent_Id", "")))
    rows.sort()
    agents = sorted({r[3] for r in rows})
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for ag in agents:
        rr = [r for r in rows if r[3] == ag]
        if nsteps:
            marks = [i for i, r in enumerate(rr) if "adamw" in r[2]]
            if len(marks) > nsteps:
                rr = rr[marks[-nsteps - 1] + 1:marks[-1] + 1]
        comm = union([(s, e) for s, e, n, _ in rr if is_comm(n)])
        comp = union([(s, e) for s, e, n, _ in rr if not is_comm(n)])
        total = sum(e - s for s, e in comm)
        exposed = subtract(comm, comp)
        div = max(nsteps, 1)
        print(f"agent {ag}: RCCL kernel time {total / div / 1e6:.3f} ms{'/step' if nsteps else ''}, of which exposed (no compute kernel running) "
              f"{exposed / div / 1e6:.3f} ms, under compute {(total - exposed) / div / 1e6:.3f} ms; {sum(1 for r in rr if is_comm(r[2]))} RCCL launches")


if __name__ == "__main__":
    main()
