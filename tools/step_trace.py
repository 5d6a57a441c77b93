"""Per-step kernel table from a rocprofv3 --kernel-trace CSV of `bench.py`: the trace is cut at the optimiser kernel (one adamw launch per
step), set-up and warm-up are dropped, and the kernels of the last N steps are summed -- milliseconds per step and launches per step per
kernel, GPU-busy time and idle gaps.  Usage: python tools/step_trace.py <*_kernel_trace.csv> [n_steps] [out.csv] [marker]
(marker: substring of a kernel launched exactly once per step -- default "adamw_kernel"; "mse_loss_kernel" for the workloads whose optimiser is several launches)"""
import collections
import csv
import re
import sys

path = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marker = sys.argv[4] if len(sys.argv) > 4 else "adamw_kernel"
marks = [i for i, r in enumerate(rows) if marker in r[2]]
if len(marks) < nsteps + 1:
    sys.exit(f"only {len(marks)} '{marker}' launches in the trace")
lo, hi = marks[-nsteps - 1] + 1, marks[-1] + 1
sel = rows[lo:hi]
span = (sel[-1][1] - rows[lo - 1][1]) / nsteps / 1e6


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n).replace("void ", "").replace("ftmi::", "")
    return n[:100]


agg = collections.OrderedDict()
busy = 0
prev_end = rows[lo - 1][1]
gaps = 0
for s, e, n in sel:
    k = short(n)
    a = agg.setdefault(k, [0, 0])
    a[0] += e - s
    a[1] += 1
    busy += e - s
    if s > prev_end:
        gaps += s - prev_end
    prev_end = max(prev_end, e)
items = sorted(agg.items(), key=lambda kv: -kv[1][0])
print(f"steps {nsteps}; wall per step {span:.3f} ms; kernel time per step {busy / nsteps / 1e6:.3f} ms; idle between kernels {gaps / nsteps / 1e6:.3f} ms; launches per step {len(sel) / nsteps:.0f}")
out = [("kernel", "ms_per_step", "launches_per_step", "avg_us")]
for k, (t, c) in items:
    out.append((k, f"{t / nsteps / 1e6:.4f}", f"{c / nsteps:.1f}", f"{t / c / 1e3:.1f}"))
for r in out[:45]:
    print("%-100s %12s %10s %10s" % r)
if len(sys.argv) > 3:
    with open(sys.argv[3], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow([f"# {nsteps} steady-state steps of bench.py; wall {span:.3f} ms/step, kernels {busy / nsteps / 1e6:.3f} ms/step, gaps {gaps / nsteps / 1e6:.3f} ms/step"])
        w.writerows(out)
