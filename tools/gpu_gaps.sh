#!/bin/bash
# inter-kernel gaps inside a step: rocprofv3 kernel trace of the bench, busy time vs span over the traced steps
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o g -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prof > /dev/null 2>&1
python - <<'PY'
import csv, glob
f=glob.glob('/tmp/gp/**/*kernel_trace.csv', recursive=True)[0]
rows=[(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# the timed region = the last 4 steps: find by locating adamw_kernel launches (one per step)
ends=[i for i,r in enumerate(rows) if 'adamw_kernel' in r[2]]
a,b=ends[-5]+1, ends[-1]+1   # 4 whole steps
seg=rows[a:b]
busy=sum(e-s for s,e,_ in seg); span=seg[-1][1]-seg[0][0]
gaps=[seg[i+1][0]-seg[i][1] for i in range(len(seg)-1)]
pos=[g for g in gaps if g>0]
print('kernels/step %.0f  busy %.2f ms/step  span %.2f ms/step  gap total %.2f ms/step  mean gap %.2f us  overlapped launches %d'%(len(seg)/4, busy/4e6, span/4e6, sum(pos)/4e6, sum(pos)/max(len(pos),1)/1e3, len(gaps)-len(pos)))
import collections
big=sorted(((g,seg[i][2][:40],seg[i+1][2][:40]) for i,g in enumerate(gaps)), reverse=True)[:5]
for g,x,y in big: print('  gap %.1f us after %s before %s'%(g/1e3,x,y))
PY
