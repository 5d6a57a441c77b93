#!/bin/bash
# registers / scratch of the gfx950 kernels inside a fat binary or shared library:  tools/kernel_regs.sh <binary> [name filter]
set -e
bin=$1; pat=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input="$bin" > $tmp/list 2>/dev/null || true
t=$(grep gfx950 $tmp/list | head -1)
if [ -n "$t" ]; then /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$bin" --targets="$t" --output=$tmp/co.o; else
  # shared library: the code objects sit in .hip_fatbin
  /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section=.hip_fatbin=$tmp/fat.bin "$bin" /dev/null 2>/dev/null || objcopy -O binary --only-section=.hip_fatbin "$bin" $tmp/fat.bin
  t=$(/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$tmp/fat.bin | grep gfx950 | head -1)
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets="$t" --output=$tmp/co.o
fi
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/co.o | python3 -c "
import sys,re
txt=sys.stdin.read()
for m in re.finditer(r'\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', txt, re.S):
    ag,name,scr,sg,vg=m.groups()
    import subprocess
    print(f'vgpr {vg:>4} agpr {ag:>4} sgpr {sg:>4} scratch {scr:>6}  {name}')
" | while read line; do n=$(echo "$line" | awk '{print $NF}'); d=$(echo $n | c++filt); echo "${line% *} $d"; done | grep -E "$pat" || true
rm -rf $tmp
