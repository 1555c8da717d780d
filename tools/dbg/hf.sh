#!/bin/bash
# build the K3hf development harness in the given variants and run them on the GPU box:  tools/dbg/hf.sh "" "-DHF_STATS" "-DHF_STOP=3"
cd /root/repo/tools/dbg || exit 1
names=()
i=0
for v in "$@"; do
  n="hf_dev_v$i"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math $v -I../../include -o $n hf_dev.hip 2>&1 | grep -E "error" -A3
  names+=("$n")
  i=$((i+1))
done
cmd="cd tools/dbg"
i=0
for v in "$@"; do
  cmd="$cmd; echo '== variant [$v]'; timeout 100 ./hf_dev_v$i $HF_ARGS"
  i=$((i+1))
done
cd /root/repo && /usr/local/graft/bin/gpurun --timeout 400 -- "$cmd" 2>&1 | grep -v "^\[gpurun\] sending\|^\[gpurun\] status"
