#!/bin/bash
# Register / scratch / LDS figures of selected kernel instances in seconds (device-only compile of a stub that instantiates them):
#   tools/dbg/res.sh 'k_scan_mfma<4, 8, 1>(const MfmaParams)' 'k_a1_verify<16, 8>(const MfmaParams)' ...
# -S: also keep the ISA in /tmp/res/res.s
set -e
CS=$(cd "$(dirname "$0")/../../multimedia-indexing_amd/csrc" && pwd)
mkdir -p /tmp/res
{
  echo '#include <hip/hip_runtime.h>'
  for h in mmidx_kernels.h mmidx_scan_grp.h mmidx_scan_mfma.h mmidx_scan_mfma_kc.h mmidx_scan_mfma_a.h mmidx_frontend.h; do echo "#include \"$h\""; done
  for k in "$@"; do [ "$k" = "-S" ] || echo "template __global__ void $k;"; done
} > /tmp/res/res.hip
EXTRA=""
for k in "$@"; do [ "$k" = "-S" ] && EXTRA="-save-temps=obj"; done
cd /tmp/res
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I"$CS/../../include" -I"$CS" --cuda-device-only -c res.hip -o res.o $EXTRA \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|error" | sed -E 's/^.*remark: +//; s/ \[-Rpass.*//; s/^Function //' | paste -s -d' ' | sed 's/Name: /\n/g' | grep -E "ILi|error" | tr -s ' ' || true
[ -n "$EXTRA" ] && ls /tmp/res/*.s 2>/dev/null
