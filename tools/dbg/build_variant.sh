#!/bin/bash
# tools/dbg/build_variant.sh <name> [-D...]: builds tools/dbg/lib_<name>.so with extra defines (kernel A/B experiments)
name=$1; shift
cd /root/repo/multimedia-indexing_amd/csrc && make -s mmidx_learn.o mmidx_probe.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -I../../include "$@" \
  -c -o /tmp/variant_$name.o mmidx_api.hip 2>&1 | grep -E "error" -A3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /root/repo/tools/dbg/lib_$name.so /tmp/variant_$name.o /root/repo/multimedia-indexing_amd/csrc/mmidx_learn.o /root/repo/multimedia-indexing_amd/csrc/mmidx_probe.o -ldl
