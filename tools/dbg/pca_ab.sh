#!/bin/bash
# PCA kernel variants on one box: tools/dbg/pca_ab.sh <lib|default> ...
for rep in 1 2; do
for L in "$@"; do
  if [ "$L" = default ]; then unset MMIDX_LIB; else export MMIDX_LIB=$PWD/multimedia-indexing_amd/csrc/ab/$L; fi
  python tools/dbg/pca_time.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    n, j = l.split(' ', 1)
    d = json.loads(j)['pca_8192_to_128']
    print(n.split('/')[-1], d['ms'], d['tflops_f64'], d['roofline']['frac'], d['max_abs_err_vs_torch'])
"
done
done
