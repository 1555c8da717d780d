#!/bin/bash
# build the library quietly; print errors and fail loudly
cd "$(dirname "$0")/../../multimedia-indexing_amd/csrc" || exit 1
if ! make -s > /tmp/mk.log 2>&1; then grep -E "error" /tmp/mk.log | head -20; echo "BUILD FAILED"; exit 1; fi
ls -la --time-style=full-iso libmmidx_hip.so | cut -c20-
