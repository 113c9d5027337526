#!/bin/bash
for v in w0 w4; do cp srba_amd/lib/variants/libsrba_hip_$v.so srba_amd/lib/libsrba_hip.so; echo "=== $v"; python tools/diag_occupancy.py "$@" 2>&1 | tail -11; done
