#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "== two workgroups per CU (default)"; PROBE_CONVS=1 python tools/overlap_probe.py 2>&1 | grep -v amdgpu
echo "== one workgroup per CU"; FB_H4_WG_PER_CU=1 PROBE_CONVS=1 python tools/overlap_probe.py 2>&1 | grep -v amdgpu
