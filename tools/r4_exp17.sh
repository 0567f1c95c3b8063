#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1
for a in 0 1 2 4 3 7; do echo "== att_abl $a (1 no exp, 2 re-fetch two tiles, 4 no PV)"; RP_OPTIONS="att_abl=$a" python tools/attn_bench.py 2>&1 | grep -v amdgpu | cut -c1-80; done
