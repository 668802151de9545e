#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
echo "== product lib"; timeout 200 python tools/parity_tick0.py 2>&1 | grep -v amdgpu.ids | tail -12
echo "== product lib, hint 1"; HINT=1 timeout 200 python tools/parity_tick0.py 2>&1 | grep -v amdgpu.ids | tail -12
echo "== head lib (round 4)"; RGB_LIB=$R/ra_amd/csrc/variants/head.so timeout 200 python tools/parity_tick0.py 2>&1 | grep -v amdgpu.ids | tail -6
