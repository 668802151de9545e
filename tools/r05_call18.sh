#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for v in fuse5; do
  echo "== $v"; RGB_LIB=$R/ra_amd/csrc/variants/$v.so T=2 timeout 200 python tools/parity_tick0.py 2>&1 | grep "^tick\|upper\|Error\|error" | head -4
done
echo "== product"; T=2 timeout 200 python tools/parity_tick0.py 2>&1 | grep "^tick\|upper\|Error\|error" | head -4
