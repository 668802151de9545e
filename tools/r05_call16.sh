#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
RGB_LIB=$R/ra_amd/csrc/variants/cur.so timeout 200 python tools/plan_kernel_time.py 2>&1 | tail -5
