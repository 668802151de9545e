#!/bin/bash
# round 6: the failed-reply fast path (fast_aer_reply_failed) against the same sources without it, one box:
# literal config 3 (a_*: N = 5 builds), literal config 5 (c5_*: N = 7 builds), the closed loop in both forms.
#   tools/build_variants.sh a_base:"-DRGB_X_FAST_FAILED=0" a_ff:""; ONLY_N=7 tools/build_variants.sh c5_base:"-DRGB_X_FAST_FAILED=0" c5_ff:""
#   gpurun -- 'bash tools/ab_fast_failed.sh TAG'
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; TAG=${1:-r06ff}
bash tools/ab_literal.sh ${TAG}_cfg3 a_ 3
bash tools/ab_literal.sh ${TAG}_cfg5 c5_ 5
bash tools/ab_variants.sh ${TAG}_loop 2 a_
