#!/bin/bash
# One parameterised A/B runner for a GPU box (replaces the one-shot r02_probe*/r02_final* scripts):
#   tools/gpu_ab.sh TAG [--pytest "ARGS"] [--bench "NAME:ARGS" ...] [--variants] [--prof]
# Every `--bench NAME:ARGS` runs `python bench.py ARGS` with the product library; with --variants it is repeated for
# every ra_amd/csrc/variants/*.so (tools/build_variants.sh; RGB_LIB=<variant>, N = 5 only).  Output: gpurun_out/TAG/.
# A variant whose name starts with "x_" is a timing probe that breaks parity: it runs with RGB_BENCH_NOCHECK=1.
set -u
TAG=${1:-ab}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
QUICK="--no-cpu-baseline --no-host-path --literal-ticks 0"
summ() {  # file label
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{sys.argv[2]:34s} {r.get('avg_tick_us', r['avg_launch_us']):7.2f} us/tick {d['value']/1e9:6.2f} G/s frac {r['frac']:.3f} "
          f"ms/step {d['ms_per_step']:.4f} {d['config'].get('launch','')[:5]} {d['config']['state_checksum']}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
    try: print(open(sys.argv[1].replace('.json', '.err')).read()[-1500:])
    except Exception: pass
PY
}
BENCHES=(); VARIANTS=0
while [ $# -gt 0 ]; do
  case "$1" in
    --pytest) shift; timeout ${PYTEST_TIMEOUT:-900} python -m pytest $1 -x -q 2>&1 | tail -15 | tee $OUT/pytest.txt ;;
    --bench) shift; BENCHES+=("$1") ;;
    --variants) VARIANTS=1 ;;
    --cmd) shift; bash -c "$1" 2>&1 | tail -40 | tee -a $OUT/cmd.txt ;;
  esac
  shift
done
for spec in "${BENCHES[@]}"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 600 python bench.py $QUICK $args > $OUT/$name.json 2> $OUT/$name.err; summ $OUT/$name.json "$name"
  if [ $VARIANTS = 1 ]; then
    for v in ra_amd/csrc/variants/*.so; do
      [ -f "$v" ] || continue
      vn=$(basename $v .so); nc=""; case $vn in x_*) nc=1;; esac
      ck=""; [ -n "$nc" ] && ck="--check-ticks 0"
      RGB_BENCH_NOCHECK=$nc RGB_LIB=$PWD/$v timeout 600 python bench.py $QUICK $args $ck --members 5 > $OUT/${name}__$vn.json 2> $OUT/${name}__$vn.err
      summ $OUT/${name}__$vn.json "$name/$vn"
    done
  fi
done
