#!/bin/bash
# copy the summaries of gpurun_out/r05 (tools/profile_round5.sh) into profiles/ under r05_ names
set -u
cd "$(dirname "$0")/.."; S=gpurun_out/r05; D=profiles
cp $S/stats/r05_kernel_stats.csv $D/r05_kernel_stats_driver_form.csv
cp $S/stats_lit/r05_kernel_stats.csv $D/r05_literal_configs_kernel_stats.csv
cp $S/pmc_summary.txt $D/r05_pmc_summary.txt
cp $S/traffic.json $D/r05_traffic.json
for n in bench bench_driver_form bench_driver_form_device_plan bench_driver_form_persistent bench_driver_form_tick bench_device_plan; do
  [ -s $S/$n.json ] && tail -1 $S/$n.json > $D/r05_$n.json
done
[ -s $S/train_timeline.txt ] && cp $S/train_timeline.txt $D/r05_train_timeline.txt
[ -s $S/train_decline_hist.txt ] && cp $S/train_decline_hist.txt $D/r05_train_decline_hist.txt
{ echo "commit $(cat $S/commit.txt)"; cat $S/timing.txt; } > $D/r05_evidence_run.txt
ls -la $D/r05_*
