#!/bin/bash
# Round 6: is the step power-bound?  Samples socket power, sclk / mclk while bench.py replays the captured step.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r6power}; mkdir -p $O
( python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline $BENCH_ARGS > $O/bench.log 2>&1 ) &
BP=$!
sleep 45
for i in $(seq 1 24); do
  rocm-smi --showpower --showclocks --showuse --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|busy|Temperature \(Sensor (junction|edge)" | tr '\n' ';' | sed 's/  */ /g' | cut -c1-600
  echo
  sleep 0.5
done > $O/smi_samples.txt
wait $BP
grep '^{' $O/bench.log | tail -1 | cut -c1-200
head -30 $O/smi_samples.txt
rocm-smi --showmaxpower 2>/dev/null | grep -i power | head -3
amd-smi static --limit 2>/dev/null | head -30
