#!/bin/bash
# The whole GPU suite N times, each run a fresh python process on ONE lease, WITHOUT -x (every failure of every run is
# recorded).  Usage on the GPU box:  bash tools/flake_hunt.sh [N=5] [extra pytest args]
# Logs: gpurun_out/flake_hunt/run_<i>.log + summary.txt (copied to profiles/round6_flake_hunt/ when it is the record).
N=${1:-5}
shift
OUT=gpurun_out/flake_hunt
mkdir -p "$OUT"
: > "$OUT/summary.txt"
git rev-parse HEAD 2>/dev/null >> "$OUT/summary.txt"
for i in $(seq 1 "$N"); do
  t0=$(date +%s)
  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=12 "$@" > "$OUT/run_$i.log" 2>&1
  rc=$?
  echo "run $i rc=$rc $(( $(date +%s) - t0 ))s : $(grep -E '^(FAILED|ERROR)|passed|failed' "$OUT/run_$i.log" | tr '\n' ' ')" >> "$OUT/summary.txt"
done
cat "$OUT/summary.txt"
