#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun):
#   scripts/collect_profiles.sh r02 [workloads...]
# Writes gpurun_out/<round>/prof/{trace_c2, <w>_fetch, <w>_write, <w>_sq}/summary.{txt,json} and
# profiles-ready pmc_<w>.json (copied by hand into profiles/ afterwards).
set -u
round="${1:-r02}"; shift || true
wls="${*:-c2 c3 c4 c5}"
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$root"
out="gpurun_out/$round/prof"
mkdir -p "$out"
# kernel trace + stats of the DEFAULT bench command (the line the driver records)
scripts/pmc_pass.sh "$out/trace_c2" none -- python "$root/bench.py" --steps 20 --warmup 5 --no-cpu-baseline
for w in $wls; do
  cmd="python $root/bench.py --workload $w --no-extra --no-cpu-baseline --steps 50 --warmup 10"
  scripts/pmc_pass.sh "$out/${w}_fetch" "FETCH_SIZE" -- $cmd
  scripts/pmc_pass.sh "$out/${w}_write" "WRITE_SIZE" -- $cmd
  scripts/pmc_pass.sh "$out/${w}_sq" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" -- $cmd
  python scripts/pmc_to_json.py "$w" "$out/${w}_fetch/summary.json" "$out/${w}_write/summary.json" "$out/${w}_sq/summary.json" > "$out/pmc_$w.log" 2>&1
  cp "profiles/pmc_$w.json" "$out/pmc_$w.json" 2>/dev/null
done
ls -la "$out"
