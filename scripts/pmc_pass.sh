#!/bin/bash
# usage: scripts/pmc_pass.sh <outdir> "<counters>" -- <command...>
# One rocprofv3 PMC pass (kernel trace + counters only), summarised to <outdir>/summary.txt
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$1"; case "$out" in /*) ;; *) out="$root/$out";; esac
ctrs="$2"; shift 3
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
work="/tmp/pmc_$$_$RANDOM"
rm -rf "$work"
if [ "$ctrs" = "none" ]; then rocprofv3 --kernel-trace --stats -d "$work" -o res -- "$@" > "$work.log" 2>&1; else rocprofv3 --kernel-trace --pmc $ctrs -d "$work" -o res -- "$@" > "$work.log" 2>&1; fi
db=$(find "$work" -name "*.db" | head -1)
if [ -n "$db" ]; then python "$root/scripts/rocpd_summary.py" "$db" "$out/summary.json" > "$out/summary.txt" 2>&1; else tail -20 "$work.log" > "$out/summary.txt"; fi
