#!/bin/bash
# build (if stale) then run a command on the GPU box:  scripts/dev/g.sh <timeout-s> '<command>'
set -e
cd "$(dirname "$0")/../.."
python -m pytorchltr_amd.build >/dev/null
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout $t -- "mkdir -p gpurun_out/r03; $*"
