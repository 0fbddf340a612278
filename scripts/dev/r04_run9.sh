#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gputests2.log 2>&1
tail -8 $OUT/gputests2.log
