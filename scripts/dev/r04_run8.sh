#!/bin/bash
OUT=gpurun_out/r04
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_mailbox.py -x -q -rs > $OUT/mailbox.log 2>&1
tail -30 $OUT/mailbox.log
