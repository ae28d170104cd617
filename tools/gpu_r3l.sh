#!/bin/bash
TAG=${1:-r3l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 300 python3 tools/gpu_stamps.py > $OUT/stamps.txt 2>&1
cat $OUT/stamps.txt
