#!/bin/bash
# O3 build as base; take single kernels from the -O1 build (model_bb58...) to find the broken one
ALT=triflow_amd/_cache/model_bb58bcdc369ee630a932.hsaco
for k in 7 8 9 10 11 12 13 14 15 16 17 18; do
  echo "== alt kernel $k"; TF_ALT_HSACO=$ALT TF_ALT_MASK=$((1<<k)) timeout -k 10 60 python tools/gpu_debug2.py 2>&1 | head -1
done
echo "== alt all bt"; TF_ALT_HSACO=$ALT TF_ALT_MASK=$(( (1<<12)|(1<<13)|(1<<14)|(1<<15)|(1<<16)|(1<<17)|(1<<18) )) timeout -k 10 60 python tools/gpu_debug2.py 2>&1 | head -1
echo "== alt all l1"; TF_ALT_HSACO=$ALT TF_ALT_MASK=$(( (1<<7)|(1<<8)|(1<<9)|(1<<10)|(1<<11) )) timeout -k 10 60 python tools/gpu_debug2.py 2>&1 | head -1
