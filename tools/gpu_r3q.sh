#!/bin/bash
# Round 3: level-1 chunk length where the chunks fill the GPU anyway (config 5, 8 members per GPU)
TAG=${1:-r3q}
bash tools/gpu_ab.sh ${TAG}_cfg5 "--steps 20 --config 5" "" "|TRIFLOW_M1=40" "|TRIFLOW_M1=48" "|TRIFLOW_M1=64"
bash tools/gpu_ab.sh ${TAG}_m8 "--steps 20 --members-per-gpu 8" "" "|TRIFLOW_M1=40" "|TRIFLOW_M1=48" "|TRIFLOW_M1=64"
