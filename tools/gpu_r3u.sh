#!/bin/bash
# Round 3: depth of the back-substitution's request ring now that a node also carries the operands of the state update
TAG=${1:-r3u}
bash tools/gpu_ab.sh ${TAG}_cfg5 "--steps 20 --config 5" "" "-DTF_BACKSUB_DEPTH=2" "-DTF_BACKSUB_DEPTH=4"
bash tools/gpu_ab.sh ${TAG}_cfg3 "--steps 20" "" "-DTF_BACKSUB_DEPTH=2" "-DTF_BACKSUB_DEPTH=4"
