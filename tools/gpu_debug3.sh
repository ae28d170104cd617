#!/bin/bash
while read -r v; do echo "== $v"; TRIFLOW_HIPCC_OPT="$v" timeout -k 10 120 python tools/gpu_debug2.py 2>&1 | head -4; done <<'EOT'
-O2
-O3 -mllvm -amdgpu-spill-vgpr-to-agpr=0
-O3 -fno-slp-vectorize
-O3 -fno-vectorize -fno-slp-vectorize
-O3 -mllvm -amdgpu-use-divergent-register-indexing
EOT
