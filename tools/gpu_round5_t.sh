#!/bin/bash
# Round 5, call t: (1) socket power / shader clock of the factorisation (column pairs on / off) and of the K-loop's pieces;
# (2) kernel trace + HBM counters of the headline bench with column pairs (tools/gpu_profile.sh).
set -u
TAG=${1:-r5t}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
[ -x theseus_amd/lib/variants/power_pieces ] || hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o theseus_amd/lib/variants/power_pieces tools/microbench/power_pieces.hip
timeout 500 python tools/power_model.py 4 2>&1 | grep -v amdgpu.ids > $OUT/power_model.txt; cat $OUT/power_model.txt
THX_CHOL_COLPAIR=0 timeout 200 python tools/power_model.py 4 f32 2>&1 | grep -v amdgpu.ids | sed 's/two streams/two streams, column pairs OFF/' > $OUT/power_model_pairs_off.txt; cat $OUT/power_model_pairs_off.txt
timeout 600 bash tools/gpu_profile.sh ${TAG}
