#!/bin/bash
# Round 5, call r: socket power and shader clock of the factorisation and of the K-loop's pieces (tools/power_model.py)
set -u
TAG=${1:-r5r}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
[ -x theseus_amd/lib/variants/power_pieces ] || hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o theseus_amd/lib/variants/power_pieces tools/microbench/power_pieces.hip
timeout 400 python tools/power_model.py 3 2>&1 | grep -v amdgpu.ids > $OUT/power_model.txt; cat $OUT/power_model.txt
THX_CHOL_SPLIT_MIN=0 timeout 200 python tools/power_model.py 3 f32 2>&1 | grep -v amdgpu.ids | sed 's/two streams/ONE stream/' > $OUT/power_model_one_stream.txt; cat $OUT/power_model_one_stream.txt
