#!/bin/bash
set -u
OUT=gpurun_out/k; mkdir -p $OUT
export THX_REFERENCE_ROOT=$(pwd)/_refcopy THX_PLUGIN_DEVICE=cuda
python tools/dropin_bench.py --steps 10 --profile > $OUT/dropin_profile.json 2> $OUT/dropin_profile.txt
tail -1 $OUT/dropin_profile.json
