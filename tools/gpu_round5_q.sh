#!/bin/bash
# Round 5, call q: socket power / shader clock while thx_chol_factor runs back to back (tools/power_sample.py)
set -u
TAG=${1:-r5q}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
timeout 60 rocm-smi --showpower --showclocks --showmaxpower > $OUT/rocm_smi_idle.txt 2>&1
for dt in f32 f64; do
  timeout 200 python tools/power_sample.py $dt 6 2>&1 | grep -v amdgpu.ids > $OUT/power_$dt.txt; cat $OUT/power_$dt.txt
done
THX_CHOL_SPLIT_MIN=0 timeout 200 python tools/power_sample.py f32 6 2>&1 | grep -v amdgpu.ids > $OUT/power_f32_one_stream.txt; cat $OUT/power_f32_one_stream.txt
(timeout 100 python tools/power_sample.py f32 8 > $OUT/power_f32_smi_side.txt 2>&1 &) ; sleep 9; for k in 1 2 3; do timeout 20 rocm-smi --showpower --showclocks 2>&1 | grep -E 'Power|sclk|mclk' ; sleep 1; done > $OUT/rocm_smi_busy.txt; sleep 6; cat $OUT/rocm_smi_busy.txt; head -30 $OUT/rocm_smi_idle.txt
