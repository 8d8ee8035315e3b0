#!/bin/bash
# round 6, last session: the measurement set of the final tree in one gpurun call -- GPU suite + default bench line (tools/gpu_round6_set.sh), then the
# rocprofv3 kernel trace + HBM counters of the headline (fp32) and of the fp64 leg's configuration (tools/gpu_profile.sh)
TAG=${1:-r6aj}
bash tools/gpu_round6_set.sh $TAG
bash tools/gpu_profile.sh ${TAG}_f32 > gpurun_out/$TAG/prof_f32.log 2>&1; tail -30 gpurun_out/prof_${TAG}_f32/summary.txt
bash tools/gpu_profile.sh ${TAG}_f64 --dtype f64 --steps 3 --warmup 1 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none > gpurun_out/$TAG/prof_f64.log 2>&1; tail -30 gpurun_out/prof_${TAG}_f64/summary.txt
