#!/bin/bash
# Round 4: the K-loop with hand-pipelined fragment reads (variants/kpipe.so, -DTHX_KLOOP_PIPE) against the current build, same box.
set -u
TAG=${1:-r4j}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ulimit -c 0
V=$(pwd)/theseus_amd/lib/variants/kpipe.so
THESEUS_HIP_LIB=$V timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_block_hessian.py -m gpu -q --maxfail=10 -p no:cacheprovider > $OUT/pytest_variant.txt 2>&1; tail -3 $OUT/pytest_variant.txt
for round in 1 2; do
  for lib in "" "$V"; do
    for cfg in "1536 4096 f32 3" "1536 1024 f32 3" "3072 256 f32 3"; do
      echo "== round $round lib=${lib:-current} n B dtype reps = $cfg" >> $OUT/ab_kpipe.txt
      THESEUS_HIP_LIB=$lib timeout 300 python tools/bench_chol.py $cfg 2>&1 | grep -v amdgpu | grep -E "^n=" >> $OUT/ab_kpipe.txt
    done
    echo -n "== round $round lib=${lib:-current} bench.py: " >> $OUT/ab_kpipe.txt
    THESEUS_HIP_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --parity-sample 8 --no-sparse-leg --legs none 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('value %.0f ms_per_step %.3f factor %.3f frac %.4f parity %.2e' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['parity']['hip_max_rel_pose_err']))" >> $OUT/ab_kpipe.txt
  done
done
cat $OUT/ab_kpipe.txt
