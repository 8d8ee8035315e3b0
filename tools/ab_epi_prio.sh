set -u
OUT=gpurun_out/r5l; mkdir -p $OUT
THESEUS_HIP_LIB=$PWD/theseus_amd/lib/variants/prof64prio.so timeout 600 python tools/prof/off_prof64.py 4096 2>&1 | grep -A9 "block-compact" > $OUT/off_prof64_prio.txt
cat $OUT/off_prof64_prio.txt
ARGS="--steps 10 --warmup 3 --cpu-sample 0 --parity-sample 0 --no-sparse-leg --legs none"
for round in 1 2; do
  for lib in "" theseus_amd/lib/variants/epiprio.so; do
    for dt in f32 f64; do
      echo -n "round $round lib=${lib:-current} $dt : " >> $OUT/ab_prio.txt
      THESEUS_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py $ARGS --dtype $dt 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('value %.0f ms_per_step %.3f factor %.3f frac %.4f' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))" >> $OUT/ab_prio.txt
    done
  done
done
cat $OUT/ab_prio.txt
