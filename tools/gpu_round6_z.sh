#!/bin/bash
# round 6: the matrix-core scatter of H's pieces on the OTHER block-compact consumers: bundle adjustment (dense tiles: the default
# stays the LDS gather; forced through the scatter for the A/B) and the level-scheduled 4096-pose graphs (few pieces: scatter by default)
O=gpurun_out/${1:-r6z4}; mkdir -p $O
for v in default scatter; do
  if [ $v = scatter ]; then export THX_HB_SCATTER_MAX_PIECES=1000000; else unset THX_HB_SCATTER_MAX_PIECES; fi
  timeout 600 python tools/bench_ba.py 512 8192 256 f32 5 2>&1 | tail -12 > $O/ba_$v.txt
  echo "== BA $v"; cat $O/ba_$v.txt
done
for v in scatter lds; do
  if [ $v = lds ]; then export THX_HB_SCATTER_MAX_PIECES=0; else unset THX_HB_SCATTER_MAX_PIECES; fi
  for b in 64 256; do
    timeout 600 python tools/bench_sparse.py 4096 $b f32 10 2>&1 | tail -4 > $O/sparse_${b}_$v.txt
    echo "== sparse 4096 b$b $v"; cat $O/sparse_${b}_$v.txt
  done
done
