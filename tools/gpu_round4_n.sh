#!/bin/bash
# is chol_potrf_kernel<double> (512 registers: needs an EMPTY SIMD) starved by the other stream's chol_offdiag?  Its duration
# with two half-batch streams vs alone on one stream, fp64 and fp32.
mkdir -p gpurun_out/r4n
for dt in f64 f32; do
  timeout 300 bash tools/kernel_stats.sh gpurun_out/r4n/${dt}_two_streams.txt -- python /root/repo/tools/bench_chol.py 1536 4096 $dt > /dev/null 2>>gpurun_out/r4n/err.txt
  THX_CHOL_SPLIT_MIN=0 timeout 300 bash tools/kernel_stats.sh gpurun_out/r4n/${dt}_one_stream.txt -- python /root/repo/tools/bench_chol.py 1536 4096 $dt > /dev/null 2>>gpurun_out/r4n/err.txt
done
head -8 gpurun_out/r4n/*.txt
