#!/bin/bash
# The round's measurement set in ONE gpurun call (outputs under gpurun_out/<tag>/; copy what is to be judged into profiles/):
# GPU test suite, bench.py (fp32 headline with cpu_baseline / parity / tile-sparse leg; fp64; implicit; adaptive), bundle
# adjustment, large sparse pose graph, rocprofv3 kernel trace + HBM counters of the bench, MFMA-busy counters of the Cholesky.
set -u
TAG=${1:-r2h}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider -s > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
python bench.py > $OUT/bench_f32.json 2> $OUT/bench_f32.err; head -c 400 $OUT/bench_f32.json; echo
python bench.py --dtype f64 --steps 5 --cpu-sample 32 > $OUT/bench_f64.json 2> $OUT/bench_f64.err; head -c 300 $OUT/bench_f64.json; echo
python bench.py --implicit --batch 1024 > $OUT/bench_implicit_b1024.json 2> $OUT/bench_implicit.err; head -c 300 $OUT/bench_implicit_b1024.json; echo
python bench.py --adaptive --cpu-sample 0 --parity-sample 0 --no-sparse-leg > $OUT/bench_f32_adaptive.json 2> $OUT/bench_adaptive.err; head -c 300 $OUT/bench_f32_adaptive.json; echo
python tools/bench_ba.py 512 8192 256 f32 5 > $OUT/ba_bench_f32.log 2>&1; tail -4 $OUT/ba_bench_f32.log
python tools/bench_sparse.py 1024 64 f32 5 > $OUT/sparse_bench.txt 2>&1; python tools/bench_sparse.py 2048 64 f32 5 >> $OUT/sparse_bench.txt 2>&1; grep -v amdgpu $OUT/sparse_bench.txt | tail -6
tools/gpu_profile.sh $TAG --steps 3 --warmup 1 --cpu-sample 0 --parity-sample 0 --no-sparse-leg > $OUT/profile.log 2>&1; head -12 $OUT/profile.log
tools/pmc.sh ${TAG}_chol "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES" -- python $(pwd)/tools/bench_chol.py 1536 4096 f32 2 > $OUT/chol_pmc.txt 2>&1; tail -30 $OUT/chol_pmc.txt
