#!/bin/bash
# the unrolled / generic GPU tests after the initial-value-gradient change (python side only), before the final set
mkdir -p gpurun_out/r4q
timeout 900 python -m pytest tests/test_gpu_unrolled.py tests/test_gpu_generic.py -q -m gpu -p no:cacheprovider > gpurun_out/r4q/pytest.txt 2>&1
tail -5 gpurun_out/r4q/pytest.txt
