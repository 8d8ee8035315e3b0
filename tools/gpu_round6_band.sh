#!/bin/bash
# the two noise-level parity tests that moved with the rounding of potrf_inv32_lanes: their figures under both schemes
for v in base blocked; do
  if [ $v = base ]; then unset THESEUS_HIP_LIB; else export THESEUS_HIP_LIB=$PWD/theseus_amd/lib/variants/$v.so; fi
  echo "== $v"
  timeout 600 python -m pytest tests/test_gpu_lm.py -m gpu -q -s -k "inside_reference_band" -p no:cacheprovider 2>&1 | grep "^\[\|passed\|failed"
  timeout 600 python -m pytest tests/test_gpu_full_size.py -m gpu -q -s -k "implicit_gradients_match" -p no:cacheprovider 2>&1 | grep "^\[\|passed\|failed"
done
