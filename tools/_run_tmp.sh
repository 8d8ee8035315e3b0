timeout 1500 python -m pytest tests/test_gpu_ba.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -15
