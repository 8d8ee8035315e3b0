"""Where does the host time of a tile-sparse LM run go?  cProfile of the timed forward of tools/bench_sparse.py's problem."""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import theseus_amd as th
from tests.test_gpu_sparse import _chain_problem

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
build = _chain_problem(th, P, B, torch.float32)
for packed in (True, False):
    opt = build(packed_factor=packed)
    opt.set_params(max_iterations=5)
    layer = th.TheseusLayer(opt)
    with torch.no_grad():
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            layer.forward(None, optimizer_kwargs=dict(damping=1e-2))
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"packed={packed} rep {rep}: host returned after {(t1 - t0) * 1e3:.1f} ms, device done after {(t2 - t0) * 1e3:.1f} ms")
        pr = cProfile.Profile()
        pr.enable()
        layer.forward(None, optimizer_kwargs=dict(damping=1e-2))
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(12)
