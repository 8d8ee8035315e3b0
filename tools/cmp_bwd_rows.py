"""Debug aid: the backward solve of one random factor through whichever kernel THX_CHOL_BWD_ROWS_MAX_BATCH selects; saves x.
usage: THX_CHOL_BWD_ROWS_MAX_BATCH=0|100000 python tools/cmp_bwd_rows.py out.pt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import theseus_amd as th

K = th.default_kernels()
B, n = 8, 1536
gen = torch.Generator().manual_seed(0)
A = torch.randn(B, n, n + 8, dtype=torch.float64, generator=gen)
M = (A @ A.transpose(1, 2) / (n + 8) + 1e-3 * torch.eye(n, dtype=torch.float64)).float()
H = torch.tril(M).cuda().contiguous()
rhs = torch.randn(B, n, generator=gen).cuda()
L = torch.zeros_like(H)
panels = torch.empty(B, n // 128, 128, 128, device="cuda")
info = torch.empty(B, dtype=torch.int32, device="cuda")
y, x = torch.empty_like(rhs), torch.empty_like(rhs)
K.chol_right_looking_max_batch(0)
K.chol_factor(H, n, None, False, 1e-8, L, panels, info, rhs=rhs, y=y)
K.chol_solve_backward(L, n, panels, y, x)
torch.cuda.synchronize()
torch.save(dict(x=x.cpu(), y=y.cpu()), sys.argv[1])
