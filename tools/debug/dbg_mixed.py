import sys, numpy as np, torch
sys.path.insert(0, '.')
import theseus_amd as th
from tests.helpers import load_golden
from tests.mixed_robust_common import run_mixed_implicit, GRAD_KEYS, specs
g = load_golden("pg_f64_mixed_robust")
g32 = {k: (v.astype(np.float32).astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float64 else v) for k, v in g.items()}
r32 = run_mixed_implicit(th, g, "cuda", dtype=torch.float32)
r64 = run_mixed_implicit(th, g32, "cuda", dtype=torch.float64)
print("final diff f32 vs f64(rounded inputs)", (r32["final"].double() - r64["final"]).abs().max().item(), "vs ref", (r32["final"].double().numpy() - g["final"]).max())
print("err hist f32", r32["info"].err_history[0].tolist())
print("err hist f64", r64["info"].err_history[0].tolist())
for key, ref in GRAD_KEYS:
    a, b, w = r32["grads"][key].double().numpy(), r64["grads"][key].numpy(), g[ref]
    print(key, "f32 vs f64r", np.abs(a - b).max() / np.abs(b).max(), "f64r vs ref", np.abs(b - w).max() / np.abs(w).max(), "scale", np.abs(w).max())
a, b = r32["grads"]["meas"].double().numpy(), r64["grads"]["meas"].numpy()
d = np.abs(a - b).reshape(a.shape[0], a.shape[1], -1).max(-1)
print("per-cost max diff (batch x edge):\n", np.round(d / np.abs(b).max(), 3))
print("specs", specs(g, "between"))
