"""Socket power / shader clock of the factorisation next to its pieces (librocm_smi64 sampler + energy counter, tools/smi.py):
idle, an HBM copy, thx_chol_factor fp32 (two streams / one) and fp64, and the modes of tools/microbench/power_pieces
(theseus_amd/lib/variants/power_pieces, built by tools/gpu_round5_t.sh).  usage: python tools/power_model.py [seconds]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smi import Smi, Sampler
from theseus_amd.kernels import default_kernels, round_up

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
smi = Smi()
print(f"power cap {smi.cap_w():.0f} W")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def measure(name, body, setup=None):
    """body(deadline, state) runs the workload until time.perf_counter() > deadline and returns a result string; setup() (outside the
    measured window) builds its state."""
    state = setup() if setup else None
    torch.cuda.synchronize()
    time.sleep(1.5)      # let the previous workload's averaging window drain
    with Sampler(smi, 0.01) as sp:
        t0 = time.perf_counter(); e0 = smi.energy_j()
        res = body(t0 + secs, state)
        t1 = time.perf_counter(); e1 = smi.energy_j()
    w = sp.window(t0 + 0.4 * (t1 - t0), t1 - 0.05)     # the second part of the run: the averaging filter has settled
    print(f"{name:58s} energy-counter mean {(e1 - e0) / (t1 - t0):7.1f} W | sampled (last 60 %) power {w['power_mean']:7.1f} W max {w['power_max']:7.1f}  "
          f"sclk {w['sclk_mean']:6.0f} MHz (min {w['sclk_min']:.0f} max {w['sclk_max']:.0f}, {w['n']} samples) | {res}", flush=True)
    del state
    torch.cuda.empty_cache()


def idle(deadline, state):
    while time.perf_counter() < deadline:
        time.sleep(0.05)
    return ""


def copy_setup():
    a = torch.empty(4 << 30, dtype=torch.uint8, device="cuda")
    return a, torch.empty_like(a)


def copy(deadline, state):
    a, b = state
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() < deadline:
        for _ in range(20):
            b.copy_(a)
        torch.cuda.synchronize(); n += 20
    dt = time.perf_counter() - t0
    return f"{2 * n * a.numel() / dt / 1e12:.2f} TB/s read + write"


def factor_setup(dt):
    def setup():
        n, B = 1536, 4096
        K = default_kernels(); ld = round_up(n, 32); nt = n // 128
        gen = torch.Generator(device="cuda").manual_seed(0)
        H = torch.empty(B, ld, ld, dtype=dt, device="cuda"); H.uniform_(-1, 1, generator=gen)
        H.diagonal(dim1=1, dim2=2).add_(float(n))
        L = torch.zeros_like(H); P = torch.empty(B, nt, 128, 128, dtype=dt, device="cuda")
        info = torch.empty(B, dtype=torch.int32, device="cuda"); lam = torch.full((B,), 1e-3, dtype=dt, device="cuda")
        K.chol_factor(H, n, lam, False, 1e-8, L, P, info); torch.cuda.synchronize()
        return K, H, n, B, lam, L, P, info
    return setup


def factor(deadline, state):
    K, H, n, B, lam, L, P, info = state
    t0 = time.perf_counter(); calls = 0
    while time.perf_counter() < deadline or calls == 0:
        for _ in range(5):
            K.chol_factor(H, n, lam, False, 1e-8, L, P, info)
        torch.cuda.synchronize(); calls += 5
    ms = (time.perf_counter() - t0) / calls * 1e3
    peak = 157.3 if H.dtype == torch.float32 else 78.6
    return f"{ms:.2f} ms per call = {B * n ** 3 / 3 / ms / 1e9:.1f} TFLOP/s ({B * n ** 3 / 3 / ms / 1e9 / peak:.3f})"


def pieces(mode):
    def body(deadline, state):
        out = subprocess.run([os.path.join(ROOT, "theseus_amd/lib/variants/power_pieces"), str(mode), str(secs - 0.6)],
                             capture_output=True, text=True, timeout=120)
        return out.stdout.strip().replace("\n", " ")
    return body


which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["idle", "copy", "f32", "f64", "pieces"]
if "idle" in which:
    measure("idle", idle)
if "copy" in which:
    measure("torch copy 4 GiB -> 4 GiB (HBM read + write)", copy, copy_setup)
if "f32" in which:
    measure("thx_chol_factor fp32 n=1536 B=4096 (two streams)", factor, factor_setup(torch.float32))
if "f64" in which:
    measure("thx_chol_factor fp64 n=1536 B=4096 (two streams)", factor, factor_setup(torch.float64))
if "pieces" in which:
    for mode, what in ((0, "MFMAs only (operands in registers)"), (1, "+ LDS fragment reads"), (2, "+ staging stores, barriers"),
                       (3, "+ operand stream from HBM = the K-loop"), (5, "the K-loop, all operands 0.0115"), (4, "operand stream alone, no MFMAs")):
        measure(f"power_pieces mode {mode}: {what}", pieces(mode))
