"""Socket power and shader clock WHILE the factorisation runs (is the MFMA-bound phase power-managed?): a sampler thread reads the
amdgpu hwmon / pp_dpm files every few ms while the main thread queues thx_chol_factor calls back to back.
usage: python tools/power_sample.py [f32|f64] [seconds] [batch]"""
import glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from theseus_amd.kernels import default_kernels, round_up

dt = {"f32": torch.float32, "f64": torch.float64}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
n = 1536


def find(pattern):
    hits = sorted(glob.glob(pattern))
    return hits[0] if hits else None


dev = find("/sys/class/drm/card*/device/pp_dpm_sclk")
dev = os.path.dirname(dev) if dev else None
f_power = find(f"{dev}/hwmon/hwmon*/power1_average") or find(f"{dev}/hwmon/hwmon*/power1_input") if dev else None
f_cap = find(f"{dev}/hwmon/hwmon*/power1_cap") if dev else None
f_freq = find(f"{dev}/hwmon/hwmon*/freq1_input") if dev else None
f_temp = find(f"{dev}/hwmon/hwmon*/temp*_input") if dev else None
print("device", dev, "| power", f_power, "| cap", f_cap, "| sclk", f_freq, "| temp", f_temp)


def rd(p):
    try:
        with open(p) as fh:
            return float(fh.read().split()[0])
    except Exception:
        return float("nan")


samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        samples.append((time.perf_counter(), rd(f_power) / 1e6 if f_power else float("nan"),
                        rd(f_freq) / 1e6 if f_freq else float("nan"), rd(f_temp) / 1e3 if f_temp else float("nan")))
        time.sleep(0.004)


K = default_kernels(); ld = round_up(n, 32); nt = n // 128
gen = torch.Generator(device="cuda").manual_seed(0)
H = torch.empty(B, ld, ld, dtype=dt, device="cuda"); H.uniform_(-1, 1, generator=gen)
H.diagonal(dim1=1, dim2=2).add_(float(n))
L = torch.zeros_like(H); P = torch.empty(B, nt, 128, 128, dtype=dt, device="cuda")
info = torch.empty(B, dtype=torch.int32, device="cuda"); lam = torch.full((B,), 1e-3, dtype=dt, device="cuda")
K.chol_factor(H, n, lam, False, 1e-8, L, P, info); torch.cuda.synchronize()
time.sleep(1.0)
th = threading.Thread(target=sampler); th.start()
time.sleep(0.5)     # idle baseline
t_start = time.perf_counter()
calls, per_call = 0, []
while time.perf_counter() - t_start < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        K.chol_factor(H, n, lam, False, 1e-8, L, P, info)
    e1.record(); torch.cuda.synchronize()
    per_call.append((time.perf_counter() - t_start, e0.elapsed_time(e1) / 5)); calls += 5
t_end = time.perf_counter()
time.sleep(0.5)
stop.set(); th.join()
cap = rd(f_cap) / 1e6 if f_cap else float("nan")
idle = [s for s in samples if s[0] < t_start]
busy = [s for s in samples if t_start + 0.3 < s[0] < t_end]
mean = lambda xs: sum(xs) / max(len(xs), 1)
print(f"{dt} n={n} B={B}: {calls} factor calls in {t_end - t_start:.1f} s; cap {cap:.0f} W")
print(f"idle : power {mean([s[1] for s in idle]):7.1f} W  sclk {mean([s[2] for s in idle]):7.1f} MHz  temp {mean([s[3] for s in idle]):5.1f} C  ({len(idle)} samples)")
print(f"busy : power {mean([s[1] for s in busy]):7.1f} W (max {max(s[1] for s in busy):.1f})  sclk {mean([s[2] for s in busy]):7.1f} MHz "
      f"(min {min(s[2] for s in busy):.0f} max {max(s[2] for s in busy):.0f})  temp {mean([s[3] for s in busy]):5.1f} C  ({len(busy)} samples)")
for k in range(0, len(per_call), max(1, len(per_call) // 8)):
    t, ms = per_call[k]
    near = [s for s in busy if abs(s[0] - t_start - t) < 0.15]
    print(f"  t = {t:5.2f} s: {ms:7.3f} ms per factor call; power {mean([s[1] for s in near]):7.1f} W sclk {mean([s[2] for s in near]):7.1f} MHz")
