"""Socket power, energy counter and shader clock of GPU 0 through librocm_smi64 (ctypes; the amdgpu hwmon files of this
image return a stale 290 W / 2398 MHz while rocm-smi itself reports 1356 W / 2207 MHz under the factorisation -- measured in
profiles/r5/q_rocm_smi_during_factor.txt).  Sampler: a thread polling power + sclk every `period` seconds; the energy accumulator gives the exact mean."""
import ctypes, threading, time


class _Freqs(ctypes.Structure):
    _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                ("frequency", ctypes.c_uint64 * 33)]


class Smi:
    def __init__(self, dev=0):
        self.lib = ctypes.CDLL("/opt/rocm/lib/librocm_smi64.so")
        rc = self.lib.rsmi_init(ctypes.c_uint64(0))
        if rc:
            raise RuntimeError(f"rsmi_init -> {rc}")
        self.dev = ctypes.c_uint32(dev)

    def power_w(self):
        p, t = ctypes.c_uint64(0), ctypes.c_int(0)
        if self.lib.rsmi_dev_power_get(self.dev, ctypes.byref(p), ctypes.byref(t)):
            return float("nan")
        return p.value / 1e6

    def cap_w(self):
        p = ctypes.c_uint64(0)
        if self.lib.rsmi_dev_power_cap_get(self.dev, ctypes.c_uint32(0), ctypes.byref(p)):
            return float("nan")
        return p.value / 1e6

    def sclk_mhz(self, clk_type=0):
        f = _Freqs()
        if self.lib.rsmi_dev_gpu_clk_freq_get(self.dev, ctypes.c_int(clk_type), ctypes.byref(f)):
            return float("nan")
        return f.frequency[min(f.current, 32)] / 1e6

    def energy_j(self):
        e, res, ts = ctypes.c_uint64(0), ctypes.c_float(0), ctypes.c_uint64(0)
        if self.lib.rsmi_dev_energy_count_get(self.dev, ctypes.byref(e), ctypes.byref(res), ctypes.byref(ts)):
            return float("nan")
        return e.value * res.value / 1e6      # counter x resolution (micro-joules)


class Sampler:
    def __init__(self, smi, period=0.01):
        self.smi, self.period, self.samples = smi, period, []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            self.samples.append((time.perf_counter(), self.smi.power_w(), self.smi.sclk_mhz()))
            time.sleep(self.period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._th.join()

    def window(self, t0, t1):
        w = [s for s in self.samples if t0 <= s[0] <= t1]
        n = max(len(w), 1)
        ps, fs = [s[1] for s in w], [s[2] for s in w]
        return {"n": len(w), "power_mean": sum(ps) / n, "power_max": max(ps) if ps else float("nan"),
                "sclk_mean": sum(fs) / n, "sclk_min": min(fs) if fs else float("nan"), "sclk_max": max(fs) if fs else float("nan")}
