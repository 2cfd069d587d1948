#!/usr/bin/env python3
"""Shader clock / socket power / temperature of one GPU sampled from a side thread while something else runs (bench.py's timed region).

The same kernel has read 5.02 ... 5.54 ms per launch on different boxes of the pool (missing #5); this records what the box was doing, so a bench line can be
read against the clock it was measured at.  Sources, in order: the amdsmi Python binding (gpu_metrics: per-XCD gfx clocks, socket power, hotspot temperature, throttle /
violation status), then the `rocm-smi --json` command line.  Nothing here touches the device's settings.

    with Sampler(0, period_s=0.5) as s:        # device ordinal
        ...timed region...
    report = s.summary()                        # {"source": ..., "samples": n, "sclk_mhz": {"min", "median", "max"}, "power_w": {...}, ...}

    python tools/gpu_telemetry.py [seconds]     # print raw samples of device 0 (what the first source returns on this box)
"""
import json
import subprocess
import sys
import threading
import time


def _stats(xs):
    xs = sorted(x for x in xs if x is not None)
    if not xs:
        return None
    n = len(xs)
    med = xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])
    return {"min": round(xs[0], 1), "median": round(med, 1), "max": round(xs[-1], 1)}


def _num(v):
    """amdsmi reports 'N/A' strings, 0xFFFF sentinels and plain numbers"""
    if isinstance(v, (int, float)) and not isinstance(v, bool):
        return None if v in (0xFFFF, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF) else float(v)
    return None


class _AmdSmi:
    name = "amdsmi"

    def __init__(self, ordinal):
        import amdsmi
        self.m = amdsmi
        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        if not hs:
            raise RuntimeError("amdsmi: no processors")
        self.h = hs[min(ordinal, len(hs) - 1)]
        self.clk_type = getattr(amdsmi.AmdSmiClkType, "GFX", None) or amdsmi.AmdSmiClkType.SYS
        self.sample()                      # raises if nothing usable comes back

    def sample(self):
        m, out = self.m, {}
        try:
            g = m.amdsmi_get_gpu_metrics_info(self.h)
            clks = [c for c in (_num(x) for x in (g.get("current_gfxclks") or [])) if c]
            if clks:
                out["sclk_mhz"] = sum(clks) / len(clks)
                out["sclk_mhz_min_xcd"] = min(clks)
            elif _num(g.get("current_gfxclk")):
                out["sclk_mhz"] = _num(g.get("current_gfxclk"))
            for key, name in (("current_socket_power", "power_w"), ("average_socket_power", "power_w"), ("temperature_hotspot", "temp_hotspot_c"),
                              ("temperature_mem", "temp_mem_c"), ("average_gfx_activity", "gfx_activity_pct"), ("current_uclk", "mclk_mhz"),
                              ("throttle_status", "throttle_status"), ("indep_throttle_status", "indep_throttle_status")):
                v = _num(g.get(key))
                if v is not None and name not in out:
                    out[name] = v
            # firmware accumulators (monotonic): their growth over the window says WHICH limiter held the clock down (power / thermal / VR / HBM thermal / PROCHOT)
            for key in ("accumulation_counter", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc", "prochot_residency_acc", "energy_accumulator"):
                v = _num(g.get(key))
                if v is not None:
                    out["acc_" + key] = v
        except Exception:
            pass
        if "sclk_mhz" not in out:
            c = m.amdsmi_get_clock_info(self.h, self.clk_type)
            v = _num(c.get("clk"))
            if v is None:
                raise RuntimeError("amdsmi: no gfx clock")
            out["sclk_mhz"] = v
            out.setdefault("sclk_max_mhz", _num(c.get("max_clk")))
        if "power_w" not in out:
            try:
                p = m.amdsmi_get_power_info(self.h)
                for key in ("current_socket_power", "average_socket_power", "socket_power"):
                    v = _num(p.get(key))
                    if v:
                        out["power_w"] = v
                        break
            except Exception:
                pass
        return out

    def static(self):
        out = {}
        try:
            c = self.m.amdsmi_get_clock_info(self.h, self.clk_type)
            out["sclk_max_mhz"], out["sclk_min_mhz"] = _num(c.get("max_clk")), _num(c.get("min_clk"))
        except Exception:
            pass
        try:
            p = self.m.amdsmi_get_power_cap_info(self.h)
            v = _num(p.get("power_cap"))
            if v:
                out["power_cap_w"] = v / 1e6 if v > 1e5 else v
        except Exception:
            pass
        return out


class _RocmSmiCli:
    name = "rocm-smi --json"

    def __init__(self, ordinal):
        self.card = "card%d" % ordinal
        self.sample()

    def sample(self):
        r = subprocess.run(["rocm-smi", "-d", self.card[4:], "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        d = d.get(self.card) or next(iter(d.values()))
        out = {}
        for k, v in d.items():
            kl = k.lower()
            try:
                if "sclk" in kl and "(" in str(v):
                    out["sclk_mhz"] = float(str(v).split("(")[1].split("M")[0])
                elif "power" in kl and "socket" in kl:
                    out["power_w"] = float(v)
                elif "temperature" in kl and ("junction" in kl or "hotspot" in kl):
                    out["temp_hotspot_c"] = float(v)
            except Exception:
                pass
        if "sclk_mhz" not in out:
            raise RuntimeError("rocm-smi: no sclk in %r" % list(d)[:8])
        return out

    def static(self):
        return {}


class Sampler:
    def __init__(self, ordinal=0, period_s=0.5):
        self.period, self.samples, self.src, self.err = period_s, [], None, None
        self._stop, self._t = threading.Event(), None
        for cls in (_AmdSmi, _RocmSmiCli):
            try:
                self.src = cls(ordinal)
                break
            except Exception as e:                      # noqa: BLE001 -- telemetry is optional: the bench line says "unavailable" and why
                self.err = "%s: %s" % (cls.__name__, str(e)[:160])

    def _run(self):
        while not self._stop.is_set():
            try:
                s = self.src.sample()
                s["t"] = time.perf_counter()
                self.samples.append(s)
            except Exception as e:                      # noqa: BLE001
                self.err = str(e)[:160]
            self._stop.wait(self.period)

    def __enter__(self):
        if self.src is not None:
            self._t = threading.Thread(target=self._run, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._t:
            self._t.join(timeout=30)

    def summary(self):
        if self.src is None:
            return {"source": None, "error": self.err}
        out = {"source": self.src.name, "samples": len(self.samples), "period_s": self.period}
        for key in ("sclk_mhz", "sclk_mhz_min_xcd", "power_w", "temp_hotspot_c", "temp_mem_c", "gfx_activity_pct", "mclk_mhz"):
            st = _stats([s.get(key) for s in self.samples])
            if st:
                out[key] = st
        first, last = (self.samples[0], self.samples[-1]) if len(self.samples) >= 2 else ({}, {})
        ticks = (last.get("acc_accumulation_counter") or 0) - (first.get("acc_accumulation_counter") or 0)
        if ticks > 0:
            res = {}
            for key in ("ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc", "prochot_residency_acc"):
                if "acc_" + key in first and "acc_" + key in last:
                    res[key.replace("_residency_acc", "")] = round((last["acc_" + key] - first["acc_" + key]) / ticks, 4)
            out["limiter_residency"] = dict(res, note="share of the firmware's sampling ticks inside the window during which that limiter was active (gpu_metrics *_residency_acc / accumulation_counter)")
        thr = sorted({int(s["throttle_status"]) for s in self.samples if s.get("throttle_status") is not None})
        if thr:
            out["throttle_status_values"] = thr
        out.update({k: v for k, v in self.src.static().items() if v is not None})
        if self.err:
            out["last_error"] = self.err
        return out


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    with Sampler(0, 0.5) as s:
        if s.src is not None and isinstance(s.src, _AmdSmi):
            try:
                g = s.src.m.amdsmi_get_gpu_metrics_info(s.src.h)
                print("gpu_metrics keys:", {k: g[k] for k in sorted(g)})
            except Exception as e:                      # noqa: BLE001
                print("gpu_metrics failed:", e)
        time.sleep(secs)
    print(json.dumps(s.samples[:4]))
    print(json.dumps(s.summary()))
