"""Clock / power / temperature / throttle telemetry of one GPU while something runs on it — so that a number measured on one box can be
compared with a number measured on another (the same attention kernel ran 228-243 ms per launch on different boxes and the
bench line carried nothing that said why).

In-process: a daemon thread calls libamd_smi (the `amdsmi` python binding that ships with ROCm; one sysfs `gpu_metrics` read per sample, no
child process, nothing on the GPU) every `period_s` seconds.  What a summary holds:
  sclk_mhz_mean / _min / _max   mean over the XCDs' current_gfxclks, then over the samples
  power_w_mean / _max, power_cap_w
  temp_c_max (hotspot), hbm_temp_c_max
  throttle_bits                 OR over the samples of gpu_metrics.throttle_status / indep_throttle_status
  residency                     the firmware's own violation accumulators (amdsmi_get_violation_status / gpu_metrics *_residency_acc), as the
                                FRACTION of the sampled interval each limiter was active: ppt_pwr (package power), socket_thrm, vr_thrm,
                                hbm_thrm, prochot, and per XCD gfx_clk_below_host_limit_{pwr,thm,total} + low_utilization — this is the
                                reading that names the limiter
  energy_j                      energy_accumulator delta (15.259 uJ units)
Everything is optional: a key the driver does not report is left out, a box without the library gives {"available": false, "reason": ...}.

CLI:  python tools/gpu_telemetry.py [--period 0.25] [--label X] [--raw] -- <command ...>     runs the command, prints ONE json line
"""
import json
import subprocess
import sys
import threading
import time

_BAD = (None, 'N/A', 0xFFFF, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF)


def _num(v):
    if isinstance(v, bool) or v in _BAD:
        return None
    if isinstance(v, (int, float)):
        return v
    return None


def _nums(vs):
    out = []
    if isinstance(vs, (list, tuple)):
        for v in vs:
            if isinstance(v, (list, tuple)):
                out.extend(_nums(v))
            else:
                n = _num(v)
                if n is not None:
                    out.append(n)
    return out


class GpuTelemetry:
    ACC_KEYS = ('accumulation_counter', 'prochot_residency_acc', 'ppt_residency_acc', 'socket_thm_residency_acc', 'vr_thm_residency_acc',
                'hbm_thm_residency_acc', 'energy_accumulator')

    def __init__(self, device_index=0, period_s=0.5):
        self.period_s = period_s
        self.available, self.reason = False, None
        self._smi = self._h = None
        self._samples = []
        self._thread = None
        self._stop = threading.Event()
        self._acc0 = self._viol0 = self._t0 = None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            if not hs:
                raise RuntimeError('no processor handles')
            self._smi, self._h = amdsmi, hs[min(device_index, len(hs) - 1)]
            self.available = True
        except Exception as e:      # noqa: BLE001 — no driver (CPU container), no library, no permission: report, do not fail
            self.reason = f'{type(e).__name__}: {str(e).strip()[:120]}'

    # ---- one reading ----------------------------------------------------------------------------------------------------------------
    def metrics(self):
        try:
            return self._smi.amdsmi_get_gpu_metrics_info(self._h)
        except Exception:           # noqa: BLE001
            return {}

    def violations(self):
        try:
            return self._smi.amdsmi_get_violation_status(self._h)
        except Exception:           # noqa: BLE001
            return {}

    def power_cap_w(self):
        try:
            c = self._smi.amdsmi_get_power_cap_info(self._h)
            v = _num(c.get('power_cap'))
            return None if v is None else (v / 1e6 if v > 1e5 else float(v))
        except Exception:           # noqa: BLE001
            return None

    def sample(self):
        m = self.metrics()
        clks = [c for c in _nums(m.get('current_gfxclks')) if c > 0] or [c for c in _nums([m.get('current_gfxclk')]) if c > 0]
        s = {'t': time.time(),
             'sclk': sum(clks) / len(clks) if clks else None,
             'power': _num(m.get('current_socket_power')) or _num(m.get('average_socket_power')),
             'temp': _num(m.get('temperature_hotspot')),
             'hbm_temp': max(_nums(m.get('temperature_hbm')) + _nums([m.get('temperature_mem')]), default=None),
             'throttle': _num(m.get('throttle_status')),
             'indep_throttle': _num(m.get('indep_throttle_status')),
             'vgfx': _num(m.get('voltage_gfx')),
             'uclk': _num(m.get('current_uclk'))}
        return s, m

    # ---- interval -------------------------------------------------------------------------------------------------------------------
    def start(self):
        if not self.available or self._thread is not None:
            return self
        self._samples, self._t0 = [], time.time()
        _, m = self.sample()
        self._acc0 = {k: _num(m.get(k)) for k in self.ACC_KEYS}
        self._viol0 = self.violations()
        self._stop.clear()

        def loop():
            while not self._stop.wait(self.period_s):
                try:
                    self._samples.append(self.sample()[0])
                except Exception:   # noqa: BLE001
                    pass
        self._thread = threading.Thread(target=loop, name='gpu-telemetry', daemon=True)
        self._thread.start()
        return self

    def stop(self):
        if not self.available:
            return {'available': False, 'reason': self.reason}
        if self._thread is None:
            return {'available': True, 'n_samples': 0}
        self._stop.set()
        self._thread.join(timeout=5)
        self._thread = None
        s_end, m = self.sample()
        acc1 = {k: _num(m.get(k)) for k in self.ACC_KEYS}
        viol1 = self.violations()
        return self._summary(self._samples + [s_end], self._acc0, acc1, self._viol0, viol1, time.time() - self._t0)

    def _summary(self, ss, a0, a1, v0, v1, wall_s):
        def col(k):
            return [s[k] for s in ss if s.get(k) is not None]
        out = {'available': True, 'n_samples': len(ss), 'period_s': self.period_s, 'interval_s': round(wall_s, 2)}
        for key, k in (('sclk_mhz', 'sclk'), ('power_w', 'power')):
            c = col(k)
            if c:
                out[key + '_mean'], out[key + '_min'], out[key + '_max'] = round(sum(c) / len(c), 1), round(min(c), 1), round(max(c), 1)
        for key, k in (('temp_c_max', 'temp'), ('hbm_temp_c_max', 'hbm_temp'), ('uclk_mhz_max', 'uclk')):
            c = col(k)
            if c:
                out[key] = max(c)
        c = col('vgfx')
        if c:
            out['vgfx_mv_mean'] = round(sum(c) / len(c), 1)
        bits = 0
        for v in col('throttle'):
            bits |= int(v)
        ibits = 0
        for v in col('indep_throttle'):
            ibits |= int(v)
        out['throttle_bits'] = bits
        out['indep_throttle_bits'] = ibits
        cap = self.power_cap_w()
        if cap:
            out['power_cap_w'] = cap
        # firmware violation accumulators -> fraction of the interval each limiter was active
        res = {}
        d = (a1.get('accumulation_counter') or 0) - (a0.get('accumulation_counter') or 0) if a0 and a1 else 0
        if d > 0:
            for k in ('prochot', 'ppt', 'socket_thm', 'vr_thm', 'hbm_thm'):
                x0, x1 = a0.get(k + '_residency_acc'), a1.get(k + '_residency_acc')
                if x0 is not None and x1 is not None:
                    res[k] = round((x1 - x0) / d, 4)
        if a0 and a1 and a0.get('energy_accumulator') is not None and a1.get('energy_accumulator') is not None:
            out['energy_j'] = round((a1['energy_accumulator'] - a0['energy_accumulator']) * 15.259e-6, 1)
        if v0 and v1:
            dv = (_num(v1.get('acc_counter')) or 0) - (_num(v0.get('acc_counter')) or 0)
            if dv > 0:
                for k in ('prochot_thrm', 'ppt_pwr', 'socket_thrm', 'vr_thrm', 'hbm_thrm', 'gfx_clk_below_host_limit'):
                    x0, x1 = _num(v0.get('acc_' + k)), _num(v1.get('acc_' + k))
                    if x0 is not None and x1 is not None:
                        res['viol_' + k] = round((x1 - x0) / dv, 4)
                for k in ('gfx_clk_below_host_limit_pwr', 'gfx_clk_below_host_limit_thm', 'gfx_clk_below_host_limit_total', 'low_utilization'):
                    x0, x1 = _nums(v0.get('acc_' + k)), _nums(v1.get('acc_' + k))
                    if x0 and len(x0) == len(x1):
                        fr = [(b - a) / dv for a, b in zip(x0, x1)]
                        res['viol_' + k + '_xcd_mean'] = round(sum(fr) / len(fr), 4)
                        res['viol_' + k + '_xcd_max'] = round(max(fr), 4)
            act = {k[7:]: v1[k] for k in v1 if k.startswith('active_') and v1[k] not in _BAD and not isinstance(v1[k], (list, tuple)) and v1[k]}
            if act:
                out['active_at_end'] = sorted(act)
        if res:
            out['residency'] = res
        return out

    def run(self, fn):
        """telemetry of one call: (fn's result, summary)"""
        self.start()
        try:
            r = fn()
        finally:
            s = self.stop()
        return r, s


def main(argv):
    period, label, raw = 0.25, None, False
    while argv and argv[0] != '--':
        a = argv.pop(0)
        if a == '--period':
            period = float(argv.pop(0))
        elif a == '--label':
            label = argv.pop(0)
        elif a == '--raw':
            raw = True
    cmd = argv[1:]
    t = GpuTelemetry(period_s=period)
    if raw and t.available:
        print(json.dumps({'raw_metrics': {k: (v if not isinstance(v, (list, tuple)) else list(v)) for k, v in t.metrics().items()},
                          'raw_violations': t.violations()}, default=str))
    t.start()
    rc = subprocess.call(cmd) if cmd else (time.sleep(2) or 0)
    s = t.stop()
    s['label'], s['rc'] = label or (cmd[0] if cmd else 'idle'), rc
    print(json.dumps(s))
    return rc


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
