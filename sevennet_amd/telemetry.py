"""Shader clock, socket power and temperature of the GPU while a timed region runs (bench.py, tools/).

Why it exists: the two fused tensor-product kernels run at the socket's power cap, so the step time of one
build moves with the clock the box's power manager settles at -- a bench line has to carry the clock and the power it was
measured under, or a 5 % swing between two boxes cannot be told from a regression.

librocm_smi64 through ctypes (the library behind `rocm-smi`; no subprocess), read by a side thread every `period_s` while the
main thread enqueues and waits.  The readings are SMU queries that contend with command submission (sampling inside a timed
bracket of twenty 39.8-ms steps produced one 61-ms step), so bench.py samples over a REPEAT of its timed steps, outside the
bracket.  Everything is best effort: a box without the library or without the sensor reports `None` fields, never an exception
into the measurement.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import time
from typing import Dict, List, Optional

RSMI_MAX_NUM_FREQUENCIES = 33


class _Freqs(C.Structure):   # rsmi_frequencies_t (rocm_smi.h)
    _fields_ = [('has_deep_sleep', C.c_bool), ('num_supported', C.c_uint32), ('current', C.c_uint32),
                ('frequency', C.c_uint64 * RSMI_MAX_NUM_FREQUENCIES)]


_lib = None
_lib_tried = False


def _load():
    global _lib, _lib_tried
    if _lib_tried:
        return _lib
    _lib_tried = True
    roots = [os.environ.get('ROCM_PATH') or '/opt/rocm']
    for root in roots:
        for name in ('librocm_smi64.so.1', 'librocm_smi64.so'):
            p = os.path.join(root, 'lib', name)
            if os.path.isfile(p):
                try:
                    lib = C.CDLL(p)
                    if lib.rsmi_init(C.c_uint64(0)) == 0:
                        _lib = lib
                        return _lib
                except OSError:
                    pass
    return None


def sample(dev: int = 0) -> Dict[str, Optional[float]]:
    """one reading: sclk_mhz, socket_power_w, temp_edge_c / temp_junction_c (None where the sensor is not there)"""
    lib = _load()
    out: Dict[str, Optional[float]] = dict(sclk_mhz=None, socket_power_w=None, temp_edge_c=None, temp_junction_c=None)
    if lib is None:
        return out
    f = _Freqs()
    if lib.rsmi_dev_gpu_clk_freq_get(C.c_uint32(dev), C.c_int(0), C.byref(f)) == 0 and f.current < RSMI_MAX_NUM_FREQUENCIES:
        out['sclk_mhz'] = f.frequency[f.current] / 1e6
    p = C.c_uint64(0)
    if lib.rsmi_dev_current_socket_power_get(C.c_uint32(dev), C.byref(p)) == 0:
        out['socket_power_w'] = p.value / 1e6
    elif lib.rsmi_dev_power_ave_get(C.c_uint32(dev), C.c_uint32(0), C.byref(p)) == 0:
        out['socket_power_w'] = p.value / 1e6
    t = C.c_int64(0)
    for key, sensor in (('temp_edge_c', 0), ('temp_junction_c', 1)):
        if lib.rsmi_dev_temp_metric_get(C.c_uint32(dev), C.c_uint32(sensor), C.c_int(0), C.byref(t)) == 0:
            out[key] = t.value / 1e3
    return out


class Sampler:
    """with Sampler(dev) as s: <timed region> ; s.summary() -> medians + ranges over the samples taken meanwhile"""

    def __init__(self, dev: int = 0, period_s: float = 0.02):
        self.dev, self.period = dev, period_s
        self.rows: List[Dict[str, Optional[float]]] = []
        self._stop = threading.Event()
        self._thr: Optional[threading.Thread] = None

    def _run(self):
        while not self._stop.is_set():
            self.rows.append(sample(self.dev))
            self._stop.wait(self.period)

    def __enter__(self):
        if _load() is not None:
            self._thr = threading.Thread(target=self._run, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thr is not None:
            self._thr.join(timeout=2.0)
        return False

    def summary(self) -> Dict[str, object]:
        def med(key):
            v = sorted(r[key] for r in self.rows if r.get(key) is not None)
            return (round(v[len(v) // 2], 1), round(v[0], 1), round(v[-1], 1)) if v else (None, None, None)
        out: Dict[str, object] = dict(telemetry_samples=len(self.rows),
                                      telemetry_source='librocm_smi64 (side thread, one reading per %d ms over a repeat of the timed steps)' % int(self.period * 1e3)
                                      if _load() is not None else 'unavailable (librocm_smi64 not loadable)')
        for key in ('sclk_mhz', 'socket_power_w', 'temp_edge_c', 'temp_junction_c'):
            m, lo, hi = med(key)
            out[key] = m
            if m is not None and key in ('sclk_mhz', 'socket_power_w'):
                out[key + '_min_max'] = [lo, hi]
        return out


if __name__ == '__main__':   # python -m sevennet_amd.telemetry : one reading
    t0 = time.perf_counter()
    print(sample(0), f'{(time.perf_counter() - t0) * 1e3:.2f} ms')
