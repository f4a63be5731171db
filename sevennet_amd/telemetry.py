"""Shader clock, socket power and temperature of the GPU while a timed region runs (bench.py, tools/).

Why it exists: the two fused tensor-product kernels run at the socket's power cap, so the step time of one
build moves with the clock the box's power manager settles at -- a bench line has to carry the clock and the power it was
measured under, or a 5 % swing between two boxes cannot be told from a regression.

librocm_smi64 through ctypes (the library behind `rocm-smi`), read every `period_s` by a child process while the measured
process enqueues and waits.  The readings are SMU queries that contend with command submission (sampling inside a timed
bracket of twenty 39.8-ms steps produced one 61-ms step), so bench.py samples over a REPEAT of its timed steps, outside the
bracket.  Everything is best effort: a box without the library or without the sensor reports `None` fields, never an exception
into the measurement.
"""
from __future__ import annotations

import ctypes as C
import os
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


def physical_index(visible_index: int) -> int:
    """rocm_smi numbers the node's GPUs; a process under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES numbers its own subset"""
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v:
            ids = [x.strip() for x in v.split(',')]
            if visible_index < len(ids) and ids[visible_index].isdigit():
                return int(ids[visible_index])
    return visible_index


class Sampler:
    """with Sampler(dev) as s: <region> ; s.summary() -> medians + ranges over the samples taken meanwhile.
    The readings come from a CHILD process (`python -m sevennet_amd.telemetry --watch`): librocm_smi64 initialised a second time
    inside a process that also runs RCCL (which holds its own rocm_smi session) aborted at exit -- SIGABRT in the world-1 RCCL soak --,
    and a child cannot hold the GIL or a driver lock of the measured process either."""

    def __init__(self, dev: int = 0, period_s: float = 0.02):
        self.dev, self.period = physical_index(dev), period_s
        self.rows: List[Dict[str, Optional[float]]] = []
        self._proc = None
        self._ok = False

    def __enter__(self):
        import subprocess
        import sys
        try:
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            self._proc = subprocess.Popen([sys.executable, '-m', 'sevennet_amd.telemetry', '--watch', str(self.period), '--dev', str(self.dev)],
                                          cwd=root, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            import select
            r, _, _ = select.select([self._proc.stdout], [], [], 20.0)   # a child stuck in rsmi_init must not hang the measurement
            if not r:
                self._proc.kill()
                self._proc = None
                return self
            first = self._proc.stdout.readline()          # 'ready' once the library is initialised (or 'unavailable')
            self._ok = first.strip() == 'ready'
        except Exception:  # noqa: BLE001
            self._proc = None
        return self

    def __exit__(self, *exc):
        import json
        if self._proc is not None:
            try:
                out, _ = self._proc.communicate(input='stop\n', timeout=5.0)
                self.rows = [json.loads(ln) for ln in out.splitlines() if ln.startswith('{')]
            except Exception:  # noqa: BLE001
                self._proc.kill()
        return False

    def summary(self) -> Dict[str, object]:
        def med(key):
            v = sorted(r[key] for r in self.rows if r.get(key) is not None)
            return (round(v[len(v) // 2], 1), round(v[0], 1), round(v[-1], 1)) if v else (None, None, None)
        out: Dict[str, object] = dict(telemetry_samples=len(self.rows),
                                      telemetry_source='librocm_smi64 in a child process, one reading per %d ms over a repeat of the timed steps' % int(self.period * 1e3)
                                      if self._ok else 'unavailable (librocm_smi64 not loadable)')
        for key in ('sclk_mhz', 'socket_power_w', 'temp_edge_c', 'temp_junction_c'):
            m, lo, hi = med(key)
            out[key] = m
            if m is not None and key in ('sclk_mhz', 'socket_power_w'):
                out[key + '_min_max'] = [lo, hi]
        return out


def _watch(period: float, dev: int):
    """child side of Sampler: print one JSON reading per period until a line arrives on stdin (or it closes)"""
    import json
    import select
    import sys
    print('ready' if _load() is not None else 'unavailable', flush=True)
    if _load() is None:
        return
    while True:
        print(json.dumps(sample(dev)), flush=True)
        r, _, _ = select.select([sys.stdin], [], [], period)
        if r:
            return


if __name__ == '__main__':   # python -m sevennet_amd.telemetry [--watch PERIOD] [--dev N]: one reading, or the Sampler's child
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--watch', type=float, default=0.0)
    ap.add_argument('--dev', type=int, default=0)
    a = ap.parse_args()
    if a.watch > 0:
        _watch(a.watch, a.dev)
    else:
        t0 = time.perf_counter()
        print(sample(a.dev), f'{(time.perf_counter() - t0) * 1e3:.2f} ms')
