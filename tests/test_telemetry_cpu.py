"""CPU: the bench line's clock / power reader (sevennet_amd/telemetry.py) is best effort -- on a box without a GPU (or without
librocm_smi64) it reports None fields and never raises into the measurement; the visible -> physical device index map."""
import time


def test_sampler_without_a_gpu_reports_none_fields():
    from sevennet_amd.telemetry import Sampler
    with Sampler(0, period_s=0.01) as s:
        time.sleep(0.05)
    out = s.summary()
    for key in ('sclk_mhz', 'socket_power_w', 'temp_edge_c', 'temp_junction_c', 'telemetry_samples', 'telemetry_source'):
        assert key in out
    if out['telemetry_samples'] == 0:
        assert out['sclk_mhz'] is None and out['socket_power_w'] is None


def test_physical_index_follows_the_visibility_mask(monkeypatch):
    from sevennet_amd.telemetry import physical_index
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(var, raising=False)
    assert physical_index(0) == 0 and physical_index(3) == 3
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '4,5,6')
    assert physical_index(0) == 4 and physical_index(2) == 6 and physical_index(7) == 7
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', 'GPU-abcdef')
    assert physical_index(0) == 0
