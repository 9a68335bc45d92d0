"""bench.py's output contract (what the round driver parses): ONE JSON line on rank 0 with the headline metric, the
roofline block of the dominant kernel measured live with HIP events, the CPU baseline and the parity check at the timed
size.  Runs bench.py as a child process on the GPU box (short: 5 timed steps, 2 s of CPU baseline, one aux entry)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_contract(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '5', '--warmup', '2',
                          '--cpu-seconds', '2', '--no-aux'], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'parity_checked', 'max_rel_err'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 5 and d['warmup'] == 2 and d['higher_is_better'] is True
    assert d['dtype'] == 'f64' and d['scaling'] == 'strong' and d['vs_baseline'] is None
    assert 'workload' in d['config'] and abs(d['config']['Nlambda_per_gpu'] - 10240) <= 102
    assert d['config']['sweep'] in ('march', 'lanes')
    assert d['value'] == pytest.approx(1e3 / d['ms_per_step'], rel=1e-9)
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and ('raymarch_kernel' in r['kernel'] or 'lanesweep_kernel' in r['kernel'])
    assert r['launches_timed'] >= 1 and 0.0 < r['kernel_ms'] < d['ms_per_step']
    assert r['achieved'] == pytest.approx(r['algorithmic_bytes_per_launch'] / (r['kernel_ms'] * 1e-3) / 1e9, rel=1e-9)
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'], rel=1e-12)
    c = d['cpu_baseline']
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == 'iters/s'
    assert d['parity_checked'] is True and d['max_rel_err'] <= 1e-8
