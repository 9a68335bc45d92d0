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
    # the in-run parity check is the one place where the REFERENCE ITSELF (oracle/_ref: the real core, scalar scheme) runs at the
    # timed size inside the GPU suite (the other timed-size tests compare with the C oracle, which is pinned to it bit for bit
    # on the small fixtures): one call 1e-9 element-wise, populations after stat_equil 1e-8
    par = d['parity']
    if c['kind'] == 'reference':
        assert 'scalar' in par['against']
    for k in ('J', 'I', 'Gamma', 'Rij', 'Rji'):
        assert par['rel_err'][k] <= 1e-9, (k, par['rel_err'][k])
    assert par['rel_err']['n'] <= 1e-8


def test_rank_launch_command_shape():
    """`python bench.py --gpus N` without a launcher starts its ranks itself: the command it spawns."""
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.rank_launch_command(4, ['--gpus', '4', '--steps', '7'], port=29511)
    assert cmd[1:3] == ['-m', 'torch.distributed.run']
    assert '--nproc-per-node=4' in cmd and '--nnodes=1' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29511'
    assert cmd[-5].endswith('bench.py') and cmd[-4:] == ['--gpus', '4', '--steps', '7']


def test_self_launch_spawns_ranks_without_a_gpu():
    """No launcher, no GPU: the parent spawns two ranks (it must not touch the GPU itself), each rank refuses to run
    without a device, and the parent passes the failure on as its own exit code."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU box: covered by test_bench_self_launch_two_ranks_on_one_gpu')
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--share-gpu',
                          '--steps', '1', '--warmup', '0'], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode != 0
    # (the launcher ends the other rank as soon as one has failed: on a loaded machine only the first may get to say so)
    assert out.stderr.count('bench.py needs a GPU') >= 1, out.stderr[-2000:]


@pytest.mark.gpu
def test_bench_self_launch_two_ranks_on_one_gpu(gpu):
    """`python3 bench.py --gpus 2` with no launcher and no WORLD_SIZE: the form the driver's SCALE command may take.  Two
    ranks share the box's one GPU under gloo (functional: the sharded path, one collective per step, rank 0's line)."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--share-gpu',
                          '--steps', '5', '--warmup', '2', '--no-weak'], capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 5 and d['scaling'] == 'strong'
    assert d['value'] == pytest.approx(1e3 / d['ms_per_step'], rel=1e-9)
    assert abs(d['config']['Nlambda_total'] - 10240) <= 102 and d['config']['Nlambda_per_gpu'] < 6000
    assert 'allreduce_us' in d


def test_column_sets_of_the_ranks_partition_the_seeded_set():
    """--mode columns: column c of BASELINE configs[3] is FAL-C perturbed with seed 1234 + c whatever the number of ranks; the
    ranks' round-robin shares are disjoint, cover the set, and a rank's problems are the ones a single rank builds for the
    same column numbers (no rank-dependent input)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from lightweaver_amd.harness import models
    whole_idx, whole = bench.column_problems(models, 6, 1, 0, lineScale=0.2)
    assert whole_idx == list(range(6))
    seen = []
    for r in range(3):
        idx, probs = bench.column_problems(models, 6, 3, r, lineScale=0.2)
        assert idx == [r, r + 3]
        for c, p in zip(idx, probs):
            assert np.array_equal(p.temperature, whole[c].temperature) and np.array_equal(p.vlosMu, whole[c].vlosMu)
            assert np.array_equal(p.atoms[1].n, whole[c].atoms[1].n)
        seen += idx
    assert sorted(seen) == list(range(6))
    assert not np.array_equal(whole[0].temperature, whole[1].temperature)


@pytest.mark.gpu
def test_bench_columns_mode_two_ranks_on_one_gpu(gpu):
    """`python3 bench.py --gpus 2 --mode columns` (BASELINE configs[3]'s split, two ranks sharing the box's one GPU under gloo):
    every rank runs ITS columns as a fused batch, no collective in the data path; rank 0's line counts the column-iterations
    of both ranks against the slower rank's clock."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--share-gpu',
                          '--mode', 'columns', '--columns', '6', '--steps', '4', '--warmup', '1'], capture_output=True, text=True,
                         timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 4 and d['scaling'] == 'weak' and d['unit'] == 'column-iters/s'
    assert d['config']['columns_total'] == 12 and d['config']['columns_per_gpu'] == 6 and d['config']['fused'] is True
    assert d['value'] == pytest.approx(12 * 1e3 / d['ms_per_step'], rel=1e-9)
    assert len(d['per_rank_ms_per_step']) == 2 and all(0.0 < x <= d['ms_per_step'] * 1.0001 for x in d['per_rank_ms_per_step'])
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['frac'] == pytest.approx(r['achieved'] / 8000.0, rel=1e-12) and 0.0 < r['frac'] < 1.0
    assert d['config']['final_dJMax_max'] > 0.0   # (the largest relative change of J over the columns: of order 10 after five iterations from LTE)
