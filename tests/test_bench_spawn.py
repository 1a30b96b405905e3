"""`bench.py --gpus N` must really run N ranks (the flag was parsed and ignored).  The launch plumbing is
exercised on CPU: RIH_BENCH_SPAWN_PROBE=1 makes every rank join a gloo group and exit before it touches a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=240):
    env = dict(os.environ, RIH_BENCH_SPAWN_PROBE='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_flag_spawns_that_many_ranks():
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d == {'spawn_probe': True, 'n_gpus': 2, 'ranks_seen': 2, 'gpus_arg': 2}


def test_default_is_one_rank_without_a_launcher():
    r = _run([])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['ranks_seen'] == 1


def test_world_size_and_gpus_flag_must_agree():
    r = _run(['--gpus', '4'], extra_env={'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'must agree' in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_config5_line_carries_a_roofline():
    """BASELINE configs[4] (`bench.py --config5`): the line names its workload, reports the deviation from the fp32 path and a
    `roofline` object with the matrix-pipe families of one forward (fp16-storage backbone, fp32 decoder, attention)."""
    env = dict(os.environ)
    env.pop('RIH_BENCH_SPAWN_PROBE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config5', '--batch', '16', '--steps', '2', '--warmup', '1'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['unit'] == 'images/sec' and d['value'] > 0 and 'configs[4]' in d['config']['workload']
    assert d['config']['vertices_rel_deviation_from_fp32_path'] < 5e-3
    roof = d['roofline']
    assert roof['bound'] == 'mfma' and 0.0 < roof['frac'] < 1.0 and roof['unit'] == 'TFLOP/s'
    fams = roof['families']
    assert any('rih_hconv' in k for k in fams) and any('rih_gemm' in k for k in fams)
    for v in fams.values():
        assert v['launches'] > 0 and v['ms'] > 0 and 0.0 < v['frac'] < 1.0
