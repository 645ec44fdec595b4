"""RCCL on the MI355X the only way a 1-GPU box allows: bench.py under torch.distributed.run with ONE
rank and BENCH_FORCE_DIST=1 -- process-group initialisation on the `nccl` (= RCCL) backend, the non-blocking all-reduce of
dL/dsdf on the collective stream, barrier and the max-over-ranks timing all run for real; the summed gradient of the last
step must equal the single-process run's (same seeds, same kernels) to the order of the float atomics."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ['--gpus', '1', '--steps', '3', '--warmup', '1', '--res', '128', '--img', '256', '--no-cpu-baseline', '--no-low-spp',
        '--no-direct', '--no-opt-iteration']


def _line(out):
    rows = [l for l in out.splitlines() if l.startswith('{') and '"metric"' in l]
    assert rows, out[-2000:]
    return json.loads(rows[-1])


def test_bench_single_rank_rccl_matches_plain_run(built):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    plain = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + ARGS, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=600, cwd=ROOT)
    assert plain.returncode == 0, plain.stderr[-2000:]
    a = _line(plain.stdout)
    port = 29400 + os.getpid() % 500
    for extra in ({}, {'BENCH_FORCE_TILED': '1'}):              # whole views, then the film-level (pixel-window) path
        denv = dict(env, BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', **extra)
        run = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                              '--master-port', str(port), os.path.join(ROOT, 'bench.py')] + ARGS, env=denv, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
        assert run.returncode == 0, run.stderr[-3000:]
        b = _line(run.stdout)
        assert b['config']['dist_backend'] == 'nccl' and b['n_gpus'] == 1
        ga, gb = a['config']['grad_l1_last_step'], b['config']['grad_l1_last_step']
        assert ga > 0 and abs(ga - gb) <= 1e-4 * ga, (ga, gb, extra)
        port += 1


def test_bench_driver_command_full_size_all_blocks(built):
    """the driver's command at full size (256^3, 12 x 512^2, 256/64 spp) with EVERY side block on -- only the
    step counts are reduced.  The run must end by itself well inside the driver's patience, print the headline first (a line with
    roofline but without the side blocks) and end with a line that carries every block."""
    import time
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    t0 = time.time()
    run = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1'], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT)
    wall = time.time() - t0
    assert run.returncode == 0, run.stderr[-3000:]
    rows = [json.loads(l) for l in run.stdout.splitlines() if l.startswith('{') and '"metric"' in l]
    assert len(rows) >= 6, run.stdout[-2000:]
    first, last = rows[0], rows[-1]
    assert first['value'] > 0 and first['roofline'] and first['roofline']['frac'] and 'cpu_baseline' not in first
    assert last['value'] == first['value'] and 'aborted' not in last
    for block in ('cpu_baseline', 'low_spp', 'direct', 'opt_iteration', 'scaling_prediction'):
        assert block in last and 'error' not in last[block] and 'skipped' not in last[block], (block, last.get(block))
    assert last['cpu_baseline']['value'] > 0 and last['cpu_baseline']['cores'] >= 1
    assert wall < 200, wall
