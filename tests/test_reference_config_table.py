"""Every named configuration of the reference against this repository's mirror, attribute by attribute.

tests/golden/refshim_config_table.json is what the REFERENCE'S OWN python/opt_configs.py, variables.py, util.py and configs.py
resolve to (tools/make_reference_fixtures.py --shim --config-table: their files imported from the checkout on the torch stand-in for
Mitsuba / Dr.Jit): all opt-configs that need no scene file -- 82 of the 85 -- with their sensors (origins of get_regular_cameras /
get_regular_cameras_top, python/util.py:84-143), film sizes, losses, batch sizes, upsampling schedules and the Variable objects
(class, key, shape, learning rate, schedule, regulariser and weight, EMA beta, box constraint and the box SDF itself,
python/variables.py:79-190), and the method configs with the warp field they build (python/configs.py:12-125).  The mirror
(differentiable-sdf-rendering_amd/python) must resolve every one of them to the same values."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import configs
import opt_configs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from make_reference_fixtures import method_config_record, opt_config_record          # noqa: E402  (the generator's own dumper)

TABLE = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'refshim_config_table.json')))
REFERENCE = os.environ.get('DSDF_REFERENCE', '/root/reference')


def same(a, b, path=''):
    """Recursive comparison; floats to 1e-6 relative (the box SDF statistics and sensor origins are computed, not copied)."""
    if isinstance(a, dict):
        assert isinstance(b, dict) and set(a) == set(b), (path, sorted(set(a) ^ set(b)))
        for k in a:
            same(a[k], b[k], f'{path}.{k}')
    elif isinstance(a, list):
        assert isinstance(b, list) and len(a) == len(b), (path, a, b)
        for i, (x, y) in enumerate(zip(a, b)):
            same(x, y, f'{path}[{i}]')
    elif isinstance(a, float) or isinstance(b, float):
        assert a is not None and b is not None and abs(a - b) <= 1e-6 * max(1.0, abs(a), abs(b)), (path, a, b)
    else:
        assert a == b, (path, a, b)


def test_table_covers_the_reference():
    assert len(TABLE['opt']) == 82 and sorted(TABLE['skipped']) == ['mirror-opt-1', 'mirror-opt-hq', 'torus-shadow-1']
    assert set(TABLE['opt']) | set(TABLE['skipped']) == set(opt_configs.SCENE_CONFIGS)


@pytest.mark.parametrize('name', sorted(TABLE['opt']))
def test_opt_config_matches_reference_code(name):
    got = opt_config_record(opt_configs.get_opt_config(name))
    same(TABLE['opt'][name], json.loads(json.dumps(got)), name)


@pytest.mark.parametrize('name', sorted(TABLE['method']))
def test_method_config_matches_reference_code(name):
    got = json.loads(json.dumps(method_config_record(configs.get_config(name))))
    ref = TABLE['method'][name]
    if ref['warpfield'] is not None and isinstance(ref['warpfield'].get('edge_eps'), list):
        ref = dict(ref, warpfield=dict(ref['warpfield'], edge_eps=ref['warpfield']['edge_eps'][0]))   # (dr.opaque(mi.Float, ...): one lane)
    same(ref, got, name)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, 'python')), reason="the reference checkout is not on this machine (GPU box)")
def test_table_is_what_the_reference_code_produces(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'make_reference_fixtures.py'), '--shim', '--config-table', '--reference', REFERENCE,
                        '--out', str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:]
    same(TABLE, json.load(open(tmp_path / 'refshim_config_table.json')), 'table')
