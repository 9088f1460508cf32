"""The drop-in boundary without a GPU: the C-ABI library builds, loads, and exports every symbol
include/det6d_ops.h declares; the product never touches the oracle; missing pieces fail loudly."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib():
    from de6d_amd import _build
    return _build.build()


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'det6d_ops.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(det6d_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol(hip_lib):
    import torch  # noqa: F401  (loads libamdhip64 first, like the product does)
    lib = ctypes.CDLL(hip_lib)
    names = declared_symbols()
    assert len(names) >= 27
    for name in names:
        assert hasattr(lib, name), "libdet6d_hip.so does not export %s" % name
    from de6d_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == names
    lib.det6d_version.restype = ctypes.c_char_p
    assert lib.det6d_version().startswith(b"det6d-hip gfx950")


def exported_symbols(path):
    out = subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.split() and line.split()[-2] in ('T', 'W', 'D', 'B'))


def test_dynamic_symbol_table_equals_the_header_both_ways(hip_lib):
    """nothing undeclared is exported (no debug hooks, no experiment entries) and nothing declared is missing"""
    ours = [s for s in exported_symbols(hip_lib) if not s.startswith(('__hip', '_fini', '_init', '__bss', '_edata', '_end'))]
    assert ours == declared_symbols(), set(ours) ^ set(declared_symbols())


def test_shipped_library_ignores_experiment_variables(hip_lib):
    """stand-ins / timing hooks / tile sweeps are compiled only with -DDET6D_EXPERIMENTS"""
    out = subprocess.run(['strings', hip_lib], capture_output=True, text=True).stdout
    for name in ('DET6D_FPS_STANDIN', 'DET6D_FPS_DBG', 'DET6D_LINEAR_K64MAX', 'DET6D_LINEAR_BK32', 'DET6D_COMPACT_TOL',
                 'fps_standin_kernel', 'det6d_dbg_fps_clock'):
        assert name not in out, name


def test_code_object_is_gfx950_only(hip_lib):
    out = subprocess.run(['strings', hip_lib], capture_output=True, text=True).stdout
    # the offload bundle's target ids (rocPRIM's host-side architecture NAME table, linked in with the device radix sort
    # of csrc/fps_coop.hip, also contains strings like "gfx90a": those are not code objects)
    targets = set(re.findall(r'amdgcn-amd-amdhsa--(gfx[0-9a-f]+)', out))
    assert targets == {'gfx950'}, targets
    assert 'sm_' not in out and 'nvptx' not in out


def test_oracle_exports_mirror_the_abi(oracle_ops):
    lib = oracle_ops.lib()
    for name in declared_symbols():
        if name in ('det6d_version', 'det6d_last_error', 'det6d_nms_to_host', 'det6d_boxes_iou_bev_cpu', 'det6d_fps_fused_workspace_bytes', 'det6d_fps_fused_status', 'det6d_fps_fused_status_offset', 'det6d_mlp_group3_supported', 'det6d_mlp_rows_supported',
                    'det6d_ball_query_grid_supported'):
            continue
        assert hasattr(lib, name.replace('det6d_', 'det6d_oracle_', 1)), name


def test_product_never_imports_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, 'de6d_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(base, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M) or 'libdet6d_oracle' in src:
                    bad.append(os.path.join(base, f))
    assert not bad, "product files reference the oracle: %s" % bad


def test_ops_fail_loudly_without_device_tensors(hip_lib):
    import torch
    from de6d_amd._lib import Det6dError
    from de6d_amd.ops import pointnet2_batch_hip as pn
    xyz = torch.zeros((1, 8, 3))
    with pytest.raises(Det6dError):
        pn.farthest_point_sampling_wrapper(1, 8, 4, xyz, torch.zeros((1, 8)), torch.zeros((1, 4), dtype=torch.int32))


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from de6d_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.Det6dError):
        _lib.lib()


def test_invalid_arguments_return_status_not_exit(hip_lib):
    import torch  # noqa: F401
    lib = ctypes.CDLL(hip_lib)
    assert lib.det6d_fps(1, 0, 4, None, None, None, None) == -1          # DET6D_EINVAL
    assert lib.det6d_ball_query(1, 8, 4, ctypes.c_float(1.0), 1000, None, None, None, None) == -1
    assert lib.det6d_linear(None, None) == -1
    assert lib.det6d_fps(0, 8, 4, ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8), None) == 0  # b = 0: nothing to do


def test_extension_modules_have_the_reference_names_and_arities():
    """tests/golden/extension_api.json is parsed from the reference's pointnet2_api.cpp:11-30 / iou3d_nms_api.cpp:11-17 and the
    prototypes they bind (make_golden.py: gen_extension_api): every exported name exists in the drop-in modules with the same
    positional arity — including the ones that only raise (grid_query_wrapper, f-fps)"""
    import inspect
    import json
    from de6d_amd.ops import pointnet2_batch_hip, iou3d_nms_hip
    api = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'extension_api.json')))
    for modname, mod in (('pointnet2_batch_cuda', pointnet2_batch_hip), ('iou3d_nms_cuda', iou3d_nms_hip)):
        assert len(api[modname]) >= 5
        for name, spec in api[modname].items():
            fn = getattr(mod, name, None)
            assert callable(fn), "%s.%s is missing" % (modname, name)
            params = [p for p in inspect.signature(fn).parameters.values()
                      if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) and p.default is p.empty]
            assert len(params) == spec['arity'], "%s.%s takes %d positional arguments, the reference's %s takes %d" % (
                modname, name, len(params), spec['binds'], spec['arity'])
    with pytest.raises(NotImplementedError):
        pointnet2_batch_hip.grid_query_wrapper(*[None] * 11)
    with pytest.raises(NotImplementedError):
        pointnet2_batch_hip.furthest_point_sampling_matrix_wrapper(*[None] * 6)


def test_install_pcdet_ops_serves_unmodified_reference_import_lines(tmp_path, monkeypatch):
    """`from . import pointnet2_batch_cuda as pointnet2` (pointnet2_utils.py:7) and `from . import iou3d_nms_cuda`
    (iou3d_nms_utils.py:9), written exactly as the reference writes them inside a package tree of the reference's shape,
    resolve to the HIP-backed modules once de6d_amd.install_pcdet_ops() has run — no edit of the importing file"""
    import importlib
    import json
    import de6d_amd
    api = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'extension_api.json')))
    assert set(api['import_sites'].values()) == set(de6d_amd.PCDET_EXTENSION_SITES)
    pkg = tmp_path / 'pcdet'
    for sub in ('', 'ops', 'ops/pointnet2', 'ops/pointnet2/pointnet2_batch', 'ops/iou3d_nms'):
        (pkg / sub).mkdir(parents=True, exist_ok=True)
        (pkg / sub / '__init__.py').write_text('')
    (pkg / 'ops/pointnet2/pointnet2_batch/pointnet2_utils.py').write_text('from . import pointnet2_batch_cuda as pointnet2\n')
    (pkg / 'ops/iou3d_nms/iou3d_nms_utils.py').write_text('from . import iou3d_nms_cuda\n')
    monkeypatch.syspath_prepend(str(tmp_path))
    for name in [n for n in sys.modules if n == 'pcdet' or n.startswith('pcdet.')]:
        monkeypatch.delitem(sys.modules, name)
    for site in de6d_amd.PCDET_EXTENSION_SITES:
        monkeypatch.delitem(sys.modules, site, raising=False)
    pn, iou = de6d_amd.install_pcdet_ops()
    a = importlib.import_module('pcdet.ops.pointnet2.pointnet2_batch.pointnet2_utils')
    b = importlib.import_module('pcdet.ops.iou3d_nms.iou3d_nms_utils')
    assert a.pointnet2 is pn and b.iou3d_nms_cuda is iou
    assert pn.__name__ == 'de6d_amd.ops.pointnet2_batch_hip' and callable(a.pointnet2.ball_query_dilated_wrapper)
    for name in [n for n in sys.modules if n == 'pcdet' or n.startswith('pcdet.')]:
        monkeypatch.delitem(sys.modules, name)


@pytest.mark.skipif(not os.path.isdir('/root/reference/core/pcdet'), reason="authoring container only: needs the reference checkout")
def test_reference_op_wrappers_import_unmodified_over_the_hip_modules():
    """the reference's OWN pointnet2_utils.py and iou3d_nms_utils.py, imported as they lie (a child process with the reference
    first on sys.path, its unrelated third-party imports stubbed), bind to de6d_amd's modules after install_pcdet_ops()"""
    code = r"""
import sys, types
sys.path.insert(0, '/root/reference/core'); sys.path.insert(1, %r)
for name in ('SharedArray', 'easydict', 'numba', 'numba.cuda', 'skimage', 'skimage.io', 'skimage.transform', 'spconv', 'spconv.pytorch'):
    sys.modules[name] = types.ModuleType(name)
sys.modules['easydict'].EasyDict = dict
v = types.ModuleType('pcdet.version'); v.__version__ = 'ref'; sys.modules['pcdet.version'] = v
import de6d_amd
pn, iou = de6d_amd.install_pcdet_ops()
from pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils
from pcdet.ops.iou3d_nms import iou3d_nms_utils
assert pointnet2_utils.__file__.startswith('/root/reference/'), pointnet2_utils.__file__
assert pointnet2_utils.pointnet2 is pn and iou3d_nms_utils.iou3d_nms_cuda is iou
print('bound', pn.__name__, iou.__name__)
""" % ROOT
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'bound de6d_amd.ops.pointnet2_batch_hip de6d_amd.ops.iou3d_nms_hip' in out.stdout, out.stderr[-3000:]
