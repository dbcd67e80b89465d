"""No-GPU checks of the C-ABI boundary: the library builds for gfx950, loads, and exports
every symbol include/montage_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

from conftest import ROOT, load_pkg


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'montage_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mg_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    pkg = load_pkg()
    native = pkg.native
    if not os.path.exists(native.LIB_PATH):
        native.build()
    lib = ctypes.CDLL(native.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 17, names
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/montage_hip.h but not exported'
    lib.mg_abi_version.restype = ctypes.c_int
    assert lib.mg_abi_version() == 1


def test_argument_errors_surface_as_runtime_errors():
    """Argument validation happens on the host before any launch, so it is testable without a GPU."""
    import pytest
    pkg = load_pkg()
    native = pkg.native
    L = native.lib()
    rc = L.mg_bias_act(None, None, None, None, None, None, 0, 0, 1, 0.0, 1.0, -1.0, 16, 0, 1, None)
    assert rc < 0
    with pytest.raises(RuntimeError, match='x must reside on the device'):
        native.check(rc)
    z = native.I32x4(1, 1, 4, 4)
    s = native.I64x4(16, 16, 4, 1)
    rc = L.mg_upfirdn2d(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), 0, z, s, native.I32x2(1, 1),
                        native.I64x2(1, 1), z, s, 0, 1, 1, 1, 0, 0, 0, 1.0, None)
    assert rc < 0
    with pytest.raises(RuntimeError, match='upsampling factor must be at least 1'):
        native.check(rc)


def test_gpu_tensor_without_library_raises(monkeypatch):
    """The product path must fail loudly when the HIP library is missing -- no silent fallback."""
    import pytest
    pkg = load_pkg()
    native = pkg.native
    monkeypatch.setattr(native, '_lib', None)
    monkeypatch.setattr(native, 'LIB_PATH', '/nonexistent/libmontage_hip.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        native.lib()


def test_torch_plugin_shim_builds_loads_and_validates_like_the_reference():
    """csrc/torch_plugin.cpp -- the torch.utils.cpp_extension (pybind11 / ATen) shim over the C ABI: builds in-tree, exposes the
    reference's two plugin functions through custom_ops.get_plugin, and raises the reference's TORCH_CHECK messages
    (upfirdn2d.cpp:19-28, bias_act.cpp:35-51) before anything touches a device."""
    import importlib
    import pytest
    import torch
    pkg = load_pkg()
    native = pkg.native
    native.build_torch_plugin()
    assert os.path.exists(native.torch_plugin_path()) and native.torch_plugin().abi_version() == 1
    co = importlib.import_module('docker-montage-gan_amd.torch_utils.custom_ops')
    co._cached_plugins.clear()
    old = co.verbosity
    co.verbosity = 'none'
    try:
        up = co.get_plugin('upfirdn2d_plugin', sources=['upfirdn2d.cpp', 'upfirdn2d.cu'], extra_cuda_cflags=['--use_fast_math'])
        ba = co.get_plugin('bias_act_plugin', sources=['bias_act.cpp', 'bias_act.cu'], extra_cuda_cflags=['--use_fast_math'])
        assert co.get_plugin('upfirdn2d_plugin') is up
        with pytest.raises(RuntimeError, match='unknown plugin'):
            co.get_plugin('nonexistent_plugin')
    finally:
        co.verbosity = old
    assert type(up).__name__ == '_TorchPlugin' and 'built-in' in repr(up.upfirdn2d)         # the pybind functions, not the ctypes classes
    null = torch.empty([0])
    with pytest.raises(RuntimeError, match='x must reside on CUDA device'):
        up.upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(2, 2), 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)
    with pytest.raises(RuntimeError, match='x must reside on CUDA device'):
        ba.bias_act(torch.zeros(2, 3), torch.zeros(3), null, null, null, 0, 1, 1, 0.0, 1.0, -1.0)
    with pytest.raises(TypeError):
        up.upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(2, 2))                               # the reference's 12-argument signature
    co._cached_plugins.clear()
