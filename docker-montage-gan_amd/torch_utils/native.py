"""ctypes binding of libmontage_hip.so (C ABI: include/montage_hip.h).

No CPU fallback lives here: if the shared library is missing or fails to load,
``lib()`` raises, and so does every op that was handed a GPU tensor.
"""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# csrc/ sits next to torch_utils/ (both in this repo and when dropped into the reference's
# montage_gan/ directory); MONTAGE_HIP_LIB overrides the location of the built library.
CSRC = os.path.join(os.path.dirname(_HERE), 'csrc')
LIB_PATH = os.environ.get('MONTAGE_HIP_LIB', os.path.join(CSRC, 'libmontage_hip.so'))

MG_F32, MG_F16, MG_BF16, MG_F64 = 0, 1, 2, 3
K_UPFIRDN2D_GENERIC, K_UPFIRDN2D_TILED, K_BIAS_ACT, K_FMA_BIAS_ACT = 0, 1, 2, 3
K_ALPHA_COMPOSITE_FWD, K_ALPHA_COMPOSITE_BWD, K_TRANSLATE_FWD, K_TRANSLATE_BWD_INPUT = 4, 5, 6, 7
K_TRANSLATE_BWD_SHIFT, K_FILTERED_LRELU, K_BWD_FUSED, K_UPFIRDN2D_STREAM = 8, 9, 10, 11
K_UPFIRDN2D_NHWC = 12
K_AFFINE_WARP_FWD, K_AFFINE_WARP_BWD, K_CHANNEL_MIX = 13, 14, 15
K_DEMOD = 16
K_STATS = 18
K_ADA_PARAMS = 19
K_MODCONV = 20
K_UPFIRDN2D_SEP = 21
K_MBSTD = 22
K_TORGB = 23
K_WGRAD = 24
K_FROMRGB = 25
K_RESIDUAL = 26
GS_SLOTS = 64       # MG_GS_SLOTS
KERNEL_NAMES = {0: 'upfirdn2d_generic', 1: 'upfirdn2d_tiled', 2: 'bias_act', 3: 'fma_bias_act', 4: 'alpha_composite_fwd',
                5: 'alpha_composite_bwd', 6: 'translate_fwd', 7: 'translate_bwd_input', 8: 'translate_bwd_shift',
                9: 'filtered_lrelu', 10: 'bwd_fused', 11: 'upfirdn2d_stream', 12: 'upfirdn2d_nhwc',
                13: 'affine_warp_fwd', 14: 'affine_warp_bwd', 15: 'channel_mix', 16: 'demod', 18: 'stats', 19: 'ada_params', 20: 'modconv', 21: 'upfirdn2d_sep', 22: 'mbstd', 23: 'torgb', 24: 'wgrad', 25: 'fromrgb', 26: 'residual'}

_lib = None
_lock = threading.RLock()

c_int, c_i32, c_i64, c_float, c_void_p = ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
I32x4, I64x4, I32x2, I64x2 = c_i32 * 4, c_i64 * 4, c_i32 * 2, c_i64 * 2


def build(verbose=False):
    """Compile every HIP source for gfx950 into csrc/libmontage_hip.so (hipcc cross-compiles without a GPU) and the
    torch.utils.cpp_extension shim on top of it (csrc/torch_plugin.cpp -> csrc/montage_torch_plugin*.so)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(['make', '-C', CSRC, '-j', str(min(8, os.cpu_count() or 1))], check=True, stdout=out)
    build_torch_plugin(verbose=verbose)
    return LIB_PATH


TORCH_PLUGIN_NAME = 'montage_torch_plugin'


def torch_plugin_path():
    import sysconfig
    return os.path.join(CSRC, TORCH_PLUGIN_NAME + sysconfig.get_config_var('EXT_SUFFIX'))


def build_torch_plugin(sources=None, verbose=False, force=False, **build_kwargs):
    """The pybind11 / ATen shim (host C++ only: tensors in, C ABI out).  Ahead of time and in-tree -- the same compiler
    invocation ``torch.utils.cpp_extension`` would issue for a ``.cpp``-only extension (its include paths, its ABI and ROCm
    defines), without the per-process JIT cache, so that the built module travels with the tree.  `sources` defaults to
    csrc/torch_plugin.cpp; `build_kwargs` understands ``extra_cflags`` / ``extra_ldflags`` / ``extra_include_paths``."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    srcs = [os.path.join(CSRC, 'torch_plugin.cpp')] if not sources else list(sources)
    out = torch_plugin_path()
    deps = srcs + [os.path.join(os.path.dirname(os.path.dirname(CSRC)), 'include', 'montage_hip.h'), LIB_PATH]
    if not force and os.path.exists(out) and all(os.path.exists(d) and os.path.getmtime(d) <= os.path.getmtime(out) for d in deps[:-1]):
        return out
    libdir = os.path.join(os.path.dirname(torch.__file__), 'lib')
    cmd = [os.environ.get('CXX', 'g++'), '-O2', '-fPIC', '-shared', '-std=c++17', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
           f'-DTORCH_EXTENSION_NAME={TORCH_PLUGIN_NAME}', '-DTORCH_API_INCLUDE_EXTENSION_H',
           f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}'] + list(build_kwargs.get('extra_cflags', []))
    for inc in ce.include_paths() + [os.path.join(ce.ROCM_HOME or '/opt/rocm', 'include'), sysconfig.get_paths()['include']] + \
            list(build_kwargs.get('extra_include_paths', [])):
        cmd += ['-isystem', inc]
    cmd += srcs + ['-o', out, '-L' + libdir, '-lc10', '-lc10_hip', '-ltorch', '-ltorch_cpu', '-ltorch_hip', '-ltorch_python',
                   '-L' + CSRC, '-lmontage_hip', '-Wl,-rpath,$ORIGIN', '-Wl,-rpath,' + libdir] + list(build_kwargs.get('extra_ldflags', []))
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return out


_torch_plugin = None


def torch_plugin():
    """The loaded pybind module (csrc/montage_torch_plugin*.so).  Raises when it has not been built."""
    global _torch_plugin
    if _torch_plugin is None:
        with _lock:
            if _torch_plugin is None:
                import importlib.util
                import torch  # noqa: F401  (libtorch must be loaded before the extension)
                path = torch_plugin_path()
                if not os.path.exists(path):
                    raise RuntimeError(f'{path} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"`')
                lib()           # libmontage_hip.so first: same search path for the extension's DT_NEEDED entry
                spec = importlib.util.spec_from_file_location(TORCH_PLUGIN_NAME, path)
                m = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(m)
                _torch_plugin = m
    return _torch_plugin


def _declare(L):
    L.mg_abi_version.restype = c_int
    L.mg_last_error.restype = ctypes.c_char_p
    L.mg_launch_count.restype = ctypes.c_uint64
    L.mg_launch_count.argtypes = [c_int]
    L.mg_prof_enable.argtypes = [c_int]
    L.mg_prof_collect.argtypes = [c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double),
                                  ctypes.POINTER(ctypes.c_double)]
    L.mg_upfirdn2d.argtypes = [c_void_p, c_void_p, c_void_p, c_int, I32x4, I64x4, I32x2, I64x2, I32x4, I64x4,
                               c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]
    L.mg_upfirdn2d_epilogue.argtypes = [c_void_p, c_void_p, c_void_p, c_int, I32x4, I32x2, I64x2, I32x4, c_int, c_int, c_int, c_float,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_float, c_float, c_float, c_void_p]
    L.mg_upfirdn2d_sep12.argtypes = [c_void_p, c_void_p, c_void_p, c_i64] + [c_i32] * 8 + [c_float, c_void_p]
    L.mg_reflect_pad_upsample12.argtypes = [c_void_p, c_void_p, c_void_p, c_i64] + [c_i32] * 6 + [c_float, c_void_p]
    L.mg_reflect_pad_upsample12_window.argtypes = [c_void_p, c_void_p, c_void_p, c_i64] + [c_i32] * 6 + [c_float, c_void_p, c_void_p]
    L.mg_mbstd_fwd.argtypes = [c_void_p, c_void_p] + [c_i32] * 5 + [c_void_p]
    L.mg_mbstd_bwd.argtypes = [c_void_p, c_void_p, c_void_p] + [c_i32] * 5 + [c_void_p]
    L.mg_demod_fwd.argtypes = [c_void_p] * 4 + [c_i32] * 5 + [c_void_p]
    L.mg_demod_bwd.argtypes = [c_void_p] * 7 + [c_i32] * 5 + [c_void_p]
    L.mg_modconv3x3_c64.argtypes = [c_void_p] * 10 + [c_i32, c_i32, c_i32, c_i64, c_int, c_float, c_float, c_float, c_void_p]
    L.mg_wgrad3x3_c64.argtypes = [c_void_p] * 6 + [c_i32, c_i32, c_i32, c_void_p]
    L.mg_residual_bias_act_fwd.argtypes = [c_void_p] * 4 + [c_int, c_int, c_float, c_float, c_float, c_float, c_i64, c_i32, c_void_p]
    L.mg_residual_bias_act_bwd.argtypes = [c_void_p] * 6 + [c_int, c_int, c_float, c_float, c_float, c_float, c_i64, c_i32, c_void_p]
    L.mg_fromrgb_nhwc_fwd.argtypes = [c_void_p] * 4 + [c_int, c_int, c_float, c_float, c_float, c_i64, c_i32, c_i32, c_void_p]
    L.mg_fromrgb_nhwc_bwd.argtypes = [c_void_p] * 9 + [c_int, c_int, c_float, c_float, c_float, c_i64, c_i32, c_i32, c_void_p]
    L.mg_torgb_nhwc_fwd.argtypes = [c_void_p] * 5 + [c_int, c_i32, c_i32, c_i32, c_i32, c_float, c_void_p]
    L.mg_torgb_nhwc_bwd.argtypes = [c_void_p] * 7 + [c_int, c_i32, c_i32, c_i32, c_i32, c_void_p]
    L.mg_torgb_bwd_finish.argtypes = [c_void_p] * 7 + [c_int, c_i32, c_i32, c_i32, c_void_p]
    L.mg_ada_params.argtypes = [c_void_p, c_void_p, c_void_p, c_i32, c_i32, c_i32, ctypes.POINTER(c_float), c_i32] + [c_void_p] * 7
    L.mg_stats_accumulate.argtypes = [c_void_p, c_int, c_i64, c_void_p, c_void_p]
    L.mg_upfirdn2d_index_table.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    L.mg_bias_act.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                              c_float, c_float, c_float, c_i64, c_i32, c_i64, c_void_p]
    L.mg_fma_bias_act.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float,
                                  c_float, c_float, c_void_p, c_i64, c_i32, c_i32, c_i32, c_void_p]
    L.mg_bwd_fused.argtypes = [c_void_p] * 9 + [c_int, c_int, c_float, c_float, c_float, c_i64, c_i32, c_i32, c_i32, c_void_p]
    L.mg_fma_bias_act_nhwc.argtypes = L.mg_fma_bias_act.argtypes
    L.mg_bwd_fused_nhwc_y.argtypes = [c_void_p] * 7 + [c_int] + [c_void_p] * 4 + [c_int, c_int, c_float, c_float, c_float, c_i64, c_i32, c_i32, c_i32, c_void_p]
    L.mg_affine_warp_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_void_p]
    L.mg_affine_warp_bwd.argtypes = L.mg_affine_warp_fwd.argtypes
    L.mg_channel_mix.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i32, c_i32, c_i64, c_void_p]
    L.mg_bwd_fused_nhwc.argtypes = L.mg_bwd_fused.argtypes
    L.mg_alpha_composite_fwd.argtypes = [c_void_p, c_void_p, c_int, c_i32, c_i32, c_i32, c_i32, c_void_p]
    L.mg_alpha_composite_bwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_i32, c_i32, c_i32, c_i32, c_int, c_void_p]
    L.mg_translate_sample_fwd.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_i32, c_i32, c_i32, c_i32, c_float, c_void_p]
    L.mg_translate_sample_bwd_input.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_i32, c_i32, c_i32, c_i32, c_void_p]
    L.mg_translate_sample_bwd_shift.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i32, c_i32, c_i32, c_i32,
                                                c_float, c_void_p]
    L.mg_filtered_lrelu.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int] + [c_i32] * 10 + \
                                   [c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_int, c_void_p]
    for name in ('mg_upfirdn2d', 'mg_upfirdn2d_index_table', 'mg_bias_act', 'mg_fma_bias_act',
                 'mg_alpha_composite_fwd', 'mg_alpha_composite_bwd', 'mg_translate_sample_fwd',
                 'mg_translate_sample_bwd_input', 'mg_translate_sample_bwd_shift', 'mg_filtered_lrelu', 'mg_bwd_fused',
                 'mg_fma_bias_act_nhwc', 'mg_bwd_fused_nhwc', 'mg_affine_warp_fwd', 'mg_affine_warp_bwd', 'mg_channel_mix', 'mg_demod_fwd', 'mg_demod_bwd', 'mg_stats_accumulate', 'mg_ada_params', 'mg_modconv3x3_c64', 'mg_upfirdn2d_sep12', 'mg_reflect_pad_upsample12', 'mg_reflect_pad_upsample12_window', 'mg_mbstd_fwd', 'mg_mbstd_bwd', 'mg_torgb_nhwc_fwd', 'mg_torgb_nhwc_bwd', 'mg_torgb_bwd_finish', 'mg_bwd_fused_nhwc_y', 'mg_upfirdn2d_epilogue', 'mg_wgrad3x3_c64', 'mg_fromrgb_nhwc_fwd', 'mg_fromrgb_nhwc_bwd', 'mg_residual_bias_act_fwd', 'mg_residual_bias_act_bwd',
                 'mg_prof_enable', 'mg_prof_collect', 'mg_prof_reset'):
        getattr(L, name).restype = c_int


def lib():
    """The loaded library.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                        f'(or `make -C {CSRC}`).  There is no CPU fallback for GPU tensors.')
                L = ctypes.CDLL(LIB_PATH)
                _declare(L)
                _lib = L
    return _lib


def check(rc):
    """Turn a negative MG_E_* into the RuntimeError a TORCH_CHECK would have raised."""
    if rc != 0:
        raise RuntimeError(lib().mg_last_error().decode('utf-8', 'replace'))


def launch_count(kernel_id):
    return int(lib().mg_launch_count(kernel_id))


def launch_counts():
    return {name: launch_count(k) for k, name in KERNEL_NAMES.items()}


def prof_enable(on=True):
    lib().mg_prof_enable(1 if on else 0)


def prof_reset():
    lib().mg_prof_reset()


def prof_collect():
    """{kernel name: dict(launches, ms, bytes)} for kernels launched while profiling was on."""
    out = {}
    for k, name in KERNEL_NAMES.items():
        n, ms, by = ctypes.c_uint64(0), ctypes.c_double(0), ctypes.c_double(0)
        check(lib().mg_prof_collect(k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)))
        if n.value:
            out[name] = dict(launches=int(n.value), ms=float(ms.value), bytes=float(by.value))
    return out


def dtype_code(torch_dtype):
    import torch
    table = {torch.float32: MG_F32, torch.float16: MG_F16, torch.bfloat16: MG_BF16, torch.float64: MG_F64}
    if torch_dtype not in table:
        raise RuntimeError(f'unsupported dtype {torch_dtype}')
    return table[torch_dtype]


_torch = None


def _t():
    """torch, imported on first use (keeps `import native` light for the ABI-only tests)."""
    global _torch
    if _torch is None:
        import torch
        _torch = torch
    return _torch


def stream_ptr(device):
    """Handle (int) of torch's CURRENT stream on `device` (the contract of the reference plugins, upfirdn2d.cpp:92)."""
    torch = _t()
    idx = device.index
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_no_guard = _NoGuard()


def guard(device):
    """``torch.cuda.device(device)`` only when `device` is not already the current one (the usual case: one process
    per GPU) -- entering the real guard costs more host time than the launch it protects."""
    torch = _t()
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _no_guard
    return torch.cuda.device(device)


def ptr(t):
    """Device address as a plain int (None -> NULL); ctypes converts it for the c_void_p parameters."""
    return t.data_ptr() if t is not None else None
