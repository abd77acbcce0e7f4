"""ctypes binding of libdctfp.so (include/dctfp.h).  No fallback: if the HIP library is
missing or no MI355X is visible, everything here raises."""

from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

LIB_PATH = os.environ.get('DCTFP_LIBRARY') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libdctfp.so')
RECCUT_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libreccut.so')
#: the same library with the engineering knobs and test hooks of dctfp_set_option compiled in (-DDCTFP_EXPERIMENTS): what
#: tools/ and the kernel-variant / cache tests load; the product (`LIB_PATH`) knows only the options of include/dctfp.h
EXPERIMENTS_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libdctfp_experiments.so')

DCTFP_OK = 0
DCTFP_ERR_INVALID = -1
DCTFP_ERR_SHAPE = -2
DCTFP_ERR_HIP = -3
DCTFP_ERR_NOMEM = -4
DCTFP_ERR_LIMIT = -5
DCTFP_ERR_UNSUPPORTED = -6
DCTFP_MAX_N = 8
DCTFP_MAX_M = 128
DCTFP_F32 = 0
DCTFP_F64 = 1
DCTFP_F16 = 2
DCTFP_BF16 = 3

#: numpy image of ``dctfp_piece`` (include/dctfp.h)
PIECE_DTYPE = np.dtype([('row_start', '<i8'), ('n_rows', '<i4'), ('domain', '<i4'),
                        ('seq', '<i4'), ('reserved', '<i4')], align=True)
assert PIECE_DTYPE.itemsize == 24


class Layer(C.Structure):
    """``dctfp_layer`` (include/dctfp.h)."""
    _fields_ = [('seq_data', C.POINTER(C.c_void_p)), ('ld', C.c_int64), ('n_cols', C.c_int32),
                ('dtype', C.c_int32), ('n_keep', C.c_int32), ('m_keep', C.c_int32),
                ('out_offset', C.c_int32), ('reserved', C.c_int32)]


class DctfpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'libdctfp error {code}: {msg}')
        self.code = code
        self.msg = msg


_lib = None
_lib_lock = threading.Lock()

EXPORTS = ('dctfp_version', 'dctfp_last_error', 'dctfp_create', 'dctfp_destroy', 'dctfp_quantize',
           'dctfp_idct_quant', 'dctfp_scale', 'dctfp_gather_rows', 'dctfp_contact_topk',
           'dctfp_contact_count', 'dctfp_stitch', 'dctfp_l1_matrix', 'dctfp_block_min', 'dctfp_row_select', 'dctfp_row_order', 'dctfp_set_option', 'dctfp_get_option', 'dctfp_profile', 'dctfp_host_device_pointer',
           'dctfp_stream_synchronize', 'dctfp_runtime_info', 'dctfp_crash_handler', 'dctfp_build_pieces', 'dctfp_contact_sort', 'dctfp_stitch_sizes',
           'dctfp_stitch_sequences', 'dctfp_quantize_windows', 'dctfp_reccut', 'dctfp_reccut_room', 'dctfp_quantize_one', 'dctfp_reccut_pieces')


def load(path: str = None):
    """Loads libdctfp.so; raises ImportError when it has not been built.  ``path`` loads another
    build of the library (A/B experiments, tools/ab_libs.py) without touching the default one."""
    global _lib
    with _lib_lock:
        if path is None and _lib is not None:
            return _lib
        if path is not None:
            return _configure(C.CDLL(path))
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f'{LIB_PATH} is missing: build the HIP extension first '
                f'(python -c "import __graft_entry__ as g; g.build()" or python build_ext.py). '
                f'dctdomain_amd has no CPU fallback.')
        # torch first: its wheel bundles libamdhip64.so.7 / libhsa-runtime64 under the same SONAMEs as
        # /opt/rocm.  Loading libdctfp.so before torch would pull the system copies in and leave the
        # process with two HSA runtimes (the second one then sees "no ROCm-capable device").
        import torch  # noqa: F401
        _lib = _configure(C.CDLL(LIB_PATH))
        _check_single_runtime()
        return _lib


def mapped_runtimes():
    """{'libamdhip64': [files], 'libhsa-runtime64': [files], 'libamd_comgr': [files]} mapped into this process
    (/proc/self/maps; no HIP call, so it is safe in a process that must not touch the GPU)."""
    found = {'libamdhip64': [], 'libhsa-runtime64': [], 'libamd_comgr': []}
    try:
        with open('/proc/self/maps') as f:
            for line in f:
                i = line.find('/')
                if i < 0:
                    continue
                path = line[i:].strip()
                for name, paths in found.items():
                    if name in os.path.basename(path) and path not in paths:
                        paths.append(path)
    except OSError:
        pass
    return found


def _check_single_runtime():
    """libdctfp.so is compiled by /opt/rocm's hipcc (ROCm 7.2) and bound -- on purpose -- to the HIP runtime of the process
    that hands it device pointers and streams: torch's bundled libamdhip64 (ROCm 7.0; same SONAME libamdhip64.so.7, which is
    why `import torch` comes first in load()).  A second copy of the runtime in the process would own no device memory of
    ours and no stream of torch's: refuse to run like that instead of failing later in some launch."""
    hip = mapped_runtimes()['libamdhip64']
    if len(hip) > 1 and os.environ.get('DCTFP_ALLOW_RUNTIME_MIX') != '1':
        raise ImportError('two HIP runtimes are mapped into this process: ' + ', '.join(hip) + '. libdctfp.so must share the '
                          'runtime of the torch that owns its tensors (import torch before anything that loads /opt/rocm\'s '
                          'libamdhip64; DCTFP_ALLOW_RUNTIME_MIX=1 overrides)')


def runtime_report(lib=None) -> str:
    """The text of ``dctfp_runtime_info`` (include/dctfp.h) + the sha256 of the libraries in use: printed in the header of the
    GPU test suite, by ``__graft_entry__.smoke()`` and by ``bench.py``, so that every GPU log says which HIP / HSA runtime
    files the hipcc-7.2-built kernels actually ran on.  Calls the HIP runtime (GPU processes only)."""
    import hashlib
    lib = lib if lib is not None else load()
    buf = C.create_string_buffer(8192)
    n = lib.dctfp_runtime_info(buf, len(buf))
    lines = [buf.value.decode('utf-8', 'replace').rstrip(), f'distinct libamdhip64 mapped: {n}']
    for path in (LIB_PATH, EXPERIMENTS_LIB_PATH, RECCUT_LIB_PATH):
        if os.path.exists(path):
            with open(path, 'rb') as f:
                lines.append(f'sha256 {hashlib.sha256(f.read()).hexdigest()}  {os.path.basename(path)}')
    return '\n'.join(lines)


def _configure(lib):
    """Declares the argument / result types of every export of include/dctfp.h."""
    if True:
        lib.dctfp_version.restype = C.c_int
        lib.dctfp_last_error.restype = C.c_char_p
        lib.dctfp_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.dctfp_destroy.argtypes = [C.c_void_p]
        lib.dctfp_quantize.argtypes = [C.c_void_p, C.POINTER(Layer), C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        lib.dctfp_quantize_windows.argtypes = [C.c_void_p, C.POINTER(Layer), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                               C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        lib.dctfp_quantize_one.argtypes = [C.c_void_p, C.POINTER(Layer), C.c_int32, C.c_int64, C.c_char_p, C.c_int64, C.c_int32, C.c_void_p,
                                           C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dctfp_reccut.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dctfp_reccut_room.argtypes = [C.c_int32]
        lib.dctfp_reccut_pieces.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p,
                                            C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.dctfp_idct_quant.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64,
                                         C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dctfp_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        lib.dctfp_gather_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64,
                                          C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        lib.dctfp_contact_topk.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dctfp_contact_count.argtypes = [C.c_int32, C.c_double]
        lib.dctfp_contact_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dctfp_l1_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                        C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
        lib.dctfp_block_min.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dctfp_row_select.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
        lib.dctfp_row_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        lib.dctfp_stitch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
        lib.dctfp_stitch_sizes.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
        lib.dctfp_stitch_sequences.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                               C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        lib.dctfp_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        lib.dctfp_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
        lib.dctfp_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        lib.dctfp_host_device_pointer.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        lib.dctfp_stream_synchronize.argtypes = [C.c_void_p]
        lib.dctfp_runtime_info.argtypes = [C.c_char_p, C.c_int64]
        lib.dctfp_crash_handler.argtypes = [C.c_int]
        lib.dctfp_build_pieces.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64,
                                           C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                           C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        for fn in EXPORTS:
            if fn not in ('dctfp_last_error', 'dctfp_contact_count', 'dctfp_reccut_room'):
                getattr(lib, fn).restype = C.c_int
        lib.dctfp_contact_count.restype = C.c_int64
        lib.dctfp_reccut_room.restype = C.c_int64
        return lib


def check(rc: int, lib=None):
    """Maps a return code to the exception the reference would raise.  ``lib``: the library the call went to (its last
    error message is per library and thread); default: the product library."""
    if rc == DCTFP_OK:
        return
    msg = (lib if lib is not None else load()).dctfp_last_error().decode('utf-8', 'replace')
    if rc == DCTFP_ERR_SHAPE:
        raise ValueError(msg)            # numpy's reshape failure at src/fingerprint.py:194
    if rc == DCTFP_ERR_NOMEM:
        raise MemoryError(msg)
    raise DctfpError(rc, msg)


class Context:
    """One ``dctfp_ctx`` (per process and device)."""

    def __init__(self, device: int, lib=None):
        lib = lib if lib is not None else load()
        handle = C.c_void_p()
        check(lib.dctfp_create(int(device), C.byref(handle)), lib)
        self._lib = lib
        self._h = handle
        self.device = int(device)

    def close(self):
        if self._h:
            self._lib.dctfp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError('context is closed')
        return self._h

    def set_option(self, name: str, value: int):
        check(self._lib.dctfp_set_option(self.handle, name.encode(), int(value)), self._lib)

    def get_option(self, name: str) -> int:
        v = C.c_int64()
        check(self._lib.dctfp_get_option(self.handle, name.encode(), C.byref(v)), self._lib)
        return v.value

    def profile(self):
        """(ms[2], launches[2]) of stage A / stage B since the last call ("profile" option on)."""
        ms = (C.c_double * 2)()
        n = (C.c_int64 * 2)()
        check(self._lib.dctfp_profile(self.handle, ms, n), self._lib)
        return [ms[0], ms[1]], [n[0], n[1]]


_contexts = {}
_ctx_lock = threading.Lock()


def get_context(device: int) -> Context:
    with _ctx_lock:
        key = (os.getpid(), int(device))
        ctx = _contexts.get(key)
        if ctx is None:
            ctx = Context(device)
            _contexts[key] = ctx
        return ctx


_experiment_contexts = {}


def experiments_context(device: int) -> Context:
    """A context of libdctfp_experiments.so (one per process and device) -- for tests and tools only."""
    with _ctx_lock:
        key = (os.getpid(), int(device))
        ctx = _experiment_contexts.get(key)
        if ctx is None:
            if not os.path.exists(EXPERIMENTS_LIB_PATH):
                raise ImportError(f'{EXPERIMENTS_LIB_PATH} is missing: run python build_ext.py')
            import torch  # noqa: F401  (before the library: see load())
            ctx = Context(device, load(EXPERIMENTS_LIB_PATH))
            _experiment_contexts[key] = ctx
        return ctx


_reccut = None


def load_reccut():
    """Loads libreccut.so (include/reccut.h): the in-process domain cutter."""
    global _reccut
    with _lib_lock:
        if _reccut is not None:
            return _reccut
        if not os.path.exists(RECCUT_LIB_PATH):
            raise ImportError(f'{RECCUT_LIB_PATH} is missing: run python build_ext.py')
        lib = C.CDLL(RECCUT_LIB_PATH)
        lib.reccut_predict.restype = C.c_int
        lib.reccut_predict.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double,
                                       C.c_double, C.c_char_p, C.c_int64, C.POINTER(C.c_int32)]
        lib.reccut_predict_batch.restype = C.c_int
        lib.reccut_predict_batch.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_double, C.c_double, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                             C.c_int32]
        lib.reccut_predict_packed.restype = C.c_int
        lib.reccut_predict_packed.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_double, C.c_double, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_int32]
        lib.reccut_contact_weight.restype = C.c_int32
        lib.reccut_contact_weight.argtypes = [C.c_float]
        lib.reccut_contact_weight_exact.restype = C.c_int32
        lib.reccut_contact_weight_exact.argtypes = [C.c_float]
        lib.reccut_format_packed.restype = C.c_int
        lib.reccut_format_packed.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        _reccut = lib
        return lib
