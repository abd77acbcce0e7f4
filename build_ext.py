"""Builds libdctfp.so (the HIP kernels + C ABI) in-tree for gfx950 with hipcc.

Standalone on purpose (python build_ext.py): importing ``dctdomain_amd`` requires the
library to exist already."""

from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, 'dctdomain_amd')
CSRC = os.path.join(PKG_DIR, 'csrc')
LIB_PATH = os.path.join(PKG_DIR, 'libdctfp.so')
RECCUT_LIB_PATH = os.path.join(PKG_DIR, 'libreccut.so')
TENSOR_TABLE_PATH = os.path.join(PKG_DIR, '_tensor_table.so')


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', shutil.which('hipcc')):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC or install ROCm under /opt/rocm)')


def _stale(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


EXPERIMENTS_LIB_PATH = os.path.join(PKG_DIR, 'libdctfp_experiments.so')


#: translation units of libdctfp.so: the host side of the C ABI, and one unit per kernel family so that the device code
#: compiles side by side (the stage-A instantiations alone are two thirds of it).  `twin` = units that differ in
#: libdctfp_experiments.so (-DDCTFP_EXPERIMENTS: option names, extra walk-kernel builds); the others are compiled once and
#: linked into both libraries.
UNITS = ['dctfp.hip', 'k_walk.hip', 'k_gen.hip', 'k_reccut.hip', 'k_stage_b.hip', 'k_stage_a_f32.hip', 'k_stage_a_f64.hip', 'k_stage_a_f16.hip',
         'k_stage_a_bf16.hip']
TWIN_UNITS = ('dctfp.hip', 'k_walk.hip')
HEADERS = [os.path.join(CSRC, 'kernels.hip.h'), os.path.join(CSRC, 'launch.h'), os.path.join(ROOT, 'include', 'dctfp.h'),
           os.path.join(CSRC, 'reccut_kernel.hip.h'), os.path.join(CSRC, 'k_stage_a.inc')]
OBJ_DIR = os.path.join(ROOT, 'build', 'dctfp_objs')


def kernel_sources():
    """Every file the device code and its dispatch come from (what a PMC traffic measurement is stamped with)."""
    return [os.path.join(CSRC, u) for u in UNITS] + HEADERS[:2] + HEADERS[3:]


def sha256_of(path: str) -> str:
    with open(path, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


def _compile_flags(extra=()):
    return [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', os.path.join(ROOT, 'include'), '-I', CSRC] + list(extra)


def build_library(force: bool = False, verbose: bool = False, jobs: int = None, extra_flags=(), lib_path: str = None,
                  experiments_path: str = None, flag_units=None) -> str:
    """hipcc --offload-arch=gfx950: every unit -> an object (in parallel, EACH IN ITS OWN SCRATCH DIRECTORY: hipcc leaves
    intermediates named after the source, <src>-hip-amdgcn-amd-amdhsa.hipfb seen on the GPU box, in its working directory, and
    two compilations of the same unit there could hand each other's device code to the link step), then two links:
    dctdomain_amd/libdctfp.so and libdctfp_experiments.so.  Prints the sha256 of what it built.  `extra_flags` / `lib_path` /
    `experiments_path`: A/B builds with other -D switches (tools/build_variant.sh).  Returns the path of the product library."""
    lib_path = os.path.abspath(lib_path or LIB_PATH)
    experiments_path = os.path.abspath(experiments_path) if experiments_path else (EXPERIMENTS_LIB_PATH if lib_path == LIB_PATH else None)
    # `flag_units`: the units the extra flags are meant for (an A/B knob of the walk kernel does not touch stage A); the
    # others are taken from the default build -- a variant costs one or two compilations instead of all seven
    tag = hashlib.sha256(' '.join(extra_flags).encode()).hexdigest()[:10] if extra_flags else 'default'
    todo = []   # (object, command)
    objs = {'product': [], 'experiments': []}
    for unit in UNITS:
        src = os.path.join(CSRC, unit)
        deps = [src] + HEADERS
        flagged = bool(extra_flags) and (flag_units is None or unit in flag_units)
        obj_dir = os.path.join(OBJ_DIR, tag if flagged else 'default')
        os.makedirs(obj_dir, exist_ok=True)
        variants = [('product', [])] + ([('experiments', ['-DDCTFP_EXPERIMENTS'])] if unit in TWIN_UNITS else [])
        for kind, flags in variants:
            if kind == 'experiments' and experiments_path is None:
                continue
            obj = os.path.join(obj_dir, f'{unit[:-4]}.{kind}.o')
            if force or _stale(obj, deps):
                todo.append((obj, _compile_flags((list(extra_flags) if flagged else []) + flags) + ['-c', src, '-o', obj]))
            objs[kind].append(obj)
        if unit not in TWIN_UNITS:
            objs['experiments'].append(os.path.join(obj_dir, f'{unit[:-4]}.product.o'))
    jobs = jobs or max(1, min(len(todo), (os.cpu_count() or 4)))
    running, failed = [], None
    queue = list(todo)
    while queue or running:
        while queue and len(running) < jobs:
            obj, cmd = queue.pop(0)
            if verbose:
                print(' '.join(cmd))
            scratch = tempfile.mkdtemp(prefix='dctfp_build_')
            running.append((obj, cmd, scratch, subprocess.Popen(cmd, cwd=scratch, env=dict(os.environ, TMPDIR=scratch))))
        obj, cmd, scratch, proc = running.pop(0)
        rc = proc.wait()
        shutil.rmtree(scratch, ignore_errors=True)
        if rc != 0:
            if os.path.exists(obj):
                os.remove(obj)
            failed = failed or subprocess.CalledProcessError(rc, cmd)
    if failed is not None:
        raise failed
    built = []
    for kind, target in (('product', lib_path), ('experiments', experiments_path)):
        if target is None:
            continue
        if force or todo or _stale(target, objs[kind]):
            cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', target] + objs[kind]
            if verbose:
                print(' '.join(cmd))
            scratch = tempfile.mkdtemp(prefix='dctfp_link_')
            try:
                subprocess.run(cmd, check=True, cwd=scratch, env=dict(os.environ, TMPDIR=scratch))
            finally:
                shutil.rmtree(scratch, ignore_errors=True)
            built.append(target)
    for target in built:
        print(f'sha256 {sha256_of(target)}  {os.path.relpath(target, ROOT)}')
    return lib_path


def build_all(force: bool = False, verbose: bool = False):
    """Every native piece of the product (HIP kernels + host-side C++ helpers)."""
    paths = [build_library(force=force, verbose=verbose), EXPERIMENTS_LIB_PATH]
    reccut_src = os.path.join(CSRC, 'reccut.cpp')
    if os.path.exists(reccut_src):
        if force or _stale(RECCUT_LIB_PATH, [reccut_src, os.path.join(ROOT, 'include', 'reccut.h')]):
            cmd = [os.environ.get('CXX', 'g++'), '-O3', '-std=c++17', '-ffp-contract=off', '-pthread', '-shared', '-fPIC',
                   '-I', os.path.join(ROOT, 'include'), '-o', RECCUT_LIB_PATH, reccut_src]
            if verbose:
                print(' '.join(cmd))
            subprocess.run(cmd, check=True)
        paths.append(RECCUT_LIB_PATH)
    tt = build_tensor_table(force=force, verbose=verbose)
    if tt:
        paths.append(tt)
    return paths


def build_tensor_table(force: bool = False, verbose: bool = False):
    """dctdomain_amd/_tensor_table.so: the geometry of a list of torch tensors in one C++ pass (host plumbing of the Python
    layer -- the C ABI takes plain pointers; see csrc/tensor_table.cpp).  Needs torch's headers; returns None (and the
    Python layer reads the tensors one attribute at a time) where they are missing."""
    src = os.path.join(CSRC, 'tensor_table.cpp')
    if not os.path.exists(src):
        return None
    if not force and not _stale(TENSOR_TABLE_PATH, [src]):
        return TENSOR_TABLE_PATH
    try:
        import sysconfig
        import torch
        from torch.utils import cpp_extension
        incs = cpp_extension.include_paths()
    except Exception as exc:      # noqa: BLE001  (no torch headers: the helper is optional)
        print(f'build_ext: _tensor_table not built ({exc})')
        return None
    libdir = os.path.join(os.path.dirname(torch.__file__), 'lib')
    cmd = [os.environ.get('CXX', 'g++'), '-O2', '-std=c++17', '-fPIC', '-shared',
           f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}', '-I', sysconfig.get_paths()['include']]
    for inc in incs:
        cmd += ['-isystem', inc]
    cmd += [src, '-o', TENSOR_TABLE_PATH, '-L', libdir, '-ltorch_python', '-lc10', '-ltorch_cpu', '-ltorch', f'-Wl,-rpath,{libdir}']
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print('build_ext: _tensor_table failed to build (the Python layer falls back to per-tensor reads):\n' + r.stderr[-2000:])
        return None
    return TENSOR_TABLE_PATH


if __name__ == '__main__':
    print(build_all(force=True, verbose=True))
