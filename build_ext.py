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


def _library_commands(force: bool):
    """[(target, command)] of what has to be (re)built: the product library and its twin with the engineering knobs and
    test hooks compiled in (-DDCTFP_EXPERIMENTS: dctfp_set_option names that tools/ and the kernel-variant / cache tests use;
    same kernels, same dispatch)."""
    sources = [os.path.join(CSRC, 'dctfp.hip'), os.path.join(CSRC, 'kernels.hip.h'),
               os.path.join(ROOT, 'include', 'dctfp.h')]
    base = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-I', os.path.join(ROOT, 'include')]
    out = []
    for target, extra in ((LIB_PATH, []), (EXPERIMENTS_LIB_PATH, ['-DDCTFP_EXPERIMENTS'])):
        if force or _stale(target, sources):
            out.append((target, base + extra + ['-o', target, sources[0]]))
    return out


def sha256_of(path: str) -> str:
    with open(path, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> dctdomain_amd/libdctfp.so (+ libdctfp_experiments.so), the two compilations side by
    side, EACH IN ITS OWN SCRATCH DIRECTORY: hipcc leaves intermediates named after the source
    (<src>-hip-amdgcn-amd-amdhsa.hipfb, seen on the GPU box) in its working directory, and two compilations of the same
    dctfp.hip in one directory could hand each other's device code to the link step.  Prints the sha256 of what it built.
    Returns the path of the product library."""
    procs = []
    for target, cmd in _library_commands(force):
        if verbose:
            print(' '.join(cmd))
        scratch = tempfile.mkdtemp(prefix='dctfp_build_')
        env = dict(os.environ, TMPDIR=scratch)
        procs.append((target, cmd, scratch, subprocess.Popen(cmd, cwd=scratch, env=env)))
    failed = None
    for target, cmd, scratch, proc in procs:
        rc = proc.wait()
        shutil.rmtree(scratch, ignore_errors=True)
        if rc != 0 and failed is None:
            failed = subprocess.CalledProcessError(rc, cmd)
    if failed is not None:
        raise failed
    for target, _, _, _ in procs:
        print(f'sha256 {sha256_of(target)}  {os.path.relpath(target, ROOT)}')
    return LIB_PATH


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
    return paths


if __name__ == '__main__':
    print(build_all(force=True, verbose=True))
