"""AddressSanitizer over the HOST side of libdctfp.so, on this machine, without a GPU.

GPU AddressSanitizer is not available on the pool, and a wrong index in a job table shows up on the GPU box as a fault
(or as an abort without a message).  So the host code of dctdomain_amd/csrc/dctfp.hip is compiled on its own
(`hipcc --cuda-host-only -fsanitize=address`), linked against a stand-in for the HIP runtime (tests/asan/hip_stub.cpp:
device memory = host memory; a kernel launch walks the job tables the way the kernel's waves do and touches every address
they would) and driven over a few hundred seeded random batches (tests/asan/driver.cpp): every shape class, option,
fused-group layout, the split at giant domains, failure injection (also std::bad_alloc from any allocation of the table
build: the C ABI must answer DCTFP_ERR_NOMEM, not std::terminate) and arena restarts of the cosine-table cache.  Round 4: the
entry points either side of dctfp_quantize too -- the domain-string parser, the top-k job / stripe tables and sorting-network
groups, the window jobs of the stitcher in both forms, the L1 matrix / block minima / row select with its candidate scratch
-- over random and malformed arguments, again with allocation failures injected."""

import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
CLANG = '/opt/rocm/lib/llvm/bin/clang++'


def _records(text, names):
    """{struct name: [field names in order]} for the job-table records."""
    out = {}
    for name in names:
        body = re.search(r'struct %s \{(.*?)\n\};' % name, text, re.S).group(1)
        body = re.sub(r'//[^\n]*', '', body)
        fields = []
        for decl in body.split(';'):
            decl = decl.strip()
            if decl:
                first, *rest = [d.strip() for d in decl.split(',')]
                fields.append(re.split(r'[\s\*]+', first)[-1])
                fields.extend(re.split(r'[\s\*]+', r)[-1] for r in rest)
        out[name] = fields
    return out


def test_stub_records_match_the_kernel_header():
    names = ['JobA', 'PieceA', 'JobB', 'Walk', 'Run', 'BasisJob']
    with open(os.path.join(ROOT, 'dctdomain_amd', 'csrc', 'kernels.hip.h')) as fh:
        a = _records(fh.read(), names)
    with open(os.path.join(ROOT, 'tests', 'asan', 'hip_stub.cpp')) as fh:
        b = _records(fh.read(), names)
    assert a == b


@pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(CLANG)), reason='needs the ROCm compilers')
def test_host_code_under_address_sanitizer(tmp_path):
    flags = ['-O1', '-g', '-std=c++17', '-fsanitize=address', '-fno-omit-frame-pointer']
    inc = ['-I', os.path.join(ROOT, 'include'), '-I', os.path.join(ROOT, 'dctdomain_amd', 'csrc')]
    import build_ext
    stub_o, syms, exe = (str(tmp_path / n) for n in ('hip_stub.o', 'fatbin_syms.cpp', 'driver'))
    inc += ['-I', os.path.join(ROOT, 'dctdomain_amd', 'csrc')]
    host_objs, procs = [], []
    for unit in build_ext.UNITS:          # every unit of the library, host side only (the launchers live in their own units)
        obj = str(tmp_path / (unit[:-4] + '.host.o'))
        host_objs.append(obj)
        procs.append(subprocess.Popen([HIPCC, '--offload-arch=gfx950', '--cuda-host-only', '-fPIC', '-DDCTFP_EXPERIMENTS', *flags, *inc, '-c',
                                       os.path.join(ROOT, 'dctdomain_amd', 'csrc', unit), '-o', obj]))
    assert all(p.wait() == 0 for p in procs)
    subprocess.run([CLANG, '-D__HIP_PLATFORM_AMD__', '-fPIC', *flags, '-I', '/opt/rocm/include', '-c',
                    os.path.join(ROOT, 'tests', 'asan', 'hip_stub.cpp'), '-o', stub_o], check=True)
    # the host objects refer to the device code they would carry: empty ones will do (no kernel ever runs here)
    undefined = subprocess.run(['nm', '-u', *host_objs], check=True, capture_output=True, text=True).stdout
    with open(syms, 'w') as fh:
        for sym in sorted(set(re.findall(r'__hip_fatbin_[0-9a-f]+', undefined))):
            fh.write(f'extern "C" {{ extern const unsigned long long {sym}[4]; const unsigned long long {sym}[4] = {{0, 0, 0, 0}}; }}\n')
    subprocess.run([CLANG, *flags, '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'asan', 'driver.cpp'), syms,
                    *host_objs, stub_o, '-o', exe, '-lpthread'], check=True)
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0')
    for seed in (1, 2, 3):
        r = subprocess.run([exe, '250', str(seed)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
        assert 'no memory error' in r.stdout
        # std::bad_alloc inside the table build comes back as DCTFP_ERR_NOMEM through the exception barrier of the C ABI
        assert int(re.search(r'reported as DCTFP_ERR_NOMEM: (\d+)', r.stdout).group(1)) >= 10, r.stdout
        walked = int(re.search(r'(\d+) walk-kernel launches', r.stdout).group(1))
        assert walked >= 12, r.stdout          # the production path is among what was exercised (a seed's share: 15-40 of 500 calls)
        # the entry points either side of dctfp_quantize (round 4): domain-string parser, top-k tables, stitch jobs, row select
        other = re.search(r'the other entry points: (\d+) calls \((\d+) ended in an expected error\)', r.stdout)
        assert int(other.group(1)) >= 500 and int(other.group(2)) < int(other.group(1)) // 3, r.stdout
        print(r.stdout.strip().splitlines()[-3:])
