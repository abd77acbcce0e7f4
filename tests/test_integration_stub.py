"""The reference-side binding shown in INTEGRATION.md section 2 is executed as it stands (only the
library path is pointed at the in-tree build) and checked against the golden vectors, so the
document cannot drift from the C ABI."""

import os
import re
from types import SimpleNamespace

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace():
    import torch  # noqa: F401  (first: the library must bind to torch's HIP runtime)
    from dctdomain_amd import _lib
    from dctdomain_amd.domains import split_domain
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r'```python\n(# src/fingerprint\.py -- replacement body.*?)```', text, re.S)
    assert m, 'binding stub not found in INTEGRATION.md'
    code = m.group(1).replace("'libdctfp.so'", repr(_lib.LIB_PATH))
    ns = {'split_domain': split_domain}
    exec(compile(code, 'INTEGRATION.md', 'exec'), ns)
    return ns


def test_integration_md_stub_reproduces_goldens():
    ns = _stub_namespace()
    picked = [c for c in gu.cases(expect='ok') if c['id'].startswith(('two_', 'dom_', 'inline'))][:12]
    assert picked
    for case in picked:
        layers = gu.build_layers(case)
        fp = SimpleNamespace(embed={i: x for i, x in enumerate(layers)}, domains=list(case['domains']), quants={})
        ns['quantize'](fp, case['qdim'])
        exp = gu.expected(case)
        assert list(fp.quants.keys()) == case['keys'], case['id']
        for k in exp:
            np.testing.assert_array_equal(fp.quants[k], exp[k].astype(np.int64), err_msg=f"{case['id']} {k}")
        assert fp.domains == case['keys']
