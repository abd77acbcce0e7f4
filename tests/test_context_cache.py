"""The context's cosine-table cache (dctfp.hip: basis_lookup / basis_publish / basis_purge) under the two events a test
never met before round 3: an error between the table lookup and the kernel that fills the fresh tables, and the arena
starting over while a call of another stream is still in flight."""

import numpy as np
import pytest

from oracle import dct_oracle as orc
from recipes import make_input

pytestmark = pytest.mark.gpu


def _xctx(torch):
    """The test hooks (test_fail_once, basis_cap_kb, basis_*) exist in libdctfp_experiments.so only."""
    from dctdomain_amd import _lib
    return _lib.experiments_context(torch.cuda.current_device())


def _batch(dd, torch, lens, seed, D=640):
    xs = [make_input('esm', L, D, seed + i) for i, L in enumerate(lens)]
    table = dd.PieceTable.whole_sequences(lens)
    lb = dd.LayerBatch([torch.from_numpy(x).cuda() for x in xs], 3, 80)
    return xs, table, lb


def _expect(xs, lens):
    return np.stack([orc.quantize([x], [f'1-{L}'], [3, 80])[f'1-{L}'] for x, L in zip(xs, lens)])


def test_failed_call_leaves_no_unfilled_table_in_the_cache():
    """a failure after basis_lookup used to leave the new lengths cached but pointing at memory no
    kernel had filled -- every later call with those lengths returned wrong fingerprints with rc == OK."""
    import torch
    import dctdomain_amd as dd
    ctx = _xctx(torch)
    lens = [1777, 1778, 1779, 91]                 # lengths no other test uses: fresh tables for this context
    xs, table, lb = _batch(dd, torch, lens, 4100)
    before = ctx.get_option('basis_tables')
    ctx.set_option('test_fail_once', 1)
    with pytest.raises(MemoryError, match='injected failure'):
        dd.quantize_batch([lb], table, ctx=ctx)
    assert ctx.get_option('basis_tables') == before            # nothing published
    out = dd.quantize_batch([lb], table, ctx=ctx).cpu().numpy()         # the retry fills and publishes them
    assert ctx.get_option('basis_tables') >= before + 3
    np.testing.assert_array_equal(out.astype(np.int64), _expect(xs, lens))
    out2 = dd.quantize_batch([lb], table, ctx=ctx).cpu().numpy()        # ... and a call served from the cache agrees
    np.testing.assert_array_equal(out2, out)


def test_arena_restart_with_a_call_in_flight_on_another_stream():
    """basis_purge (device synchronisation + free of every table) had never run.  Cap the arena at 64 KB,
    put a long call on stream A, then ask for new lengths on stream B: the second call starts the arena over while the
    first is in flight; both results must be right and the restart counted."""
    import torch
    import dctdomain_amd as dd
    ctx = _xctx(torch)
    lens_a = [1500 + 3 * i for i in range(300)]                 # 300 tables, ~48 KB each: far above the cap
    lens_b = [1201 + 2 * i for i in range(40)]
    xa, ta, la = _batch(dd, torch, lens_a, 5200)
    xb, tb, lb = _batch(dd, torch, lens_b, 6200)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    restarts = ctx.get_option('basis_restarts')
    old_cap = ctx.get_option('basis_cap_kb')
    torch.cuda.synchronize()
    try:
        ctx.set_option('basis_cap_kb', 64)
        out_a = dd.quantize_batch([la], ta, stream=sa, ctx=ctx)          # fills ~14 MB of tables: over the cap from now on
        out_b = dd.quantize_batch([lb], tb, stream=sb, ctx=ctx)          # -> purge (waits for stream A), fresh tables, other stream
        out_a2 = dd.quantize_batch([la], ta, stream=sa, ctx=ctx)         # -> purge again, call B possibly still in flight
        torch.cuda.synchronize()
    finally:
        ctx.set_option('basis_cap_kb', old_cap)
    assert ctx.get_option('basis_restarts') >= restarts + 2
    pick = [0, 1, 150, 299]
    np.testing.assert_array_equal(out_a.cpu().numpy()[pick].astype(np.int64), _expect([xa[i] for i in pick], [lens_a[i] for i in pick]))
    np.testing.assert_array_equal(out_a2.cpu().numpy(), out_a.cpu().numpy())
    pick = [0, 20, 39]
    np.testing.assert_array_equal(out_b.cpu().numpy()[pick].astype(np.int64), _expect([xb[i] for i in pick], [lens_b[i] for i in pick]))
    # and the context is healthy afterwards, cache refilled under the normal cap
    out_b2 = dd.quantize_batch([lb], tb, ctx=ctx).cpu().numpy()
    np.testing.assert_array_equal(out_b2, out_b.cpu().numpy())


def test_cached_tables_filled_on_one_stream_are_ordered_for_another():
    """tables filled by a call on stream A, then a call on stream B that needs them AND fresh ones --
    the fresh fill re-records the one event that guards the cache; the older fills must stay ordered behind it."""
    import torch
    import dctdomain_amd as dd
    lens_a = [901 + i for i in range(64)]
    lens_b = lens_a[:32] + [1001 + i for i in range(32)]
    xa, ta, la = _batch(dd, torch, lens_a, 7100)
    xb = xa[:32] + [make_input('esm', L, 640, 7300 + i) for i, L in enumerate(lens_b[32:])]
    tb = dd.PieceTable.whole_sequences(lens_b)
    lb = dd.LayerBatch([torch.from_numpy(x).cuda() for x in xb], 3, 80)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    out_a = dd.quantize_batch([la], ta, stream=sa)            # (the product library: no hook needed here)
    out_b = dd.quantize_batch([lb], tb, stream=sb)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out_b.cpu().numpy()[:32], out_a.cpu().numpy()[:32])
    pick = [0, 31, 32, 63]
    np.testing.assert_array_equal(out_b.cpu().numpy()[pick].astype(np.int64), _expect([xb[i] for i in pick], [lens_b[i] for i in pick]))
