#!/usr/bin/env python3
"""Can fence class 1 (an exactly constant channel) be closed by a rule?  Build container only: IMPORTS
THE REFERENCE (like the make_golden*.py generators next to it; data out, no source copied).

Asks the reference, at the first lengths of `fence_golden.json: const.noisy_L`, what it writes for other constants and
other neighbouring channels.  Outcome (profiles/r04/fence_class1_probe.txt): whether a length is "noisy" depends on the VALUE
of the constant (L = 5: 1.2345 gives a non-zero block, 1.5 / -0.7 / 100 the zero block; L = 11: 1.5 gives a different
non-zero block), exactly-doubled constants give the same block (power-of-two scaling commutes with rounding), and the
block changes with the other channels (the noise row is mixed with them by the channel-axis DCT).  The "225 lengths" are
therefore a property of the value 1.2345, not of the length: what the reference does there is the round-off of pocketfft's
radix-5 / 11 / 13 / generic passes on the constant's mantissa, followed by the ulp-level rounding of its length-3 inverse --
reproducing it means re-implementing those passes operation for operation, not a closed rule.  The fence stays."""
import hashlib
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
sys.path.insert(0, '/root/reference/src')
from fingerprint import Fingerprint  # noqa: E402  (the reference)
from recipes import make_input  # noqa: E402

warnings.simplefilter('ignore')
const = json.load(open(os.path.join(HERE, 'fence_golden.json')))['const']


def block(L, val, seed=const['seed0'], D=const['D'], col=const['col']):
    x = make_input('gauss', L, D, seed + L)
    x[:, col] = np.float32(val)
    fp = Fingerprint(pid='f', seq='A' * L, embed={0: x}, domains=[f'1-{L}'])
    fp.quantize([3, 80])
    q = np.asarray(fp.quants[f'1-{L}']).astype(np.int16)
    return 'zero block' if not q.any() else 'block ' + hashlib.md5(q.tobytes()).hexdigest()[:6]


def factors(n):
    f, d = [], 2
    while d * d <= n:
        while n % d == 0:
            f.append(d)
            n //= d
        d += 1
    return f + ([n] if n > 1 else [])


print(f"{len(const['noisy_L'])} of the lengths 3..2000 give a non-zero block for the constant {const['value']}; the first: "
      f"{[(L, factors(L)) for L in const['noisy_L'][:16]]}")
for L in const['noisy_L'][:8]:
    print(f'L = {L:3d}:', {v: block(L, v) for v in (1.2345, 2.469, 1.5, -0.7, 100.0)})
L = const['noisy_L'][0]
print(f'L = {L}, constant 1.2345, other neighbouring channels (seed 777): {block(L, 1.2345, seed=777)}  (golden neighbours: {block(L, 1.2345)})')
