"""Domain strings -> row pieces, with the exact behaviour of the reference's
``Fingerprint.get_doms`` (mgtools/DCTdomain src/fingerprint.py:145-171).

A domain string is ``"b-e"`` (1-based, inclusive) or several pieces joined by commas
(discontinuous domain).  The reference gathers ``embed[int(b)-1:int(e)]`` per piece and
drops a piece whose *begin* lies beyond the sequence -- with three quirks that a drop-in
must keep:

* ``(int(beg) or int(end)) > L`` tests ``beg`` unless ``beg == 0`` (then ``end``);
* the piece is removed from the list *while iterating over it*, so the piece that
  follows a removed one is never looked at, yet stays in the returned key;
* ``end > L`` is clipped silently, ``beg == 0`` means the slice ``[-1:end]``.
"""

from __future__ import annotations

from typing import List, Tuple


def split_domain(dom: str, n_rows: int) -> Tuple[List[Tuple[int, int]], str]:
    """Returns ``(pieces, key)``: ``pieces`` = list of ``(row_start, n_rows)`` with
    ``n_rows > 0`` in concatenation order (0-based rows), ``key`` = the cleaned domain
    string under which the reference files the fingerprint."""
    parts = dom.split(',')
    pieces: List[Tuple[int, int]] = []
    i = 0
    while i < len(parts):
        beg_s, end_s = parts[i].split('-')
        beg, end = int(beg_s), int(end_s)
        if (beg or end) > n_rows:
            parts.remove(parts[i])   # removes the first equal string, like list.remove() there
            i += 1                   # the reference's list iterator advances regardless
            continue
        start, stop, _ = slice(beg - 1, end).indices(n_rows)
        if stop > start:
            pieces.append((start, stop - start))
        i += 1
    return pieces, ','.join(parts)


def domain_rows(pieces: List[Tuple[int, int]]) -> int:
    return sum(n for _, n in pieces)
