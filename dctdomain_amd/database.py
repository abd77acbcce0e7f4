"""``Database`` -- the output side of the fingerprint path: SQLite ``.db``, ``-dct.npz``, ``.dom``
and (optionally) the FAISS ``.index``, layout-compatible with mgtools/DCTdomain
``src/database.py`` so that the reference's ``query_db.py`` / ``dct-sim.py`` read what this
build writes.

Layout contract (reference ``src/database.py``):
* tables ``sequences(pid PK, sequence, length, fpcount)``, ``fingerprints(vid PK, domain,
  fingerprint BLOB, pid FK)``, ``metadata(datetime PK, seq_num, avg_len, fp_num, seqs_fp)`` (:100-126);
* sequences inserted in ascending length order, stable (:79);
* one row per domain; the blob is ``np.save`` of the int8 vector (:215-223); ``fpcount`` marks a
  protein as done -- it is the resume mechanism (:150, :211-213);
* ``-dct.npz`` = ``np.savez(sid, idx, dom, dct)`` in table order, consecutive equal pids grouped (:351-375);
* ``.dom`` = ``pid ndom d1;d2`` without the trailing whole-protein entry (:378-392).

Built for batches, not for one protein at a time: a batch of proteins is one transaction
(``add_fprints``; the reference commits per protein), a million blobs share one cached npy header,
renumbering is two set-based statements, and ``yield_seqs`` does not lose the tail of the input
in multi-sequence mode (reference bug at :166-178)."""

from __future__ import annotations

import os
import sqlite3
import struct
from datetime import datetime
from io import BytesIO
from itertools import groupby
from typing import Iterable, Iterator, List, Tuple

import numpy as np

# column lists of the three tables (names, types and constraints are the reference's, :100-126)
_TABLES = (
    ('sequences', 'pid text PRIMARY KEY, sequence text NOT NULL, length integer NOT NULL, fpcount integer NOT NULL'),
    ('fingerprints', 'vid integer PRIMARY KEY, domain text NOT NULL, fingerprint blob NOT NULL, pid text NOT NULL, '
                     'FOREIGN KEY(pid) REFERENCES sequences(pid)'),
    ('metadata', 'datetime text PRIMARY KEY, seq_num integer NOT NULL, avg_len real NOT NULL, '
                 'fp_num integer NOT NULL, seqs_fp string NOT NULL'),
)
_FASTA_SUFFIXES = ('.fa', '.fasta')
_NPY_HEADERS = {}


def _npy_bytes(vec: np.ndarray) -> bytes:
    """``np.save`` bytes of a 1-D C-contiguous array; the header of a (dtype, shape) is built by
    numpy once and reused (a million blobs share one header)."""
    vec = np.ascontiguousarray(vec)
    key = (vec.dtype.str, vec.shape)
    head = _NPY_HEADERS.get(key)
    if head is None:
        buf = BytesIO()
        np.save(buf, vec, allow_pickle=True)
        blob = buf.getvalue()
        head = blob[:len(blob) - vec.nbytes]
        _NPY_HEADERS[key] = head
        return blob
    return head + vec.tobytes()


def _npy_vector(blob: bytes) -> np.ndarray:
    """The array of an ``np.save`` blob.  Blobs written by this module (or by the reference: same numpy call, same header)
    start with a header already seen -- then the payload is taken as it is, without re-parsing the header."""
    for (descr, shape), head in _NPY_HEADERS.items():
        if len(blob) == len(head) + int(np.prod(shape)) * np.dtype(descr).itemsize and blob.startswith(head):
            return np.frombuffer(blob, dtype=descr, offset=len(head)).reshape(shape)
    vec = np.load(BytesIO(blob), allow_pickle=False)     # int8 vectors only: a crafted .db must not reach the unpickler
    if vec.ndim == 1 and vec.flags.c_contiguous:
        _NPY_HEADERS.setdefault((vec.dtype.str, vec.shape), blob[:len(blob) - vec.nbytes])
    return vec


def _fasta_records(path: str) -> Iterator[Tuple[str, str]]:
    """(pid, sequence) per record; pid = the header up to the first blank, a repeated pid keeps its
    first position and its last sequence (what a dict filled line by line gives, :60-77)."""
    pid, chunks = None, []
    with open(path, encoding='utf8') as handle:
        for raw in handle:
            text = raw.strip()
            if raw.startswith('>'):
                if pid is not None:
                    yield pid, ''.join(chunks)
                pid, chunks = text.split()[0][1:], []
            else:
                chunks.append(text)
    if pid is not None:
        yield pid, ''.join(chunks)


class _Serial:
    """Stand-in for the reference's shared ``multiprocessing.Value`` vid counter."""

    def __init__(self, value: int):
        self.value = value


class Database:
    """SQLite database of sequences and fingerprints (reference ``Database``, :16-392)."""

    def __init__(self, dbfile: str, fafile: str = None):
        from_fasta = bool(fafile) and fafile.endswith(_FASTA_SUFFIXES)
        if from_fasta:
            print(f'Reading file: {fafile}')
            self.path = dbfile
            self.init_db(self.read_fasta(fafile))
            return
        if not os.path.exists(dbfile):
            raise FileNotFoundError(f'Database file not found: {dbfile}')
        print(f'Opening database: {dbfile}')
        self._open(dbfile)

    def _open(self, name: str):
        self.path = os.path.splitext(name)[0]
        self.conn = sqlite3.connect(self.path + '.db')
        self.cur = self.conn.cursor()

    def _one(self, sql: str, args=()):
        return self.cur.execute(sql, args).fetchone()

    def close(self):
        print(f'Closing database: {self.path}\n')
        self.conn.close()

    # -- input ---------------------------------------------------------------------------
    def read_fasta(self, fafile: str) -> dict:
        """pid -> sequence, ascending length (stable), as :60-81."""
        seqs = {}
        for pid, seq in _fasta_records(fafile):
            seqs[pid] = seq
        by_length = sorted(seqs, key=lambda p: len(seqs[p]))
        return {p: seqs[p] for p in by_length}

    def init_db(self, seqs: dict):
        self._open(self.path)
        for name, columns in _TABLES:
            self.cur.execute(f'CREATE TABLE IF NOT EXISTS {name} ({columns})')
        self.cur.executemany('INSERT OR IGNORE INTO sequences(pid, sequence, length, fpcount) VALUES(?, ?, ?, 0)',
                             ((pid, seq, len(seq)) for pid, seq in seqs.items()))
        self.conn.commit()

    def pending(self, dim1: int = 3, dim2: int = 80) -> List[Tuple[str, str]]:
        """(pid, sequence) of every protein still to fingerprint (``fpcount = 0``), in table order,
        without those too short to quantise (``(length-2)*dim2 < dim1*dim2``, :149-155)."""
        todo = self.cur.execute('SELECT pid, sequence, length FROM sequences WHERE fpcount = 0').fetchall()
        return [(pid, seq) for pid, seq, length in todo if (length - 2) * dim2 >= dim1 * dim2]

    def yield_seqs(self, maxlen: int, cpu: int, dim1: int = 3, dim2: int = 80):
        """Batches of (pid, sequence): proteins are packed while their total length stays within
        ``maxlen`` and their number within ``cpu + 1``; a protein longer than ``maxlen`` travels alone
        (and is embedded in windows); ``maxlen = 1`` gives one protein per batch, the reference's CPU
        mode (make_db.py:136).  Same intent as :136-178, but every pending protein is yielded exactly
        once (the reference drops the tail of the input when its last batch holds several)."""
        todo = self.pending(dim1, dim2)
        if not todo:
            print('No sequences to fingerprint!\n')
            return
        batch, residues = [], 0
        for item in todo:
            full = residues + len(item[1]) > maxlen or len(batch) > cpu
            if batch and full:
                yield batch
                batch, residues = [], 0
            batch.append(item)
            residues += len(item[1])
        yield batch

    def get_last_vid(self) -> int:
        top = self._one('SELECT MAX(vid) FROM fingerprints')[0]
        return 1 if top is None else top + 1

    # -- output ---------------------------------------------------------------------------
    def add_fprint(self, fp, lock=None, counter=None):
        """One protein (reference signature, :197-224).  ``counter`` is any object with ``.value``."""
        self.add_fprints([fp], lock, counter)

    def add_fprints(self, fps: Iterable, lock=None, counter=None):
        """Many proteins in one transaction.  Each item needs ``pid``, ``domains`` and
        ``quants[dom]`` (0..127 ints) -- a ``Fingerprint`` or anything shaped like one.  vids are
        drawn from ``counter`` (incremented first, like :216-218), under ``lock`` when one is given."""
        if counter is None:
            # own numbering: continue right after the last vid, so that a fresh build comes out as 1..N and
            # ``rename_vid`` has nothing to move (a caller's shared counter keeps the reference's off-by-one start)
            counter = _Serial(self.get_last_vid() - 1)

        def next_vid():
            counter.value += 1
            return counter.value

        rows, done = [], []
        for fp in fps:
            done.append((len(fp.domains), fp.pid))
            for dom in fp.domains:
                if lock is None:
                    vid = next_vid()
                else:
                    with lock:
                        vid = next_vid()
                rows.append((vid, dom, _npy_bytes(np.asarray(fp.quants[dom]).astype(np.int8)), fp.pid))
        self.cur.executemany('UPDATE sequences SET fpcount = ? WHERE pid = ?', done)
        self.cur.executemany('INSERT INTO fingerprints(vid, domain, fingerprint, pid) VALUES(?, ?, ?, ?)', rows)
        self.conn.commit()

    def _all_fprints(self) -> np.ndarray:
        blobs = self.cur.execute('SELECT fingerprint FROM fingerprints')
        return np.array([_npy_vector(b) for b, in blobs], dtype=np.int8)

    def create_index(self):
        """``faiss.IndexFlatL2`` over all fingerprints -> ``<path>.index`` (:227-243).  Uses faiss when
        it is installed; otherwise writes the same flat-index file with ``write_flat_index``
        (format restated from faiss 1.7.4's index_write.cpp -- parity unpinned: no faiss here to
        read it back)."""
        vectors = self._all_fprints().astype(np.float32)
        target = self.path + '.index'
        try:
            import faiss
        except ImportError:
            write_flat_index(target, vectors)
        else:
            flat = faiss.IndexFlatL2(vectors.shape[1])
            flat.add(vectors)
            faiss.write_index(flat, target)

    def load_fprints(self, pid: str = '') -> list:
        found = self.cur.execute('SELECT vid, fingerprint FROM fingerprints WHERE pid = ?', (pid,))
        return [(vid, _npy_vector(blob).copy()) for vid, blob in found]      # writable, like the reference's np.load

    def rename_vid(self):
        """vids 1..N in table order (:268-282), set-based: rows move to -1..-N first so that the
        PRIMARY KEY stays unique, then flip sign.  The usual case -- one writer, vids consecutive from some
        offset -- is a single shift statement."""
        vids = [v for v, in self.cur.execute('SELECT vid FROM fingerprints')]
        if vids and vids[0] > 1 and all(b - a == 1 for a, b in zip(vids, vids[1:])):
            # (a direct "vid = vid - shift" trips over the key it is about to free: measured, IntegrityError)
            self.cur.execute('UPDATE fingerprints SET vid = -(vid - ?)', (vids[0] - 1,))
            self.cur.execute('UPDATE fingerprints SET vid = -vid')
        elif any(v != i for i, v in enumerate(vids, 1)):
            self.cur.executemany('UPDATE fingerprints SET vid = ? WHERE vid = ?',
                                 ((-i, v) for i, v in enumerate(vids, 1)))
            self.cur.execute('UPDATE fingerprints SET vid = -vid')
        self.conn.commit()

    def update_metadata(self):
        print('Updating metadata...')
        n_seq, mean_len = self._one('SELECT COUNT(*), AVG(length) FROM sequences')
        n_fp, n_done = self._one('SELECT SUM(fpcount), COUNT(*) FROM sequences WHERE fpcount > 0')
        stamp = datetime.now().strftime('%Y-%m-%d %H:%M:%S')
        self.cur.execute('INSERT OR REPLACE INTO metadata(datetime, seq_num, avg_len, fp_num, seqs_fp) '
                         'VALUES(?, ?, ?, ?, ?)', (stamp, n_seq, mean_len, n_fp or 0, f'{n_done}/{n_seq}'))
        self.conn.commit()
        self.db_info()

    def db_info(self):
        latest = self._one('SELECT * FROM metadata ORDER BY datetime DESC LIMIT 1')
        if latest is None:
            self.update_metadata()      # prints the fresh row itself
            return
        stamp, n_seq, mean_len, n_fp, done = latest
        print(f'Last Updated: {stamp}\nNumber of Sequences: {n_seq}\nAverage Sequence Length: {mean_len:.2f}\n'
              f'Number of Fingerprints: {n_fp} ({done} fingerprinted)\n')

    def seq_info(self, seq: str):
        print(f'Protein ID: {seq}')
        hit = self._one('SELECT sequence FROM sequences WHERE pid = ?', (seq,))
        if hit is None:
            print('Sequence not found in database\n')
            return
        print(f'Sequence: {hit[0]}')
        names = [d for d, in self.cur.execute('SELECT domain FROM fingerprints WHERE pid = ?', (seq,))]
        print(f'Domains: {", ".join(names)}\n' if names else 'No domains in database\n')

    def save_fprints(self, file: str):
        """``np.savez(file, sid=, idx=, dom=, dct=)`` (:351-375): ``idx`` = prefix offsets of the runs
        of consecutive equal pids."""
        table = self.cur.execute('SELECT pid, domain, fingerprint FROM fingerprints').fetchall()
        sid, idx = [], [0]
        for pid, run in groupby(table, key=lambda row: row[0]):
            sid.append(pid)
            idx.append(idx[-1] + sum(1 for _ in run))
        if not table:
            idx = [0]
        # lists in, like the reference: numpy picks <U / int64 / int8 exactly as it does there
        np.savez(file, sid=sid, idx=idx, dom=[row[1] for row in table], dct=[_npy_vector(row[2]) for row in table])

    def save_doms(self, file: str):
        """``pid ndom d1;d2`` per protein, whole-protein entry dropped when there are several (:378-392)."""
        per_pid = {}
        for pid, dom in self.cur.execute('SELECT pid, domain FROM fingerprints'):
            per_pid.setdefault(pid, []).append(dom)
        lines = []
        for pid, names in per_pid.items():
            keep = names[:-1] if len(names) > 1 else names
            lines.append(f'{pid} {len(keep)} {";".join(keep)}\n')
        with open(file, 'w', encoding='utf8') as out:
            out.writelines(lines)


def write_flat_index(path: str, vectors: np.ndarray):
    """Serialises an ``IndexFlatL2`` the way faiss 1.7.x ``write_index`` does ("IxF2", header,
    codes as a byte vector counted in 4-byte units).  Restated from the published format; parity
    unpinned in this environment (faiss is not installed)."""
    x = np.ascontiguousarray(vectors, dtype=np.float32)
    n, d = x.shape
    with open(path, 'wb') as f:
        f.write(b'IxF2')
        f.write(struct.pack('<i', d))
        f.write(struct.pack('<q', n))
        f.write(struct.pack('<qq', 1 << 20, 1 << 20))
        f.write(struct.pack('<B', 1))          # is_trained
        f.write(struct.pack('<i', 1))          # METRIC_L2
        f.write(struct.pack('<Q', n * d))      # codes.size() / 4
        f.write(x.tobytes())


def read_flat_index(path: str) -> np.ndarray:
    with open(path, 'rb') as f:
        if f.read(4) != b'IxF2':
            raise ValueError('not an IndexFlatL2 file')
        d, = struct.unpack('<i', f.read(4))
        n, = struct.unpack('<q', f.read(8))
        f.read(16 + 1 + 4)
        cnt, = struct.unpack('<Q', f.read(8))
        return np.frombuffer(f.read(cnt * 4), dtype=np.float32).reshape(n, d)
