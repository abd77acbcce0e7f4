"""``Database`` -- the output side of the fingerprint path: SQLite ``.db``, ``-dct.npz``, ``.dom``
and (optionally) the FAISS ``.index``, byte-compatible with mgtools/DCTdomain
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

Differences, on purpose: inserts are batched in one transaction per batch of proteins
(``add_fprints``; the reference commits per protein), and ``yield_seqs`` does not lose the tail
of the input in multi-sequence mode (reference bug at :166-178)."""

from __future__ import annotations

import datetime
import os
import sqlite3
import struct
from io import BytesIO
from typing import Iterable, List, Sequence, Tuple

import numpy as np


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


class Database:
    """SQLite database of sequences and fingerprints (reference ``Database``, :16-392)."""

    def __init__(self, dbfile: str, fafile: str = None):
        if fafile and (fafile.endswith('.fa') or fafile.endswith('.fasta')):
            print(f'Reading file: {fafile}')
            self.path = dbfile
            self.init_db(self.read_fasta(fafile))
        else:
            if not os.path.exists(dbfile):
                raise FileNotFoundError(f'Database file not found: {dbfile}')
            print(f'Opening database: {dbfile}')
            self.path = os.path.splitext(dbfile)[0]
            self.conn = sqlite3.connect(f'{self.path}.db')
            self.cur = self.conn.cursor()

    def close(self):
        print(f'Closing database: {self.path}\n')
        self.conn.close()

    # -- input ---------------------------------------------------------------------------
    def read_fasta(self, fafile: str) -> dict:
        """pid -> sequence, ascending length (stable), as :60-81."""
        seqs = {}
        pid = None
        with open(fafile, 'r', encoding='utf8') as f:
            for line in f:
                if line.startswith('>'):
                    pid = line.strip().split()[0][1:]
                    seqs[pid] = ''
                else:
                    seqs[pid] += line.strip()
        return dict(sorted(seqs.items(), key=lambda item: len(item[1])))

    def init_db(self, seqs: dict):
        self.path = os.path.splitext(self.path)[0]
        self.conn = sqlite3.connect(f'{self.path}.db')
        self.cur = self.conn.cursor()
        self.cur.execute("""CREATE TABLE IF NOT EXISTS sequences (
                pid text PRIMARY KEY,
                sequence text NOT NULL,
                length integer NOT NULL,
                fpcount integer NOT NULL
                ); """)
        self.cur.execute("""CREATE TABLE IF NOT EXISTS fingerprints (
                vid integer PRIMARY KEY,
                domain text NOT NULL,
                fingerprint blob NOT NULL,
                pid text NOT NULL,
                FOREIGN KEY(pid) REFERENCES sequences(pid)
                ); """)
        self.cur.execute("""CREATE TABLE IF NOT EXISTS metadata (
                datetime text PRIMARY KEY,
                seq_num integer NOT NULL,
                avg_len real NOT NULL,
                fp_num integer NOT NULL,
                seqs_fp string NOT NULL
                ); """)
        self.cur.executemany(""" INSERT OR IGNORE INTO sequences(pid, sequence, length, fpcount)
            VALUES(?, ?, ?, ?) """, [(pid, seq, len(seq), 0) for pid, seq in seqs.items()])
        self.conn.commit()

    def pending(self, dim1: int = 3, dim2: int = 80) -> List[Tuple[str, str]]:
        """(pid, sequence) of every protein still to fingerprint (``fpcount = 0``), in table order,
        without those too short to quantise (``(length-2)*dim2 < dim1*dim2``, :149-155)."""
        rows = self.cur.execute(""" SELECT pid, sequence, length FROM sequences WHERE fpcount = 0 """).fetchall()
        return [(pid, seq) for pid, seq, length in rows if (length - 2) * dim2 >= dim1 * dim2]

    def yield_seqs(self, maxlen: int, cpu: int, dim1: int = 3, dim2: int = 80):
        """Batches of (pid, sequence): proteins are packed while their total length stays within
        ``maxlen`` and their number within ``cpu + 1``; a protein longer than ``maxlen`` travels alone
        (and is embedded in windows); ``maxlen = 1`` gives one protein per batch, the reference's CPU
        mode (make_db.py:136).  Same intent as :136-178, but every pending protein is yielded exactly
        once (the reference drops the tail of the input when its last batch holds several)."""
        batch, cur = [], 0
        any_seq = False
        for pid, seq in self.pending(dim1, dim2):
            any_seq = True
            if batch and (cur + len(seq) > maxlen or len(batch) > cpu):
                yield batch
                batch, cur = [], 0
            batch.append((pid, seq))
            cur += len(seq)
        if batch:
            yield batch
        if not any_seq:
            print('No sequences to fingerprint!\n')

    def get_last_vid(self) -> int:
        row = self.cur.execute(""" SELECT vid FROM fingerprints ORDER BY vid DESC LIMIT 1 """).fetchone()
        return row[0] + 1 if row else 1

    # -- output ---------------------------------------------------------------------------
    def add_fprint(self, fp, lock=None, counter=None):
        """One protein (reference signature, :197-224).  ``counter`` is any object with ``.value``."""
        self.add_fprints([fp], lock, counter)

    def add_fprints(self, fps: Iterable, lock=None, counter=None):
        """Many proteins in one transaction.  Each item needs ``pid``, ``domains`` and
        ``quants[dom]`` (0..127 ints) -- a ``Fingerprint`` or anything shaped like one."""
        class _Ctr:
            value = None
        if counter is None:
            counter = _Ctr()
            counter.value = self.get_last_vid()
        rows, updates = [], []
        for fp in fps:
            quants = np.array([fp.quants[dom] for dom in fp.domains], dtype=np.int8)
            updates.append((len(fp.domains), fp.pid))
            for dom, quant in zip(fp.domains, quants):
                if lock is not None:
                    with lock:
                        counter.value += 1
                        vid = counter.value
                else:
                    counter.value += 1
                    vid = counter.value
                rows.append((vid, dom, _npy_bytes(quant), fp.pid))
        self.cur.executemany(""" UPDATE sequences SET fpcount = ? WHERE pid = ? """, updates)
        self.cur.executemany(""" INSERT INTO fingerprints(vid, domain, fingerprint, pid)
            VALUES(?, ?, ?, ?) """, rows)
        self.conn.commit()

    def _all_fprints(self) -> np.ndarray:
        fps = [np.load(BytesIO(row[0]), allow_pickle=True)
               for row in self.cur.execute(""" SELECT fingerprint FROM fingerprints """)]
        return np.array(fps, dtype=np.int8)

    def create_index(self):
        """``faiss.IndexFlatL2`` over all fingerprints -> ``<path>.index`` (:227-243).  Uses faiss when
        it is installed; otherwise writes the same flat-index file with ``write_flat_index``
        (format restated from faiss 1.7.4's index_write.cpp -- parity unpinned: no faiss here to
        read it back)."""
        fps = self._all_fprints()
        try:
            import faiss
        except ImportError:
            write_flat_index(f'{self.path}.index', fps.astype(np.float32))
            return
        index = faiss.IndexFlatL2(fps.shape[1])
        index.add(fps.astype(np.float32))
        faiss.write_index(index, f'{self.path}.index')

    def load_fprints(self, pid: str = '') -> list:
        self.cur.execute(""" SELECT vid, fingerprint FROM fingerprints WHERE pid = ? """, (pid,))
        return [(row[0], np.load(BytesIO(row[1]), allow_pickle=True)) for row in self.cur]

    def rename_vid(self):
        """vids 1..N in table order (:268-282), in one pass."""
        vids = [v[0] for v in self.cur.execute(""" SELECT vid FROM fingerprints """).fetchall()]
        if vids != list(range(1, len(vids) + 1)):
            # two-step renumbering keeps the PRIMARY KEY unique while rows move
            self.cur.executemany(""" UPDATE fingerprints SET vid = ? WHERE vid = ? """,
                                 [(-(i + 1), v) for i, v in enumerate(vids)])
            self.cur.execute(""" UPDATE fingerprints SET vid = -vid """)
        self.conn.commit()

    def update_metadata(self):
        print('Updating metadata...')
        num_seqs = self.cur.execute(""" SELECT COUNT(*) FROM sequences """).fetchone()[0]
        avg_len = self.cur.execute(""" SELECT AVG(length) FROM sequences """).fetchone()[0]
        nom_dom, dom_seqs = self.cur.execute(
            """ SELECT SUM(fpcount), COUNT(*) FROM sequences WHERE fpcount > 0 """).fetchone()
        if not nom_dom:
            nom_dom = 0
        date = datetime.datetime.now().strftime('%Y-%m-%d %H:%M:%S')
        self.cur.execute(""" INSERT OR REPLACE INTO metadata(datetime, seq_num, avg_len, fp_num, seqs_fp)
            VALUES(?, ?, ?, ?, ?) """, (date, num_seqs, avg_len, nom_dom, f'{dom_seqs}/{num_seqs}'))
        self.conn.commit()
        self.db_info()

    def db_info(self):
        metadata = self.cur.execute(""" SELECT * FROM metadata ORDER BY datetime DESC LIMIT 1 """).fetchone()
        if metadata is None:
            self.update_metadata()
            return
        print(f'Last Updated: {metadata[0]}')
        print(f'Number of Sequences: {metadata[1]}')
        print(f'Average Sequence Length: {metadata[2]:.2f}')
        print(f'Number of Fingerprints: {metadata[3]} ({metadata[4]} fingerprinted)\n')

    def seq_info(self, seq: str):
        print(f'Protein ID: {seq}')
        row = self.cur.execute(""" SELECT sequence FROM sequences WHERE pid = ? """, (seq,)).fetchone()
        if row is None:
            print('Sequence not found in database\n')
            return
        domains = self.cur.execute(""" SELECT domain FROM fingerprints WHERE pid = ? """, (seq,)).fetchall()
        print(f'Sequence: {row[0]}')
        if domains:
            print(f'Domains: {", ".join([dom[0] for dom in domains])}\n')
        else:
            print('No domains in database\n')

    def save_fprints(self, file: str):
        """``np.savez(file, sid=, idx=, dom=, dct=)`` (:351-375)."""
        seqs, idxs, doms, fps = [], [], [], []
        seq, idx = '', 0
        for pid, dom, blob in self.cur.execute(""" SELECT pid, domain, fingerprint FROM fingerprints """):
            if pid != seq:
                seq = pid
                seqs.append(seq)
                idxs.append(idx)
            doms.append(dom)
            fps.append(np.load(BytesIO(blob), allow_pickle=True))
            idx += 1
        idxs.append(idx)
        np.savez(file, sid=seqs, idx=idxs, dom=doms, dct=fps)

    def save_doms(self, file: str):
        """``pid ndom d1;d2`` per protein, whole-protein entry dropped when there are several (:378-392)."""
        doms = {}
        for pid, dom in self.cur.execute(""" SELECT pid, domain FROM fingerprints """):
            doms.setdefault(pid, []).append(dom)
        with open(file, 'w', encoding='utf8') as f:
            for pid, domains in doms.items():
                if len(domains) > 1:
                    domains = domains[:-1]
                f.write(f'{pid} {len(domains)} {";".join(domains)}\n')


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
