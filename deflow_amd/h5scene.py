"""Reader for the preprocessed scene files the reference trains on (``<scene>.h5``, written by OpenSceneFlow's
``dataprocess/extract_av2.py`` with h5py; [REF README.md:52], [REF assets/slurm/0_process.sh:17-35]) -- without h5py,
which this image does not have.

Only the part of the HDF5 file format those files use is implemented, straight from the format specification (HDF5 File
Format Specification version 2.0/3.0, the "version 0/1" on-disk structures h5py writes with its default
``libver='earliest'``):

  superblock v0 / v1 -> root symbol-table entry -> object headers v1 (with continuation blocks)
  groups  = symbol-table message -> v1 B-tree (node type 0, any depth) -> SNOD symbol nodes + local heap names
  dataset = dataspace v1 / v2, datatype (fixed point, IEEE float, enum over int8 = numpy bool), layout v3
            (contiguous, compact, or chunked through a v1 B-tree of node type 1) and the filter pipeline
            (gzip = deflate, shuffle; fletcher32 checksums are skipped)

Anything else (new-style groups, layout v4, other filters, other datatype classes) raises ``H5FormatError`` naming the
construct -- never a silent wrong read.  Pinned by ``tests/test_h5scene.py`` against fixtures written by real h5py 3.3.0 /
HDF5 1.10.6 (``tests/golden/gen_h5_fixtures.py``): every array must equal what h5py itself read back.
"""
from __future__ import annotations

import mmap
import zlib
from typing import List, Tuple

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
_SIG = b"\x89HDF\r\n\x1a\n"


class H5FormatError(RuntimeError):
    pass


class _Dataset:
    """Lazy handle: shape / dtype are known after the header parse; ``read()`` materialises a numpy array."""

    def __init__(self, f: "H5File", shape, dtype, layout, filters):
        self._f, self.shape, self.dtype, self._layout, self._filters = f, tuple(shape), dtype, layout, filters

    def read(self) -> np.ndarray:
        f, n = self._f, int(np.prod(self.shape, dtype=np.int64))
        kind = self._layout[0]
        if kind == "contiguous":
            _, addr, size = self._layout
            if n == 0 or addr == UNDEF:
                return np.zeros(self.shape, self.dtype)
            return np.frombuffer(f._mm, self.dtype, n, f._base + addr).reshape(self.shape).copy()
        if kind == "compact":
            return np.frombuffer(self._layout[1], self.dtype, n).reshape(self.shape).copy()
        _, btree, chunk = self._layout      # chunked
        out = np.zeros(self.shape, self.dtype)
        if n == 0 or btree == UNDEF:
            return out
        rank = len(self.shape)
        for offs, addr, nbytes, mask in f._chunks(btree, rank):
            raw = bytes(f._mm[f._base + addr: f._base + addr + nbytes])
            for k in range(len(self._filters) - 1, -1, -1):        # undo the pipeline back to front
                if mask >> k & 1:
                    continue                                          # this filter was skipped for this chunk
                fid = self._filters[k]
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    es = self.dtype.itemsize
                    raw = np.frombuffer(raw, np.uint8).reshape(es, -1).T.tobytes()
                elif fid == 3:
                    raw = raw[:-4]
                else:
                    raise H5FormatError(f"filter id {fid} is not supported")
            block = np.frombuffer(raw, self.dtype, int(np.prod(chunk))).reshape(chunk)
            sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, self.shape))
            sel_in = tuple(slice(0, s.stop - s.start) for s in sel_out)
            out[sel_out] = block[sel_in]
        return out

    def __getitem__(self, key):
        return self.read()[key]


class H5Group(dict):
    """name -> H5Group | dataset handle (iteration order = the file's B-tree order = sorted names, as in h5py)."""


class H5File:
    def __init__(self, path: str):
        self.path = path
        self._fh = open(path, "rb")
        self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        self._base = 0
        self.root = self._open()

    # ---- context manager / mapping sugar -------------------------------------------------------------------------
    def close(self):
        self._mm.close()
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __getitem__(self, name):
        node = self.root
        for part in str(name).strip("/").split("/"):
            node = node[part]
        return node

    def keys(self):
        return self.root.keys()

    # ---- primitives ----------------------------------------------------------------------------------------------
    def _u(self, off: int, size: int) -> int:
        return int.from_bytes(self._mm[off: off + size], "little")

    def _open(self) -> H5Group:
        mm = self._mm
        pos = 0
        while mm[pos: pos + 8] != _SIG:             # the superblock may sit at 0, 512, 1024, ...
            pos = 512 if pos == 0 else pos * 2
            if pos + 8 > len(mm):
                raise H5FormatError(f"{self.path}: not an HDF5 file")
        ver = mm[pos + 8]
        if ver not in (0, 1):
            raise H5FormatError(f"{self.path}: superblock version {ver} (written with libver='latest'?) is not supported")
        self._O, self._L = mm[pos + 13], mm[pos + 14]
        if (self._O, self._L) != (8, 8):
            raise H5FormatError(f"offset/length sizes {self._O}/{self._L} are not supported")
        p = pos + 24 + (4 if ver == 1 else 0)
        self._base = self._u(p, 8)
        ste = p + 32                                  # base, free-space, end-of-file, driver-info addresses
        obj = self._u(ste + 8, 8)
        return self._object(obj)

    # ---- object headers -------------------------------------------------------------------------------------------
    def _messages(self, addr: int) -> List[Tuple[int, bytes]]:
        mm, a = self._mm, self._base + addr
        if mm[a] != 1:
            raise H5FormatError(f"object header version {mm[a]} at {addr:#x} is not supported (only v1)")
        nmsg = self._u(a + 2, 2)
        size = self._u(a + 8, 4)
        blocks = [(a + 16, size)]
        out: List[Tuple[int, bytes]] = []
        while blocks and len(out) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize = self._u(p, 2), self._u(p + 2, 2)
                body = bytes(mm[p + 8: p + 8 + msize])
                p += 8 + msize
                if mtype == 0x0010:               # continuation: (offset, length)
                    blocks.append((self._base + int.from_bytes(body[:8], "little"), int.from_bytes(body[8:16], "little")))
                out.append((mtype, body))
        return out

    def _object(self, addr: int):
        msgs = self._messages(addr)
        types = {t for t, _ in msgs}
        if 0x0011 in types:
            body = next(b for t, b in msgs if t == 0x0011)
            btree, heap = int.from_bytes(body[:8], "little"), int.from_bytes(body[8:16], "little")
            g = H5Group()
            for name, child in self._group_entries(btree, heap):
                g[name] = self._object(child)
            return g
        if 0x0002 in types or 0x0006 in types:
            raise H5FormatError("new-style (link message) groups are not supported")
        if 0x0008 not in types:
            raise H5FormatError(f"object at {addr:#x} is neither an old-style group nor a dataset")
        shape = dtype = layout = None
        filters: List[int] = []
        for t, b in msgs:
            if t == 0x0001:
                shape = self._dataspace(b)
            elif t == 0x0003:
                dtype = self._datatype(b)
            elif t == 0x0008:
                layout = self._layout(b)
            elif t == 0x000B:
                filters = self._pipeline(b)
        if shape is None or dtype is None or layout is None:
            raise H5FormatError(f"dataset at {addr:#x} lacks a dataspace / datatype / layout message")
        return _Dataset(self, shape, dtype, layout, filters)

    # ---- groups ----------------------------------------------------------------------------------------------------
    def _heap_name(self, heap_data: int, off: int) -> str:
        a = heap_data + off
        e = self._mm.find(b"\0", a)
        return self._mm[a:e].decode("utf-8")

    def _group_entries(self, btree: int, heap: int):
        mm = self._mm
        h = self._base + heap
        if mm[h: h + 4] != b"HEAP":
            raise H5FormatError("bad local heap signature")
        heap_data = self._base + self._u(h + 24, 8)

        def walk(node):
            a = self._base + node
            if mm[a: a + 4] == b"SNOD":
                n = self._u(a + 6, 2)
                for i in range(n):
                    e = a + 8 + 40 * i
                    yield self._heap_name(heap_data, self._u(e, 8)), self._u(e + 8, 8)
                return
            if mm[a: a + 4] != b"TREE" or mm[a + 4] != 0:
                raise H5FormatError("bad group B-tree node")
            used = self._u(a + 6, 2)
            p = a + 24 + 8                       # past the header and key 0
            for _ in range(used):
                yield from walk(self._u(p, 8))
                p += 16                          # child address + next key
        yield from walk(btree)

    # ---- dataset messages ------------------------------------------------------------------------------------------
    @staticmethod
    def _dataspace(b: bytes):
        ver, rank, flags = b[0], b[1], b[2]
        if ver == 1:
            p = 8
        elif ver == 2:
            if b[3] == 2:
                raise H5FormatError("null dataspace")
            p = 4
        else:
            raise H5FormatError(f"dataspace version {ver}")
        return [int.from_bytes(b[p + 8 * i: p + 8 * i + 8], "little") for i in range(rank)]

    @classmethod
    def _datatype(cls, b: bytes) -> np.dtype:
        klass, bits0, size = b[0] & 15, b[1], int.from_bytes(b[4:8], "little")
        order = ">" if bits0 & 1 else "<"
        if klass == 0:
            return np.dtype(f"{order}{'i' if bits0 & 8 else 'u'}{size}")
        if klass == 1:
            if size not in (2, 4, 8):
                raise H5FormatError(f"float of {size} bytes")
            return np.dtype(f"{order}f{size}")
        if klass == 8:                         # enum: h5py stores numpy bool as ENUM{FALSE=0, TRUE=1} over int8
            base = cls._datatype(b[8:])
            nmemb = b[1] | b[2] << 8
            names = b[8 + cls._datatype_len(b[8:]):]
            if base.itemsize == 1 and nmemb == 2 and b"FALSE" in names and b"TRUE" in names:
                return np.dtype(np.bool_)
            return base
        raise H5FormatError(f"datatype class {klass} is not supported (fixed point, float and bool enums are)")

    @staticmethod
    def _datatype_len(b: bytes) -> int:
        klass = b[0] & 15
        return 8 + {0: 4, 1: 12}.get(klass, 0)

    def _layout(self, b: bytes):
        if b[0] != 3:
            raise H5FormatError(f"data layout message version {b[0]} is not supported (only v3)")
        klass = b[1]
        if klass == 1:
            return ("contiguous", int.from_bytes(b[2:10], "little"), int.from_bytes(b[10:18], "little"))
        if klass == 0:
            n = int.from_bytes(b[2:4], "little")
            return ("compact", bytes(b[4:4 + n]))
        if klass == 2:
            rank1 = b[2]
            btree = int.from_bytes(b[3:11], "little")
            dims = [int.from_bytes(b[11 + 4 * i: 15 + 4 * i], "little") for i in range(rank1)]
            return ("chunked", btree, tuple(dims[:-1]))
        raise H5FormatError(f"layout class {klass}")

    @staticmethod
    def _pipeline(b: bytes) -> List[int]:
        ver, n = b[0], b[1]
        p = 8 if ver == 1 else 2
        ids = []
        for _ in range(n):
            fid = int.from_bytes(b[p: p + 2], "little")
            if ver == 1 or fid >= 256:
                nlen = int.from_bytes(b[p + 2: p + 4], "little")
                p += 2
            else:
                nlen = 0
            ncli = int.from_bytes(b[p + 4: p + 6], "little")
            p += 6
            p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            p += 4 * ncli
            if ver == 1 and ncli % 2:
                p += 4
            ids.append(fid)
        return ids

    def _chunks(self, btree: int, rank: int):
        """(offsets[rank], address, stored bytes, filter mask) of every chunk under a type-1 B-tree."""
        mm = self._mm
        ksz = 8 + 8 * (rank + 1)

        def walk(node):
            a = self._base + node
            if mm[a: a + 4] != b"TREE" or mm[a + 4] != 1:
                raise H5FormatError("bad chunk B-tree node")
            level, used = mm[a + 5], self._u(a + 6, 2)
            p = a + 24
            for _ in range(used):
                nbytes, mask = self._u(p, 4), self._u(p + 4, 4)
                offs = [self._u(p + 8 + 8 * i, 8) for i in range(rank)]
                child = self._u(p + ksz, 8)
                if level == 0:
                    yield offs, child, nbytes, mask
                else:
                    yield from walk(child)
                p += ksz + 8
        yield from walk(btree)
