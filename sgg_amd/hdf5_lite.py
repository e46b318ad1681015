"""Read-only HDF5 for the two files the path consumes -- `VG-SGG.h5` (dataloaders/visual_genome.py:536-576) and `features.hdf5`
(extract_features.py:50-70, augment/gan.py:64) -- without h5py, which this image does not have.

Implements the part of the HDF5 file format (v1.8 spec) that h5py's default writer (`libver='earliest'`) produces:
  * superblock version 0 / 1, version-1 object headers with continuation blocks;
  * old-style groups: symbol-table message -> v1 B-tree (type 0) -> symbol-table nodes, names in the local heap;
  * datasets: dataspace v1 / v2, datatype classes fixed-point, floating-point (IEEE 2 / 4 / 8 bytes), fixed-length string;
    data layout v3 -- compact, contiguous, chunked (v1 B-tree type 1 chunk index, incl. extensible datasets);
  * filter pipeline v1 / v2: deflate (gzip), shuffle, fletcher32.
Anything else (superblock v2+, fractal-heap groups, variable-length types, virtual / external storage) raises
NotImplementedError naming the feature, never a wrong array.

    with File(path) as f:
        f.keys(); name in f
        d = f['boxes_1024']            # Dataset: .shape, .dtype, .chunks
        d[:], d[:, 0], d[mask], d[i], d[a:b], d[[3, 5, 8]]
Index expressions follow numpy on the full array; when the first index is an int / slice / index list and the dataset is chunked
with chunk length 1 along axis 0 (how the reference writes features.hdf5) only the chunks touched are read and inflated."""
import os
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
_SIG = b'\x89HDF\r\n\x1a\n'


class _Reader(object):
    def __init__(self, path):
        self.f = open(path, 'rb')
        self.size = os.fstat(self.f.fileno()).st_size
        self.base = 0

    def read(self, addr, n):
        self.f.seek(self.base + addr)
        b = self.f.read(n)
        if len(b) != n:
            raise IOError('hdf5_lite: short read at %d (+%d): file truncated?' % (addr, n))
        return b

    def close(self):
        self.f.close()


def _u(b, off, n):
    return int.from_bytes(b[off:off + n], 'little')


class File(object):
    def __init__(self, path, mode='r'):
        if mode != 'r':
            raise NotImplementedError('hdf5_lite is read-only')
        self.filename = str(path)
        self._r = _Reader(path)
        self._parse_superblock()
        self._root = Group(self, self._root_addr, '/')

    # ------------------------------------------------------------------ superblock (spec II.A)
    def _parse_superblock(self):
        r = self._r
        off = 0
        while True:                                    # the signature sits at 0 or at a power-of-two offset >= 512
            if off + 8 > r.size:
                raise IOError('hdf5_lite: %s is not an HDF5 file (no signature)' % self.filename)
            if r.read(off, 8) == _SIG:
                break
            off = 512 if off == 0 else off * 2
        head = r.read(off, 24)
        version = head[8]
        if version not in (0, 1):
            raise NotImplementedError('hdf5_lite: superblock version %d (written with libver="latest"?); only 0 / 1 are read' % version)
        self.size_of_offsets, self.size_of_lengths = head[13], head[14]
        if self.size_of_offsets != 8 or self.size_of_lengths != 8:
            raise NotImplementedError('hdf5_lite: %d-byte offsets / %d-byte lengths' % (self.size_of_offsets, self.size_of_lengths))
        p = off + 24 + (4 if version == 1 else 0)      # v1 adds the indexed-storage k + 2 reserved bytes
        body = r.read(p, 32 + 40)
        base = _u(body, 0, 8)
        r.base = base if base != UNDEF else 0
        # root group symbol-table entry: link name offset, object header address, cache type, reserved, scratch
        ent = body[32:]
        self._root_addr = _u(ent, 8, 8)

    # ------------------------------------------------------------------ mapping interface over the root group
    def __getitem__(self, name):
        return self._root[name]

    def __contains__(self, name):
        return name in self._root

    def keys(self):
        return self._root.keys()

    def __iter__(self):
        return iter(self._root.keys())

    def close(self):
        self._r.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


# ---------------------------------------------------------------------------------------------------- object headers (IV.A.1.a)
def _messages(r, addr):
    """-> [(type, flags, body bytes)] of a version-1 object header, continuation blocks followed."""
    pre = r.read(addr, 16)
    if pre[0] != 1:
        if pre[:4] == b'OHDR':
            raise NotImplementedError('hdf5_lite: version-2 object headers (file written with libver="latest")')
        raise IOError('hdf5_lite: object header version %d at %d' % (pre[0], addr))
    nmsg, hsize = _u(pre, 2, 2), _u(pre, 8, 4)
    blocks = [(addr + 16, hsize)]
    out = []
    while blocks and len(out) < nmsg:
        a, n = blocks.pop(0)
        buf = r.read(a, n)
        p = 0
        while p + 8 <= n and len(out) < nmsg:
            mtype, msize, flags = _u(buf, p, 2), _u(buf, p + 2, 2), buf[p + 4]
            body = buf[p + 8:p + 8 + msize]
            p += 8 + msize
            if mtype == 0x10:                           # continuation
                blocks.append((_u(body, 0, 8), _u(body, 8, 8)))
            out.append((mtype, flags, body))
    return out


def _pad8(n):
    return (n + 7) & ~7


def _dataspace(body):
    ver = body[0]
    rank, flags = body[1], body[2]
    if ver == 1:
        p = 8
    elif ver == 2:
        if body[3] == 2:
            raise NotImplementedError('hdf5_lite: null dataspace')
        p = 4
    else:
        raise NotImplementedError('hdf5_lite: dataspace message version %d' % ver)
    dims = tuple(_u(body, p + 8 * i, 8) for i in range(rank))
    return dims


def _datatype(body):
    """-> (numpy dtype, size in bytes)"""
    cls, bits0, size = body[0] & 0x0F, body[1], _u(body, 4, 4)
    order = '>' if bits0 & 1 else '<'
    if cls == 0:                                        # fixed-point
        signed = bool(bits0 & 0x08)
        if size not in (1, 2, 4, 8):
            raise NotImplementedError('hdf5_lite: %d-byte integers' % size)
        return np.dtype('%s%s%d' % (order, 'i' if signed else 'u', size)), size
    if cls == 1:                                        # floating point: IEEE layouts only (what numpy / h5py write)
        if size not in (2, 4, 8):
            raise NotImplementedError('hdf5_lite: %d-byte floats' % size)
        return np.dtype('%sf%d' % (order, size)), size
    if cls == 3:                                        # fixed-length string
        return np.dtype('S%d' % size), size
    if cls == 8:                                        # enumeration over an integer base type; h5py stores numpy bool this way
        base, bsize = _datatype(body[8:])
        nmem = bits0 | (body[2] << 8)
        names_at = 8 + 8 + (4 if (body[8] & 0x0F) == 0 else 12)      # base type header + its properties (fixed-point: 4 bytes)
        members, p = [], names_at
        for _ in range(nmem):
            end = body.index(b'\x00', p)
            members.append(bytes(body[p:end]))
            p += _pad8(end - p + 1)
        if members == [b'FALSE', b'TRUE'] and bsize == 1:
            return np.dtype(np.bool_), 1
        return base, bsize
    names = {2: 'time', 4: 'bitfield', 5: 'opaque', 6: 'compound', 7: 'reference', 8: 'enum', 9: 'variable-length', 10: 'array'}
    raise NotImplementedError('hdf5_lite: datatype class %s' % names.get(cls, cls))


def _filters(body):
    """-> [(filter id, client data values)] in pipeline order"""
    ver, n = body[0], body[1]
    p = 8 if ver == 1 else 2
    out = []
    for _ in range(n):
        fid = _u(body, p, 2)
        if ver == 1 or fid >= 256:
            name_len = _u(body, p + 2, 2)
            flags, ncd = _u(body, p + 4, 2), _u(body, p + 6, 2)
            p += 8 + (_pad8(name_len) if ver == 1 else name_len)
        else:
            flags, ncd = _u(body, p + 2, 2), _u(body, p + 4, 2)
            p += 6
        cd = [_u(body, p + 4 * i, 4) for i in range(ncd)]
        p += 4 * ncd
        if ver == 1 and ncd % 2:
            p += 4
        out.append((fid, cd))
    return out


# ---------------------------------------------------------------------------------------------------- groups (III.A, III.C, III.D)
class Group(object):
    def __init__(self, file, addr, name):
        self._file, self._addr, self.name = file, addr, name
        self._links = None

    def _load(self):
        if self._links is not None:
            return
        r = self._file._r
        links = {}
        for mtype, _, body in _messages(r, self._addr):
            if mtype == 0x11:                           # symbol table message: B-tree + local heap
                btree, heap = _u(body, 0, 8), _u(body, 8, 8)
                hb = r.read(heap, 32)
                if hb[:4] != b'HEAP':
                    raise IOError('hdf5_lite: bad local heap at %d' % heap)
                heap_data, heap_size = _u(hb, 24, 8), _u(hb, 8, 8)
                names = r.read(heap_data, heap_size)
                self._walk(btree, names, links)
            elif mtype in (0x02, 0x06):
                raise NotImplementedError('hdf5_lite: new-style groups (link messages; file written with libver="latest")')
        self._links = links

    def _walk(self, addr, names, links):
        r = self._file._r
        head = r.read(addr, 24)
        if head[:4] == b'TREE':
            if head[4] != 0:
                raise IOError('hdf5_lite: group B-tree node of type %d' % head[4])
            n = _u(head, 6, 2)
            body = r.read(addr + 24, (2 * n + 1) * 8)
            for i in range(n):
                self._walk(_u(body, 8 + 16 * i, 8), names, links)
            return
        if head[:4] != b'SNOD':
            raise IOError('hdf5_lite: expected a symbol-table node at %d' % addr)
        n = _u(head, 6, 2)
        ents = r.read(addr + 8, n * 40)
        for i in range(n):
            e = ents[40 * i:40 * (i + 1)]
            noff, oaddr = _u(e, 0, 8), _u(e, 8, 8)
            end = names.index(b'\x00', noff)
            links[names[noff:end].decode('utf-8')] = oaddr

    def keys(self):
        self._load()
        return list(self._links.keys())

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __iter__(self):
        return iter(self.keys())

    def __getitem__(self, name):
        parts = [p for p in name.split('/') if p]
        node = self
        for i, part in enumerate(parts):
            if not isinstance(node, Group):
                raise KeyError(name)
            node._load()
            if part not in node._links:
                raise KeyError(name)
            node = _open(node._file, node._links[part], node.name.rstrip('/') + '/' + part)
        return node


def _open(file, addr, name):
    msgs = _messages(file._r, addr)
    types = set(m[0] for m in msgs)
    if 0x08 in types:
        return Dataset(file, msgs, name)
    return Group(file, addr, name)


# ---------------------------------------------------------------------------------------------------- datasets
class Dataset(object):
    def __init__(self, file, msgs, name):
        self._file, self.name = file, name
        self.shape = self.dtype = None
        self._filters = []
        self._layout = None
        self.chunks = None
        for mtype, _, body in msgs:
            if mtype == 0x01:
                self.shape = _dataspace(body)
            elif mtype == 0x03:
                self.dtype, self._esize = _datatype(body)
            elif mtype == 0x0B:
                self._filters = _filters(body)
            elif mtype == 0x08:
                self._parse_layout(body)
        if self.shape is None or self.dtype is None or self._layout is None:
            raise IOError('hdf5_lite: %s lacks a dataspace / datatype / layout message' % name)
        for fid, _ in self._filters:
            if fid not in (1, 2, 3):
                raise NotImplementedError('hdf5_lite: filter %d (%s) on %s' % (fid, {4: 'szip', 5: 'nbit', 6: 'scaleoffset', 32000: 'lzf'}.get(fid, '?'), name))

    def _parse_layout(self, body):
        ver = body[0]
        if ver != 3:
            raise NotImplementedError('hdf5_lite: data layout message version %d' % ver)
        cls = body[1]
        if cls == 0:
            n = _u(body, 2, 2)
            self._layout = ('compact', bytes(body[4:4 + n]))
        elif cls == 1:
            self._layout = ('contiguous', _u(body, 2, 8), _u(body, 10, 8))
        elif cls == 2:
            rank = body[2]                                # dataset rank + 1 (the element size is the last "dimension")
            btree = _u(body, 3, 8)
            dims = tuple(_u(body, 11 + 4 * i, 4) for i in range(rank))
            self.chunks = dims[:-1]
            self._layout = ('chunked', btree)
        else:
            raise NotImplementedError('hdf5_lite: data layout class %d (virtual?)' % cls)

    def __len__(self):
        return self.shape[0]

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    # ---- chunk index: v1 B-tree type 1 (III.A.1)
    def _chunk_table(self):
        if getattr(self, '_chunk_map', None) is not None:
            return self._chunk_map
        r = self._file._r
        rank = len(self.shape)
        out = {}

        def walk(addr):
            head = r.read(addr, 24)
            if head[:4] != b'TREE' or head[4] != 1:
                raise IOError('hdf5_lite: bad chunk B-tree node at %d' % addr)
            level, n = head[5], _u(head, 6, 2)
            ksz = 8 + 8 * (rank + 1)
            body = r.read(addr + 24, n * (ksz + 8) + ksz)
            for i in range(n):
                k = body[i * (ksz + 8):i * (ksz + 8) + ksz]
                child = _u(body, i * (ksz + 8) + ksz, 8)
                if level > 0:
                    walk(child)
                else:
                    nbytes, mask = _u(k, 0, 4), _u(k, 4, 4)
                    off = tuple(_u(k, 8 + 8 * d, 8) for d in range(rank))
                    out[off] = (child, nbytes, mask)
        kind, btree = self._layout
        if btree != UNDEF:
            walk(btree)
        self._chunk_map = out
        return out

    def _decode_chunk(self, raw, mask):
        n = len(self._filters)
        for i in range(n - 1, -1, -1):                    # the pipeline is undone last filter first
            if mask & (1 << i):
                continue
            fid, cd = self._filters[i]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = cd[0] if cd else self._esize
                a = np.frombuffer(raw, dtype=np.uint8)
                m = len(raw) // es
                raw = a[:m * es].reshape(es, m).T.tobytes() + bytes(a[m * es:])
            elif fid == 3:
                raw = raw[:-4]
        return raw

    def _read_chunk(self, off):
        ent = self._chunk_table().get(off)
        if ent is None:
            return np.zeros(self.chunks, dtype=self.dtype)      # never written: fill value 0 (h5py default)
        addr, nbytes, mask = ent
        raw = self._decode_chunk(self._file._r.read(addr, nbytes), mask)
        return np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(self.chunks))).reshape(self.chunks)

    def _read_all(self):
        kind = self._layout[0]
        if self.size == 0:
            return np.zeros(self.shape, dtype=self.dtype)
        if kind == 'compact':
            return np.frombuffer(self._layout[1], dtype=self.dtype, count=self.size).reshape(self.shape).copy()
        if kind == 'contiguous':
            addr = self._layout[1]
            if addr == UNDEF:
                return np.zeros(self.shape, dtype=self.dtype)
            raw = self._file._r.read(addr, self.size * self._esize)
            return np.frombuffer(raw, dtype=self.dtype).reshape(self.shape).copy()
        out = np.zeros(self.shape, dtype=self.dtype)
        for off in self._chunk_table():
            if any(o >= s for o, s in zip(off, self.shape)):
                continue                                  # a chunk left behind by a shrink
            blk = self._read_chunk(off)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(off, self.chunks, self.shape))
            out[sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out

    def _rows(self, idx):
        """rows `idx` (list of ints) of a dataset chunked one row per chunk along axis 0: only the chunks touched are read"""
        out = np.zeros((len(idx),) + tuple(self.shape[1:]), dtype=self.dtype)
        grid = [range(0, s, c) for s, c in zip(self.shape[1:], self.chunks[1:])]
        for k, i in enumerate(idx):
            for off in np.ndindex(*[len(g) for g in grid]) if grid else [()]:
                o = (i,) + tuple(g[j] for g, j in zip(grid, off))
                blk = self._read_chunk(o)[0]
                sl = tuple(slice(a, min(a + c, s)) for a, c, s in zip(o[1:], self.chunks[1:], self.shape[1:]))
                out[(k,) + sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out

    def __getitem__(self, key):
        if isinstance(key, tuple) and len(key) == 0:
            return self._read_all()[()]
        first = key[0] if isinstance(key, tuple) else key
        rest = key[1:] if isinstance(key, tuple) else ()
        chunk_rows = self._layout[0] == 'chunked' and self.chunks and self.chunks[0] == 1 and len(self.shape) >= 1
        if chunk_rows and first is not Ellipsis and not (isinstance(first, slice) and first == slice(None)):
            n = self.shape[0]
            if isinstance(first, (int, np.integer)):
                i = int(first) + (n if first < 0 else 0)
                if not 0 <= i < n:
                    raise IndexError('index %d out of range for axis 0 with size %d' % (first, n))
                row = self._rows([i])[0]
                return row[rest] if rest else row
            if isinstance(first, slice):
                idx = list(range(*first.indices(n)))
            else:
                a = np.asarray(first)
                if a.dtype == bool:
                    idx = np.nonzero(a)[0].tolist()
                elif a.dtype.kind in 'iu' and a.ndim == 1:
                    idx = [int(v) + (n if v < 0 else 0) for v in a]
                else:
                    idx = None
            if idx is not None:
                rows = self._rows(idx)
                return rows[(slice(None),) + tuple(rest)] if rest else rows
        return self._read_all()[key]

    def __array__(self, dtype=None, copy=None):
        a = self._read_all()
        return a.astype(dtype) if dtype is not None else a
