"""Minimal pure-Python HDF5 reader / writer for the MVP files.

The reference reads its datasets and writes its submission with h5py
(completion/dataset.py:21-34: `incomplete_pcds`, `complete_pcds`, `labels`;
completion/test.py:57-61: `results.h5` / dataset `results`).  h5py is not part
of this image, and the format subset those files use is small, so the two ends
of the loop are covered here without the dependency:

reader  -- superblock v0/v1 (symbol-table groups: B-tree v1 + local heap) and
           v2/v3 (object header v2, compact link messages); object headers v1
           (with continuation blocks) and v2; fixed-point and IEEE floating
           point datatypes of either byte order; simple dataspaces; layouts
           compact / contiguous / chunked (B-tree v1 chunk index) with the
           deflate and shuffle filters.  That is everything h5py writes for
           plain numeric arrays with or without `compression="gzip"`.
writer  -- what `h5py.File(p, "w").create_dataset(name, data=array)` produces in
           its default mode: superblock v0, one root group with a symbol table,
           one contiguous little-endian dataset per array.

`File` mimics the sliver of the h5py API the harness touches (`f[name][()]`,
`f.create_dataset(name, data=...)`, context manager, `keys()`).  Files written
here are checked against the real library (`h5dump`, where the image has it) in
tests/test_h5lite.py, and the reader against fixtures the real library wrote.
"""
import mmap
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(IOError):
    pass


# ------------------------------------------------------------------- reader
class _Reader:
    def __init__(self, buf):
        self.buf = buf
        if buf[:8] != SIGNATURE:
            raise H5Error("not an HDF5 file (no signature at offset 0)")
        ver = buf[8]
        if ver in (0, 1):
            self.so, self.sl = buf[13], buf[14]
            pos = 24 + (4 if ver == 1 else 0)
            self.base = self._off(pos)
            root_entry = pos + 4 * self.so
            self.root = self._off(root_entry + self.so)
        elif ver in (2, 3):
            self.so, self.sl = buf[9], buf[10]
            self.base = self._off(12)
            self.root = self._off(12 + 3 * self.so)
        else:
            raise H5Error("unsupported superblock version %d" % ver)
        if self.so != 8 or self.sl != 8:
            raise H5Error("only 8-byte offsets / lengths are supported")

    def _off(self, pos):
        return struct.unpack_from("<Q", self.buf, pos)[0]

    # -- object headers -> list of (type, flags, payload bytes)
    def messages(self, addr):
        buf = self.buf
        addr += self.base
        if buf[addr:addr + 4] == b"OHDR":
            return self._messages_v2(addr)
        if buf[addr] != 1:
            raise H5Error("unsupported object header version %d" % buf[addr])
        nmsg, = struct.unpack_from("<H", buf, addr + 2)
        size, = struct.unpack_from("<I", buf, addr + 8)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, mflags = struct.unpack_from("<HHB", buf, pos)
                data = bytes(buf[pos + 8:pos + 8 + msize])
                pos += 8 + msize
                if mtype == 0x0010:                      # continuation
                    off, ln = struct.unpack_from("<QQ", data)
                    blocks.append((off + self.base, ln))
                out.append((mtype, mflags, data))
        return out

    def _messages_v2(self, addr):
        buf = self.buf
        flags = buf[addr + 5]
        pos = addr + 6
        if flags & 0x20:
            pos += 16
        if flags & 0x10:
            pos += 4
        nbytes = 1 << (flags & 3)
        size = int.from_bytes(buf[pos:pos + nbytes], "little")
        pos += nbytes
        blocks = [(pos, size)]
        out = []
        while blocks:
            pos, left = blocks.pop(0)
            end = pos + left
            hdr = 4 + (2 if flags & 4 else 0)
            while pos + hdr <= end:
                mtype = buf[pos]
                msize, mflags = struct.unpack_from("<HB", buf, pos + 1)
                data = bytes(buf[pos + hdr:pos + hdr + msize])
                pos += hdr + msize
                if mtype == 0x10:
                    off, ln = struct.unpack_from("<QQ", data)
                    # continuation block: "OCHK" + messages + checksum
                    blocks.append((off + self.base + 4, ln - 8))
                out.append((mtype, mflags, data))
        return out

    # -- groups
    def links(self, addr):
        """name -> object header address of a group."""
        out = {}
        for mtype, _, data in self.messages(addr):
            if mtype == 0x0011:                          # symbol table
                btree, heap = struct.unpack_from("<QQ", data)
                self._walk_group_btree(btree + self.base, self._heap_data(heap + self.base), out)
            elif mtype == 0x0006:                        # link message
                name, target = self._link_message(data)
                if target is not None:
                    out[name] = target
            elif mtype == 0x0002:                        # link info: dense storage?
                flags = data[1]
                pos = 2 + (8 if flags & 1 else 0)
                fheap = struct.unpack_from("<Q", data, pos)[0]
                if fheap != UNDEF:
                    raise H5Error("dense link storage (fractal heap) is not supported")
        return out

    def _heap_data(self, addr):
        if self.buf[addr:addr + 4] != b"HEAP":
            raise H5Error("bad local heap signature")
        size, _free, data = struct.unpack_from("<QQQ", self.buf, addr + 8)
        return bytes(self.buf[data + self.base:data + self.base + size])

    def _walk_group_btree(self, addr, heap, out):
        buf = self.buf
        if buf[addr:addr + 4] != b"TREE":
            raise H5Error("bad B-tree signature")
        level, used = buf[addr + 5], struct.unpack_from("<H", buf, addr + 6)[0]
        pos = addr + 24
        for i in range(used):
            child = self._off(pos + 8 + i * 16)
            if level:
                self._walk_group_btree(child + self.base, heap, out)
                continue
            node = child + self.base
            if buf[node:node + 4] != b"SNOD":
                raise H5Error("bad symbol table node signature")
            nsym, = struct.unpack_from("<H", buf, node + 6)
            for s in range(nsym):
                e = node + 8 + s * 40
                name_off, ohdr = struct.unpack_from("<QQ", buf, e)
                out[heap[name_off:heap.index(b"\0", name_off)].decode()] = ohdr

    @staticmethod
    def _link_message(data):
        flags = data[1]
        pos = 2
        ltype = 0
        if flags & 0x08:
            ltype = data[pos]
            pos += 1
        if flags & 0x04:
            pos += 8
        if flags & 0x10:
            pos += 1
        n = 1 << (flags & 3)
        ln = int.from_bytes(data[pos:pos + n], "little")
        pos += n
        name = data[pos:pos + ln].decode()
        pos += ln
        if ltype != 0:
            return name, None                            # soft / external links are ignored
        return name, struct.unpack_from("<Q", data, pos)[0]

    # -- datasets
    def dataset(self, addr):
        shape = dtype = layout = None
        filters = []
        for mtype, _, data in self.messages(addr):
            if mtype == 0x0001:
                shape = self._dataspace(data)
            elif mtype == 0x0003:
                dtype = self._datatype(data)
            elif mtype == 0x0008:
                layout = data
            elif mtype == 0x000B:
                filters = self._filters(data)
        if shape is None or dtype is None or layout is None:
            raise H5Error("object is not a dataset")
        return self._read_layout(layout, shape, dtype, filters)

    @staticmethod
    def _dataspace(data):
        ver, rank, flags = data[0], data[1], data[2]
        if ver == 1:
            pos = 8
        elif ver == 2:
            pos = 4
            if data[3] == 2:
                raise H5Error("null dataspace")
        else:
            raise H5Error("unsupported dataspace version %d" % ver)
        return tuple(struct.unpack_from("<%dQ" % rank, data, pos)) if rank else ()

    @staticmethod
    def _datatype(data):
        cls, ver = data[0] & 0x0F, data[0] >> 4
        bits0 = data[1]
        size, = struct.unpack_from("<I", data, 4)
        order = ">" if bits0 & 1 else "<"
        if cls == 0:
            kind = "i" if bits0 & 0x08 else "u"
        elif cls == 1:
            kind = "f"
            if size not in (2, 4, 8):
                raise H5Error("unsupported float size %d" % size)
        else:
            raise H5Error("unsupported datatype class %d (version %d)" % (cls, ver))
        return np.dtype("%s%s%d" % (order if size > 1 else "|", kind, size))

    @staticmethod
    def _filters(data):
        ver, n = data[0], data[1]
        pos = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid, = struct.unpack_from("<H", data, pos)
            if ver == 1 or fid >= 256:
                name_len, = struct.unpack_from("<H", data, pos + 2)
                pos += 4
            else:
                name_len = 0
                pos += 2
            _flags, ncd = struct.unpack_from("<HH", data, pos)
            pos += 4
            if ver == 1:
                name_len = (name_len + 7) // 8 * 8
            pos += name_len
            cd = struct.unpack_from("<%dI" % ncd, data, pos)
            pos += 4 * ncd
            if ver == 1 and ncd % 2:
                pos += 4
            out.append((fid, cd))
        return out

    def _read_layout(self, data, shape, dtype, filters):
        ver, cls = data[0], data[1]
        if ver not in (3, 4) or (ver == 4 and cls == 2):
            # v4 (libver=latest) encodes compact / contiguous like v3; its chunk indexes
            # (fixed / extensible array, B-tree v2) are not implemented
            raise H5Error("unsupported data layout version %d class %d" % (ver, cls))
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if cls == 0:                                     # compact
            size, = struct.unpack_from("<H", data, 2)
            raw = data[4:4 + size]
            return np.frombuffer(raw, dtype=dtype, count=count).reshape(shape).copy()
        if cls == 1:                                     # contiguous
            addr, size = struct.unpack_from("<QQ", data, 2)
            if addr == UNDEF:                            # never written: fill value 0
                return np.zeros(shape, dtype=dtype)
            return np.frombuffer(self.buf, dtype=dtype, count=count, offset=addr + self.base).reshape(shape).copy()
        if cls == 2:                                     # chunked, B-tree v1 index
            rank = data[2]
            addr, = struct.unpack_from("<Q", data, 3)
            dims = struct.unpack_from("<%dI" % rank, data, 11)
            chunk = dims[:-1]                            # last entry = element size
            out = np.zeros(shape, dtype=dtype)
            if addr != UNDEF:
                self._walk_chunk_btree(addr + self.base, rank - 1, chunk, out, dtype, filters)
            return out
        raise H5Error("unsupported data layout class %d" % cls)

    def _walk_chunk_btree(self, addr, rank, chunk, out, dtype, filters):
        buf = self.buf
        if buf[addr:addr + 4] != b"TREE" or buf[addr + 4] != 1:
            raise H5Error("bad chunk B-tree node")
        level, used = buf[addr + 5], struct.unpack_from("<H", buf, addr + 6)[0]
        keysize = 8 + 8 * (rank + 1)
        pos = addr + 24
        for i in range(used):
            k = pos + i * (keysize + 8)
            csize, cmask = struct.unpack_from("<II", buf, k)
            offs = struct.unpack_from("<%dQ" % rank, buf, k + 8)
            child = self._off(k + keysize) + self.base
            if level:
                self._walk_chunk_btree(child, rank, chunk, out, dtype, filters)
                continue
            raw = bytes(buf[child:child + csize])
            for n, (fid, cd) in reversed(list(enumerate(filters))):
                if cmask & (1 << n):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    es = cd[0] if cd else dtype.itemsize
                    a = np.frombuffer(raw, dtype=np.uint8)
                    m = len(a) // es
                    raw = a[:m * es].reshape(es, m).T.tobytes() + a[m * es:].tobytes()
                else:
                    raise H5Error("unsupported filter id %d" % fid)
            block = np.frombuffer(raw, dtype=dtype, count=int(np.prod(chunk))).reshape(chunk)
            sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, out.shape))
            sel_in = tuple(slice(0, s.stop - s.start) for s in sel_out)
            out[sel_out] = block[sel_in]


# ------------------------------------------------------------------- writer
def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _dtype_message(dt):
    dt = np.dtype(dt)
    if dt.byteorder == ">":
        raise H5Error("big-endian arrays are written after .astype('<...')")
    if dt.kind == "f" and dt.itemsize in (4, 8):
        exp, man, bias = ((8, 23, 127), (11, 52, 1023))[dt.itemsize == 8]
        head = struct.pack("<BBBBI", 0x11, 0x20, dt.itemsize * 8 - 1, 0, dt.itemsize)
        return head + struct.pack("<HHBBBBI", 0, dt.itemsize * 8, man, exp, 0, man, bias)
    if dt.kind in "iu":
        head = struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0, 0, 0, dt.itemsize)
        return head + struct.pack("<HH", 0, dt.itemsize * 8)
    raise H5Error("unsupported dtype %s" % dt)


def _object_header(messages):
    body = b"".join(messages)
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body


def write_file(path, arrays):
    """arrays: dict name -> ndarray; one contiguous dataset per entry under the
    root group (the layout h5py's create_dataset(name, data=a) produces)."""
    names = sorted(arrays)                               # symbol table nodes are sorted by name
    if len(names) > 8:
        raise H5Error("at most 8 datasets per file (one symbol table node)")
    arrays = {k: np.ascontiguousarray(arrays[k]) for k in names}
    for k, a in arrays.items():
        if a.dtype.byteorder == ">":
            arrays[k] = a.astype(a.dtype.newbyteorder("<"))

    # local heap data segment: "" at offset 0, then the names, then one free block
    heap = bytearray(8)
    name_off = {}
    for k in names:
        name_off[k] = len(heap)
        heap += _pad8(k.encode() + b"\0")
    free_off = len(heap)
    heap += struct.pack("<QQ", 1, 16)                    # free block: next = H5HL_FREE_NULL, size 16

    pos = 96                                             # superblock v0 with 8-byte offsets
    root_ohdr = pos
    pos += 16 + 8 + 16                                   # prefix + one symbol-table message
    btree = pos
    pos += 24 + 32 * 8 + 33 * 8                          # group node, internal K = 16
    heap_hdr = pos
    pos += 32
    heap_data = pos
    pos += len(heap)
    snod = pos
    pos += 8 + 8 * 40                                    # leaf K = 4 -> 8 entries

    ohdr_at, data_at, headers = {}, {}, {}
    for k in names:
        a = arrays[k]
        space = struct.pack("<BBB5x", 1, a.ndim, 0) + struct.pack("<%dQ" % a.ndim, *a.shape)
        headers[k] = [_msg(0x0001, space), _msg(0x0003, _dtype_message(a.dtype), flags=1),
                      _msg(0x0005, struct.pack("<BBBB", 2, 1, 2, 0)), None]
        ohdr_at[k] = pos
        pos += 16 + sum(len(m) for m in headers[k][:3]) + 8 + 24
    for k in names:
        pos = (pos + 7) // 8 * 8
        data_at[k] = pos
        pos += arrays[k].nbytes
    eof = pos

    out = bytearray(eof)
    out[0:8] = SIGNATURE
    struct.pack_into("<BBBBBBBBHHI", out, 8, 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
    struct.pack_into("<QQQQ", out, 24, 0, UNDEF, eof, UNDEF)
    struct.pack_into("<QQII QQ", out, 56, 0, root_ohdr, 1, 0, btree, heap_hdr)
    hdr = _object_header([_msg(0x0011, struct.pack("<QQ", btree, heap_hdr))])
    out[root_ohdr:root_ohdr + len(hdr)] = hdr
    out[btree:btree + 8] = b"TREE" + struct.pack("<BBH", 0, 0, 1 if names else 0)
    struct.pack_into("<QQ", out, btree + 8, UNDEF, UNDEF)
    if names:
        struct.pack_into("<QQQ", out, btree + 24, 0, snod, name_off[names[-1]])
    out[heap_hdr:heap_hdr + 8] = b"HEAP\0\0\0\0"
    struct.pack_into("<QQQ", out, heap_hdr + 8, len(heap), free_off, heap_data)
    out[heap_data:heap_data + len(heap)] = heap
    out[snod:snod + 8] = b"SNOD" + struct.pack("<BBH", 1, 0, len(names))
    for i, k in enumerate(names):
        struct.pack_into("<QQII", out, snod + 8 + i * 40, name_off[k], ohdr_at[k], 0, 0)
        a = arrays[k]
        headers[k][3] = _msg(0x0008, struct.pack("<BBQQ", 3, 1, data_at[k], a.nbytes))
        hdr = _object_header(headers[k])
        out[ohdr_at[k]:ohdr_at[k] + len(hdr)] = hdr
        out[data_at[k]:data_at[k] + a.nbytes] = a.tobytes()
    with open(path, "wb") as f:
        f.write(out)


# ------------------------------------------------------------------- h5py-like facade
class _Dataset:
    def __init__(self, reader, addr):
        self._reader, self._addr = reader, addr
        self._value = None

    def _load(self):
        if self._value is None:
            self._value = self._reader.dataset(self._addr)
        return self._value

    def __getitem__(self, key):
        return self._load()[key]

    def __array__(self, dtype=None, copy=None):
        a = self._load()
        return a.astype(dtype) if dtype is not None else a

    @property
    def shape(self):
        return self._load().shape

    @property
    def dtype(self):
        return self._load().dtype


class File:
    """`File(path, "r")[name][()]` / `File(path, "w").create_dataset(name, data=a)`."""

    def __init__(self, path, mode="r"):
        if mode not in ("r", "w"):
            raise ValueError("mode must be 'r' or 'w'")
        self.path, self.mode = path, mode
        self._pending = {}
        self._reader = self._links = None
        if mode == "r":
            with open(path, "rb") as fh:
                self._reader = _Reader(mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ))
            self._links = self._reader.links(self._reader.root)

    def keys(self):
        return list(self._links) if self.mode == "r" else list(self._pending)

    def __contains__(self, name):
        return name in self.keys()

    def __getitem__(self, name):
        if self.mode != "r":
            raise H5Error("file is open for writing")
        if name not in self._links:
            raise KeyError("Unable to open object (object '%s' doesn't exist)" % name)
        return _Dataset(self._reader, self._links[name])

    def create_dataset(self, name, data):
        if self.mode != "w":
            raise H5Error("file is open read-only")
        if name in self._pending:
            raise ValueError("Unable to create dataset (name already exists)")
        self._pending[name] = np.asarray(data)
        return self._pending[name]

    def close(self):
        if self.mode == "w" and self._pending is not None:
            write_file(self.path, self._pending)
            self._pending = None
        self._reader = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
