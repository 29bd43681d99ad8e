"""Keras weight files (`model_*_weight.h5`) without h5py: a reader and a writer for the part of HDF5 those files use.

The reference stores weights with Keras 2.1 `Model.save_weights` / reads them with `load_weights` (agent/model.py:82-101;
Keras==2.1.2, h5py and libhdf5 are third-party dependencies that are not part of the reference tree).  What that call
puts into the file (keras/engine/topology.py `save_weights_to_hdf5_group`, restated):

    /                       attrs  layer_names   [bytes]  every layer of the model, in model.layers order
                                   backend, keras_version
    /<layer>                attrs  weight_names  [bytes]  e.g. conv2d_3/kernel:0, conv2d_3/bias:0   (empty for Activation..)
    /<layer>/<weight name>  dataset float32 - the weight name contains a '/', so h5py makes /<layer>/<layer>/kernel:0

and what h5py + libhdf5 make of it on disk with their default settings ("earliest" format; HDF5 File Format
Specification version 1.1/2.0): a version-0 superblock, version-1 object headers, "old style" groups (symbol table =
v1 B-tree + local heap + symbol table nodes), contiguous little-endian datasets, attributes as header messages with
fixed-length strings (h5py 2.x) or variable-length strings kept in global heap collections (h5py 3.x).

`H5File` reads that subset - plus chunked datasets with the deflate / shuffle / fletcher32 filters, in case a file went
through h5repack - and raises `H5FormatError` naming the feature for anything else (e.g. files written with
libver="latest": version-2 object headers, fractal-heap groups).  `write_h5` writes the same subset, byte layouts chosen
like libhdf5's own so that h5py / Keras read the result.

Pinned by tests/test_keras_h5.py: fixtures under tests/golden/keras_h5/ were written by real h5py 3.3.0 / libhdf5
1.10.6 (tests/golden/make_golden_keras_h5.py) in both string flavours; where an h5py interpreter is present the writer's
files are also read back through h5py.
"""
import struct
import zlib

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5FormatError(ValueError):
    """The file is not HDF5, is damaged, or uses a feature outside the subset Keras weight files need."""


# ---------------------------------------------------------------------------------------------------------------------
# reading
# ---------------------------------------------------------------------------------------------------------------------
class _Type:
    """A decoded datatype message: numpy dtype for fixed-size elements, or a variable-length string marker."""

    def __init__(self, size, dtype=None, vlen_str=False, strpad=None):
        self.size, self.dtype, self.vlen_str, self.strpad = size, dtype, vlen_str, strpad


class H5File:
    """`H5File(path)`; `f.attrs`, `f.keys()`, `f["conv2d_1/conv2d_1/kernel:0"]` -> Group / Dataset, `f.visit_datasets()`."""

    def __init__(self, path_or_bytes):
        if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
            self.buf = bytes(path_or_bytes)
        else:
            with open(path_or_bytes, "rb") as f:
                self.buf = f.read()
        self._superblock()
        self.root = Group(self, self._root_header, "/")

    # -- primitives -----------------------------------------------------------------------------------------------------
    def _u(self, off, n):
        if off < 0 or off + n > len(self.buf):
            raise H5FormatError(f"truncated file: {n} bytes wanted at offset {off}, file has {len(self.buf)}")
        return int.from_bytes(self.buf[off:off + n], "little")

    def _addr(self, off):
        return self._u(off, self.O)

    def _len(self, off):
        return self._u(off, self.L)

    def _bytes(self, off, n):
        if off < 0 or off + n > len(self.buf):
            raise H5FormatError(f"truncated file: {n} bytes wanted at offset {off}, file has {len(self.buf)}")
        return self.buf[off:off + n]

    def _superblock(self):
        b = self.buf
        base = 0
        while base + 8 <= len(b) and b[base:base + 8] != SIGNATURE:   # a user block moves it to 512, 1024, 2048, ...
            base = 512 if base == 0 else base * 2
        if base + 8 > len(b):
            raise H5FormatError("not an HDF5 file (signature not found)")
        ver = b[base + 8]
        if ver not in (0, 1):
            raise H5FormatError(f"HDF5 superblock version {ver} (file written with libver='latest'?): only the default "
                                f"'earliest' on-disk format (versions 0/1) is supported")
        self.O, self.L = b[base + 13], b[base + 14]
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise H5FormatError(f"unsupported offset/length sizes {self.O}/{self.L}")
        self.undef = (1 << 8 * self.O) - 1   # the "undefined address"
        self.leaf_k, self.internal_k = self._u(base + 16, 2), self._u(base + 18, 2)
        p = base + 24 + (4 if ver == 1 else 0)
        self.base = self._addr(p)
        if self.base == 0 and base:
            self.base = base   # h5jam'ed files keep base address 0 relative to the superblock in old versions
        p += 4 * self.O   # base, free-space info, end of file, driver info
        # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
        self._root_header = self._addr(p + self.O)

    # -- object headers (version 1) -------------------------------------------------------------------------------------
    def _messages(self, addr):
        """[(type, flags, offset of the data, size)] of the version-1 object header at `addr`, continuation blocks followed."""
        a = self.base + addr
        if self._bytes(a, 4) == b"OHDR":
            raise H5FormatError("version-2 object header (file written with libver='latest'): not supported")
        if self.buf[a] != 1:
            raise H5FormatError(f"object header version {self.buf[a]} at {addr}: not supported")
        n_msgs, hdr_size = self._u(a + 2, 2), self._u(a + 8, 4)
        blocks = [(a + 16, hdr_size)]   # 12 bytes of prefix + 4 of padding: messages are 8-byte aligned
        out = []
        while blocks and len(out) < n_msgs:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and len(out) < n_msgs:
                mtype, msize, mflags = self._u(p, 2), self._u(p + 2, 2), self.buf[p + 4]
                data = p + 8
                if mtype == 0x0010:   # continuation: offset, length
                    blocks.append((self.base + self._addr(data), self._len(data + self.O)))
                out.append((mtype, mflags, data, msize))
                p = data + msize
        return out

    # -- datatype / dataspace ---------------------------------------------------------------------------------------------
    def _datatype(self, p):
        cls, ver = self.buf[p] & 0x0F, self.buf[p] >> 4
        bits = self._u(p + 1, 3)
        size = self._u(p + 4, 4)
        if ver not in (1, 2, 3):
            raise H5FormatError(f"datatype message version {ver}")
        if cls == 0:   # fixed point
            dt = np.dtype(("i" if bits & 8 else "u") + str(size)).newbyteorder(">" if bits & 1 else "<")
            return _Type(size, dt)
        if cls == 1:   # floating point
            if size not in (2, 4, 8):
                raise H5FormatError(f"{size}-byte floating point type")
            return _Type(size, np.dtype("f" + str(size)).newbyteorder(">" if bits & 1 else "<"))
        if cls == 3:   # fixed-length string
            return _Type(size, np.dtype("S" + str(size)), strpad=bits & 0x0F)
        if cls == 9:   # variable length
            if bits & 0x0F != 1:
                raise H5FormatError("variable-length sequence type (only variable-length strings are supported)")
            return _Type(size, vlen_str=True)
        names = {2: "time", 4: "bitfield", 5: "opaque", 6: "compound", 7: "reference", 8: "enum", 10: "array"}
        raise H5FormatError(f"HDF5 datatype class {names.get(cls, cls)}: not supported")

    def _dataspace(self, p):
        ver, rank, flags = self.buf[p], self.buf[p + 1], self.buf[p + 2]
        if ver == 1:
            q = p + 8
        elif ver == 2:
            if self.buf[p + 3] == 2:   # null dataspace
                return None
            q = p + 4
        else:
            raise H5FormatError(f"dataspace message version {ver}")
        return tuple(self._len(q + i * self.L) for i in range(rank))

    def _global_heap_object(self, addr, index):
        a = self.base + addr
        if self._bytes(a, 4) != b"GCOL":
            raise H5FormatError(f"global heap collection expected at {addr}")
        end = a + self._len(a + 8)
        p = a + 8 + self.L
        while p + 8 + self.L <= end:
            idx, size = self._u(p, 2), self._len(p + 8)
            if idx == 0:
                break
            if idx == index:
                return self._bytes(p + 8 + self.L, size)
            p += 8 + self.L + (size + 7) // 8 * 8
        raise H5FormatError(f"global heap object {index} not found in the collection at {addr}")

    def _decode(self, typ, shape, raw):
        """raw element bytes -> numpy array (fixed-size types) / list of bytes (variable-length strings)"""
        n = 1
        for d in (shape or ()):
            n *= d
        if shape is None:
            n = 0
        if typ.vlen_str:
            out = []
            step = 4 + self.O + 4
            for i in range(n):
                e = raw[i * step:(i + 1) * step]
                length, addr, index = int.from_bytes(e[:4], "little"), int.from_bytes(e[4:4 + self.O], "little"), \
                    int.from_bytes(e[4 + self.O:], "little")
                out.append(b"" if addr in (0, self.undef) else self._global_heap_object(addr, index)[:length])
            return out[0] if shape == () else out
        if len(raw) < n * typ.size:
            raise H5FormatError(f"{n} elements of {typ.size} bytes expected, {len(raw)} bytes present")
        a = np.frombuffer(raw, dtype=typ.dtype, count=n)
        if typ.dtype.kind == "S" and typ.strpad == 2:   # space padded
            a = np.char.rstrip(a, b" ")
        a = a.reshape(shape or ())
        return a[()] if shape == () else a

    def _attribute(self, p, size):
        ver = self.buf[p]
        if ver not in (1, 2, 3):
            raise H5FormatError(f"attribute message version {ver}")
        if ver >= 2 and self.buf[p + 1] & 3:
            raise H5FormatError("attribute with a shared (committed) datatype or dataspace: not supported")
        name_size, dt_size, ds_size = self._u(p + 2, 2), self._u(p + 4, 2), self._u(p + 6, 2)
        q = p + 8 + (1 if ver == 3 else 0)
        pad = (lambda n: (n + 7) // 8 * 8) if ver == 1 else (lambda n: n)
        name = self._bytes(q, name_size).split(b"\0")[0].decode("utf8")
        q += pad(name_size)
        typ = self._datatype(q)
        q += pad(dt_size)
        shape = self._dataspace(q)
        q += pad(ds_size)
        return name, self._decode(typ, shape, self._bytes(q, p + size - q))

    def _attrs(self, messages):
        out = {}
        for mtype, mflags, p, size in messages:
            if mtype == 0x000C:
                if mflags & 2:
                    raise H5FormatError("shared attribute message: not supported")
                name, value = self._attribute(p, size)
                out[name] = value
            elif mtype == 0x0015:   # attribute info: dense storage only exists behind version-2 headers
                if self._addr(p + 2 + (2 if self.buf[p + 1] & 1 else 0)) != self.undef:
                    raise H5FormatError("attributes in dense (fractal heap) storage: not supported")
        return out

    # -- old-style groups -----------------------------------------------------------------------------------------------
    def _heap_string(self, heap_addr, off):
        a = self.base + heap_addr
        if self._bytes(a, 4) != b"HEAP":
            raise H5FormatError(f"local heap expected at {heap_addr}")
        data = self.base + self._addr(a + 8 + 2 * self.L)
        end = self.buf.index(b"\0", data + off)
        return self.buf[data + off:end].decode("utf8")

    def _group_entries(self, btree_addr, heap_addr):
        """{name: object header address} of the group whose symbol table is the v1 B-tree at `btree_addr`."""
        out = {}
        stack = [btree_addr]
        while stack:
            a = self.base + stack.pop()
            sig = self._bytes(a, 4)
            if sig == b"TREE":
                if self.buf[a + 4] != 0:
                    raise H5FormatError("group B-tree node of the wrong type")
                n = self._u(a + 6, 2)
                p = a + 8 + 2 * self.O
                for i in range(n):   # key, child, key, child, ..., key
                    stack.append(self._addr(p + self.L + i * (self.L + self.O)))
            elif sig == b"SNOD":
                n = self._u(a + 6, 2)
                p = a + 8
                for i in range(n):
                    e = p + i * (2 * self.O + 24)
                    if self._u(e + 2 * self.O, 4) == 2:
                        raise H5FormatError("symbolic link in a group: not supported")
                    out[self._heap_string(heap_addr, self._addr(e))] = self._addr(e + self.O)
            else:
                raise H5FormatError(f"group B-tree: unknown node signature {sig!r}")
        return out

    # -- dataset storage ------------------------------------------------------------------------------------------------
    def _filters(self, p):
        ver, n = self.buf[p], self.buf[p + 1]
        q = p + (8 if ver == 1 else 2)
        out = []
        for _ in range(n):
            fid = self._u(q, 2)
            if ver == 1 or fid >= 256:
                name_len = self._u(q + 2, 2)
                q += 2
            else:
                name_len = 0
            ncd = self._u(q + 4, 2)
            q += 6
            q += (name_len + 7) // 8 * 8 if ver == 1 else name_len
            cd = [self._u(q + 4 * i, 4) for i in range(ncd)]
            q += 4 * ncd + (4 if ver == 1 and ncd % 2 else 0)
            out.append((fid, cd))
        return out

    def _unfilter(self, raw, filters, mask, elem_size):
        for k in range(len(filters) - 1, -1, -1):   # the pipeline is undone back to front
            if mask >> k & 1:
                continue
            fid, cd = filters[k]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = cd[0] if cd else elem_size
                n = len(raw) // es
                a = np.frombuffer(raw, np.uint8, n * es).reshape(es, n).T
                raw = a.tobytes() + raw[n * es:]
            elif fid == 3:
                raw = raw[:-4]
            else:
                raise H5FormatError(f"HDF5 filter {fid} (only deflate, shuffle, fletcher32 are supported)")
        return raw

    def _chunks(self, btree_addr, rank):
        """[(offsets, size, filter mask, address)] of the raw-data chunk B-tree (v1, node type 1)."""
        out = []
        stack = [btree_addr]
        key = 8 + 8 * (rank + 1)
        while stack:
            a = self.base + stack.pop()
            if self._bytes(a, 4) != b"TREE" or self.buf[a + 4] != 1:
                raise H5FormatError("chunk B-tree node expected")
            level, n = self.buf[a + 5], self._u(a + 6, 2)
            p = a + 8 + 2 * self.O
            for i in range(n):
                k = p + i * (key + self.O)
                child = self._addr(k + key)
                if level:
                    stack.append(child)
                else:
                    out.append((tuple(self._u(k + 8 + 8 * d, 8) for d in range(rank)), self._u(k, 4), self._u(k + 4, 4), child))
        return out


class _Object:
    def __init__(self, f, addr, name):
        self.file, self.addr, self.name = f, addr, name
        self._msgs = f._messages(addr)
        for mtype, mflags, p, size in self._msgs:
            if mflags & 2 and mtype in (0x0001, 0x0003, 0x0008, 0x000B):
                raise H5FormatError(f"{name}: shared header message (committed datatype?): not supported")
        self.attrs = f._attrs(self._msgs)

    def _find(self, mtype):
        return next(((p, size) for t, _, p, size in self._msgs if t == mtype), None)


class Group(_Object):
    def __init__(self, f, addr, name):
        super().__init__(f, addr, name)
        st = self._find(0x0011)
        if st is None:
            if self._find(0x0002) is not None or self._find(0x0006) is not None:
                raise H5FormatError(f"{name}: new-style group (link messages): not supported")
            raise H5FormatError(f"{name}: not a group")
        self._entries = f._group_entries(f._addr(st[0]), f._addr(st[0] + f.O))

    def keys(self):
        return sorted(self._entries)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._entries:
                raise KeyError(f"{path!r} not found in {self.name}")
            addr = node._entries[part]
            child_name = node.name.rstrip("/") + "/" + part
            msgs = {t for t, _, _, _ in self.file._messages(addr)}
            node = Dataset(self.file, addr, child_name) if 0x0008 in msgs else Group(self.file, addr, child_name)
        return node

    def visit_datasets(self):
        """{path relative to this group: Dataset} of everything below it."""
        out = {}
        for k in self.keys():
            child = self[k]
            if isinstance(child, Dataset):
                out[k] = child
            else:
                out.update({k + "/" + n: d for n, d in child.visit_datasets().items()})
        return out


class Dataset(_Object):
    def __init__(self, f, addr, name):
        super().__init__(f, addr, name)
        dt, ds, lay = self._find(0x0003), self._find(0x0001), self._find(0x0008)
        if dt is None or ds is None or lay is None:
            raise H5FormatError(f"{name}: dataset without datatype / dataspace / layout message")
        self._type = f._datatype(dt[0])
        self.shape = f._dataspace(ds[0])
        self.dtype = self._type.dtype
        self._layout = lay[0]

    def read(self):
        f, p = self.file, self._layout
        shape = self.shape
        n = 1
        for d in (shape or ()):
            n *= d
        nbytes = n * self._type.size
        ver = f.buf[p]
        if ver != 3:
            raise H5FormatError(f"{self.name}: data layout message version {ver}: not supported")
        cls = f.buf[p + 1]
        if cls == 0:   # compact: the data sits in the header
            raw = f._bytes(p + 4, f._u(p + 2, 2))
        elif cls == 1:   # contiguous
            addr, size = f._addr(p + 2), f._len(p + 2 + f.O)
            if addr == f.undef:
                raw = bytes(nbytes)   # never written: the fill value (zeros)
            else:
                raw = f._bytes(f.base + addr, min(size, nbytes) if size else nbytes)
        elif cls == 2:   # chunked
            rank = f.buf[p + 2] - 1
            btree = f._addr(p + 3)
            cdims = tuple(f._u(p + 3 + f.O + 4 * i, 4) for i in range(rank))
            if rank != len(shape):
                raise H5FormatError(f"{self.name}: chunk rank {rank} != dataset rank {len(shape)}")
            if self._type.vlen_str:
                raise H5FormatError(f"{self.name}: chunked variable-length strings: not supported")
            flt = self._find(0x000B)
            filters = f._filters(flt[0]) if flt else []
            out = np.zeros(shape, self._type.dtype)
            if btree != f.undef:
                for offs, size, mask, addr in f._chunks(btree, rank):
                    raw = f._unfilter(f._bytes(f.base + addr, size), filters, mask, self._type.size)
                    c = np.frombuffer(raw, self._type.dtype, int(np.prod(cdims))).reshape(cdims)
                    sel = tuple(slice(o, min(o + cd, s)) for o, cd, s in zip(offs, cdims, shape))
                    out[sel] = c[tuple(slice(0, s.stop - s.start) for s in sel)]
            return out
        else:
            raise H5FormatError(f"{self.name}: data layout class {cls}")
        return f._decode(self._type, shape, raw)


# Keep h5py's spelling available on the file object
H5File.attrs = property(lambda self: self.root.attrs)
H5File.keys = lambda self: self.root.keys()
H5File.__getitem__ = lambda self, path: self.root[path]
H5File.__contains__ = lambda self, path: path in self.root
H5File.visit_datasets = lambda self: self.root.visit_datasets()


def _names(value):
    """An attribute holding names (fixed-length 'S' array, list of bytes, or an empty placeholder) -> [str]."""
    if value is None:
        return []
    if isinstance(value, (bytes, np.bytes_)):
        return [bytes(value).decode("utf8")]
    return [bytes(v).decode("utf8") for v in (value.tolist() if isinstance(value, np.ndarray) and value.dtype.kind == "S" else
                                               value if isinstance(value, list) else [])]


def read_keras_weights(path_or_bytes):
    """A Keras `save_weights` file (or the `model_weights` group of a `model.save` file) ->
    ({'<layer>/<weight>:0': array} in the file's own order, {'layer_names': [...], 'backend': .., 'keras_version': ..})."""
    f = H5File(path_or_bytes)
    g = f.root
    if "layer_names" not in g.attrs and "model_weights" in g:
        g = g["model_weights"]
    if "layer_names" not in g.attrs:
        raise H5FormatError("no 'layer_names' attribute: not a Keras weight file")
    layer_names = _names(g.attrs["layer_names"])
    arrays = {}
    for lname in layer_names:
        lg = g[lname]
        for wname in _names(lg.attrs.get("weight_names")):
            d = lg[wname]
            if not isinstance(d, Dataset):
                raise H5FormatError(f"{lname}/{wname} is not a dataset")
            arrays[wname if wname.startswith(lname + "/") else f"{lname}/{wname}"] = np.array(d.read())
    info = {"layer_names": layer_names}
    for k in ("backend", "keras_version"):
        if k in g.attrs:
            info[k] = _names(g.attrs[k])[0] if _names(g.attrs[k]) else None
    return arrays, info


# ---------------------------------------------------------------------------------------------------------------------
# writing
# ---------------------------------------------------------------------------------------------------------------------
_LEAF_K, _INTERNAL_K = 4, 16   # libhdf5's defaults: a symbol table node holds 2 * 4 entries, a B-tree node 2 * 16 children
_O = _L = 8


def _pad8(b):
    return b + bytes(-len(b) % 8)


def _u16(x):
    return struct.pack("<H", x)


def _u32(x):
    return struct.pack("<I", x)


def _u64(x):
    return struct.pack("<Q", x)


def _msg(mtype, data, flags=0):
    data = _pad8(data)
    if len(data) > 0xFFF8:   # the size field of a version-1 header message is 16 bits (HDF5's 64 KiB attribute limit)
        raise H5FormatError(f"header message of {len(data)} bytes: attributes are limited to 64 KiB in this file format")
    return _u16(mtype) + _u16(len(data)) + bytes([flags, 0, 0, 0]) + data


def _dataspace_msg(shape):
    """version 1, with maximum dimensions (what h5py writes)"""
    if shape == ():
        return bytes([1, 0, 0, 0, 0, 0, 0, 0])
    return bytes([1, len(shape), 1, 0, 0, 0, 0, 0]) + b"".join(_u64(d) for d in shape) * 2


def _datatype_msg(dtype):
    dtype = np.dtype(dtype)
    if dtype.kind == "S":   # fixed-length, null-padded ASCII string
        return bytes([0x13, 0x01, 0, 0]) + _u32(max(dtype.itemsize, 1))
    if dtype.kind == "f" and dtype.itemsize in (4, 8):
        exp_bits, mant_bits = (8, 23) if dtype.itemsize == 4 else (11, 52)
        return (bytes([0x11, 0x20, dtype.itemsize * 8 - 1, 0]) + _u32(dtype.itemsize) + _u16(0) + _u16(dtype.itemsize * 8)
                + bytes([mant_bits, exp_bits, 0, mant_bits]) + _u32((1 << (exp_bits - 1)) - 1))
    if dtype.kind in "iu":
        return (bytes([0x10, 0x08 if dtype.kind == "i" else 0, 0, 0]) + _u32(dtype.itemsize) + _u16(0) + _u16(dtype.itemsize * 8))
    raise H5FormatError(f"cannot write dtype {dtype}")


def _attribute_msg(name, value):
    """version-1 attribute message: everything padded to 8 bytes"""
    if isinstance(value, (bytes, str)):
        value = np.bytes_(value.encode("utf8") if isinstance(value, str) else value)
    a = np.asarray(value)
    if a.dtype.kind == "U":
        a = np.char.encode(a, "utf8")
    if a.dtype.kind == "S" and a.dtype.itemsize == 0:
        a = a.astype("S1")
    a = np.asarray(a, a.dtype.newbyteorder("<") if a.dtype.kind in "fiu" else a.dtype, order="C")   # keeps 0-d scalars 0-d
    nm = name.encode("utf8") + b"\0"
    dt, ds = _datatype_msg(a.dtype), _dataspace_msg(a.shape)
    return _msg(0x000C, bytes([1, 0]) + _u16(len(nm)) + _u16(len(dt)) + _u16(len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + a.tobytes(),
                flags=4)


class _Writer:
    def __init__(self):
        self.parts = []   # (address, bytes)
        self.end = 0

    def alloc(self, n):
        a = (self.end + 7) // 8 * 8
        self.end = a + n
        return a

    def put(self, addr, data):
        self.parts.append((addr, data))

    def header(self, messages):
        body = b"".join(messages)
        addr = self.alloc(16 + len(body))
        self.put(addr, bytes([1, 0]) + _u16(len(messages)) + _u32(1) + _u32(len(body)) + bytes(4) + body)
        return addr

    def dataset(self, array):
        a = np.asarray(array)
        a = np.asarray(a, a.dtype.newbyteorder("<") if a.dtype.kind in "fiu" else a.dtype, order="C")
        raw = a.tobytes()
        data_addr = self.alloc(len(raw)) if raw else UNDEF
        if raw:
            self.put(data_addr, raw)
        return self.header([
            _msg(0x0001, _dataspace_msg(a.shape)),
            _msg(0x0003, _datatype_msg(a.dtype), flags=1),
            _msg(0x0005, bytes([2, 2, 2, 1]) + _u32(0), flags=1),   # fill value: allocate late, write if set, default value
            _msg(0x0008, bytes([3, 1]) + _u64(data_addr) + _u64(len(raw)))])

    def group(self, attrs, children):
        """children: {name: object header address}.  Returns (header address, B-tree address, heap address)."""
        names = sorted(children, key=lambda s: s.encode("utf8"))
        # local heap: the empty string at offset 0, then the names, each padded to 8 bytes
        heap_data, offs = bytearray(8), {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data += _pad8(n.encode("utf8") + b"\0")
        heap_data_addr_slot = self.alloc(32 + len(heap_data))
        heap_addr = heap_data_addr_slot
        self.put(heap_addr, b"HEAP" + bytes(4) + _u64(len(heap_data)) + _u64(1) + _u64(heap_addr + 32) + bytes(heap_data))
        # symbol table nodes: up to 2 * leaf K entries each, always allocated at full size
        leaves = []   # (address, heap offset of the largest name)
        per = 2 * _LEAF_K
        chunks = [names[i:i + per] for i in range(0, len(names), per)] or [[]]
        for chunk in chunks:
            a = self.alloc(8 + per * 40)
            body = b"SNOD" + bytes([1, 0]) + _u16(len(chunk))
            for n in chunk:
                body += _u64(offs[n]) + _u64(children[n]) + _u32(0) + _u32(0) + bytes(16)
            self.put(a, body + bytes(8 + per * 40 - len(body)))
            leaves.append((a, offs[chunk[-1]] if chunk else 0))
        # B-tree: level 0 nodes point at the symbol table nodes; more levels only for very large groups
        fan = 2 * _INTERNAL_K
        node_size = 8 + 2 * _O + (fan + 1) * _L + fan * _O
        level, kids = 0, leaves
        while True:
            groups = [kids[i:i + fan] for i in range(0, len(kids), fan)]
            addrs = [self.alloc(node_size) for _ in groups]
            nodes = []
            first_key = 0
            for gi, g in enumerate(groups):
                body = b"TREE" + bytes([0, level]) + _u16(len(g)) + _u64(addrs[gi - 1] if gi else UNDEF) + \
                    _u64(addrs[gi + 1] if gi + 1 < len(groups) else UNDEF) + _u64(first_key)
                for child_addr, max_key in g:
                    body += _u64(child_addr) + _u64(max_key)
                self.put(addrs[gi], body + bytes(node_size - len(body)))
                first_key = g[-1][1]
                nodes.append((addrs[gi], g[-1][1]))
            if len(nodes) == 1:
                btree_addr = nodes[0][0]
                break
            level, kids = level + 1, nodes
        msgs = [_msg(0x0011, _u64(btree_addr) + _u64(heap_addr))] + [_attribute_msg(k, v) for k, v in attrs.items()]
        return self.header(msgs), btree_addr, heap_addr

    def finish(self, root):
        root_addr, btree, heap = root
        sb = (SIGNATURE + bytes([0, 0, 0, 0, 0, _O, _L, 0]) + _u16(_LEAF_K) + _u16(_INTERNAL_K) + _u32(0)
              + _u64(0) + _u64(UNDEF) + _u64(self.end) + _u64(UNDEF)
              + _u64(0) + _u64(root_addr) + _u32(1) + _u32(0) + _u64(btree) + _u64(heap))
        out = bytearray(self.end)
        out[:len(sb)] = sb
        for addr, data in self.parts:
            out[addr:addr + len(data)] = data
        return bytes(out)


def _write_node(w, node):
    """node: ndarray (dataset) or (attrs dict, {name: node}) (group) -> object header address"""
    if not isinstance(node, tuple):
        return w.dataset(node)
    attrs, children = node
    return w.group(attrs, {name: _write_node(w, child) for name, child in children.items()})[0]


def write_h5(path, attrs, children):
    """Writes the file `path` with root attributes `attrs` and `children` = {name: ndarray | (attrs, children)}.
    The bytes depend on the content only (no time stamps): equal weights give equal files, hence equal digests
    (agent/model.py:74-80 identifies a model by the sha256 of its weight file)."""
    w = _Writer()
    w.alloc(96)   # the superblock
    kids = {name: _write_node(w, child) for name, child in children.items()}
    data = w.finish(w.group(attrs, kids))
    if path is None:
        return data
    with open(path, "wb") as f:
        f.write(data)
    return data


def write_keras_weights(path, layers, backend="tensorflow", keras_version="2.1.2"):
    """`layers`: [(layer name, [(weight name as Keras gives it, e.g. 'conv2d_1/kernel:0', array), ...]), ...] in
    model.layers order, weightless layers included with an empty list - the layout of Keras' save_weights (module
    docstring).  Strings are written as fixed-length byte strings, as h5py 2.x (the reference's era) did."""
    def names_attr(names):
        return np.array([n.encode("utf8") for n in names], dtype="S") if names else np.zeros((0,), "S1")
    children = {}
    for lname, weights in layers:
        sub = {}
        for wname, arr in weights:   # a '/' in the weight name nests groups, exactly as h5py's create_dataset does
            parts = wname.split("/")
            d = sub
            for p in parts[:-1]:
                d = d.setdefault(p, ({}, {}))[1]
            d[parts[-1]] = np.asarray(arr, dtype=np.float32)
        children[lname] = ({"weight_names": names_attr([w for w, _ in weights])}, sub)
    attrs = {"layer_names": names_attr([l for l, _ in layers]), "backend": backend, "keras_version": keras_version}
    return write_h5(path, attrs, children)
