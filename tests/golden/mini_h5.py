"""A minimal pure-Python HDF5 reader -- TEST INFRASTRUCTURE for tests/golden/gen_golden.py only.

h5py / anndata / scanpy are not installed in the build container, and the reference's own test
(/root/reference/test/test_pilot.py:6) starts with ``sc.read_h5ad('Tutorial/Datasets/Kidney_IgAN_G.h5ad')``.  This module reads
just enough of the HDF5 file format (version-0 superblock, version-1 object headers, symbol-table groups with their
version-1 B-trees and local heaps, contiguous / compact / chunked (+ gzip, shuffle) dataset layouts, fixed-point, floating-point,
fixed-length and variable-length string datatypes with global heap collections, simple attributes) to load an ``.h5ad`` written
by anndata into numpy arrays.  Written from the published HDF5 File Format Specification (version 1.10, sections II-IV); it is
not a general HDF5 library and raises NotImplementedError on anything outside that subset.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class H5File:
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.b = fh.read()
        if self.b[:8] != b"\x89HDF\r\n\x1a\n":
            raise ValueError("not an HDF5 file")
        ver = self.b[8]
        if ver not in (0, 1):
            raise NotImplementedError("superblock version %d" % ver)
        self.so, self.sl = self.b[13], self.b[14]           # size of offsets / lengths
        if (self.so, self.sl) != (8, 8):
            raise NotImplementedError("offset/length sizes %d/%d" % (self.so, self.sl))
        p = 24 if ver == 0 else 28
        base, _free, _eof, _drv = struct.unpack_from("<4Q", self.b, p)
        if base != 0:
            raise NotImplementedError("non-zero base address")
        p += 32
        # root group symbol table entry
        _name_off, ohdr, cache_type = struct.unpack_from("<QQI", self.b, p)
        self.root = Group(self, ohdr)
        self._gcol = {}

    # -- global heap (variable-length data) -----------------------------------------------------------------------------
    def gheap_object(self, addr, index):
        col = self._gcol.get(addr)
        if col is None:
            if self.b[addr:addr + 4] != b"GCOL":
                raise ValueError("bad global heap collection at %#x" % addr)
            size = struct.unpack_from("<Q", self.b, addr + 8)[0]
            col = {}
            p, end = addr + 16, addr + size
            while p + 16 <= end:
                idx, _ref, _res, osz = struct.unpack_from("<HHIQ", self.b, p)
                if idx == 0:
                    break
                col[idx] = (p + 16, osz)
                p += 16 + ((osz + 7) & ~7)
            self._gcol[addr] = col
        off, osz = col[index]
        return self.b[off:off + osz]

    def __getitem__(self, path):
        node = self.root
        for part in [q for q in path.split("/") if q]:
            node = node[part]
        return node


class Obj:
    """An object header (version 1): its messages."""

    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        b = f.b
        if b[addr] != 1:
            raise NotImplementedError("object header version %d at %#x" % (b[addr], addr))
        n_msgs, _refc, hsize = struct.unpack_from("<HII", b, addr + 2)
        self.msgs = []
        blocks = [(addr + 16, hsize)]
        while blocks and len(self.msgs) < n_msgs:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and len(self.msgs) < n_msgs:
                mtype, msize, mflags = struct.unpack_from("<HHB", b, p)
                body = p + 8
                if mtype == 0x10:                                   # continuation
                    blocks.append(struct.unpack_from("<QQ", b, body))
                self.msgs.append((mtype, body, msize, mflags))
                p = body + msize

    def find(self, mtype):
        return [(body, size) for t, body, size, _ in self.msgs if t == mtype]

    # -- attributes ---------------------------------------------------------------------------------------------------------
    @property
    def attrs(self):
        out = {}
        b = self.f.b
        for body, _size in self.find(0x0C):
            ver = b[body]
            if ver == 1:
                nsz, dtsz, dssz = struct.unpack_from("<HHH", b, body + 2)
                p = body + 8
                name = b[p:p + nsz].split(b"\0")[0].decode()
                p += (nsz + 7) & ~7
                dt = parse_datatype(b, p)
                p += (dtsz + 7) & ~7
                shape = parse_dataspace(b, p)
                p += (dssz + 7) & ~7
            elif ver in (2, 3):
                nsz, dtsz, dssz = struct.unpack_from("<HHH", b, body + 2)
                p = body + (9 if ver == 3 else 8)
                name = b[p:p + nsz].split(b"\0")[0].decode()
                p += nsz
                dt = parse_datatype(b, p)
                p += dtsz
                shape = parse_dataspace(b, p)
                p += dssz
            else:
                raise NotImplementedError("attribute message version %d" % ver)
            n = int(np.prod(shape)) if shape else 1
            val = decode(self.f, dt, b[p:p + n * dt["size"]], n)
            out[name] = val.reshape(shape) if shape else val[0]
        return out


class Group(Obj):
    def __init__(self, f, addr):
        super().__init__(f, addr)
        st = self.find(0x11)
        if not st:
            raise NotImplementedError("group without a symbol table message (new-style group) at %#x" % addr)
        self.btree, self.heap = struct.unpack_from("<QQ", f.b, st[0][0])
        self._entries = None

    def _load(self):
        if self._entries is not None:
            return
        b = self.f.b
        if b[self.heap:self.heap + 4] != b"HEAP":
            raise ValueError("bad local heap")
        heap_data = struct.unpack_from("<Q", b, self.heap + 24)[0]
        self._entries = {}

        def walk(node):
            if b[node:node + 4] == b"TREE":
                ntype, level, used = struct.unpack_from("<BBH", b, node + 4)
                if ntype != 0:
                    raise ValueError("group B-tree expected")
                p = node + 24
                for i in range(used):
                    child = struct.unpack_from("<Q", b, p + 8 + i * 16)[0]
                    walk(child)
            elif b[node:node + 4] == b"SNOD":
                n = struct.unpack_from("<H", b, node + 6)[0]
                for i in range(n):
                    e = node + 8 + i * 40
                    name_off, ohdr = struct.unpack_from("<QQ", b, e)
                    s = heap_data + name_off
                    name = b[s:b.index(b"\0", s)].decode()
                    self._entries[name] = ohdr
            else:
                raise ValueError("unexpected node signature %r" % b[node:node + 4])
        walk(self.btree)

    def keys(self):
        self._load()
        return list(self._entries)

    def __contains__(self, name):
        self._load()
        return name in self._entries

    def __getitem__(self, name):
        self._load()
        addr = self._entries[name]
        o = Obj(self.f, addr)
        if o.find(0x11):
            return Group(self.f, addr)
        return Dataset(self.f, addr)


def parse_dataspace(b, p):
    ver, rank, flags = b[p], b[p + 1], b[p + 2]
    if ver == 1:
        q = p + 8
    elif ver == 2:
        if b[p + 3] == 2:            # null dataspace
            return (0,)
        q = p + 4
    else:
        raise NotImplementedError("dataspace version %d" % ver)
    return tuple(struct.unpack_from("<%dQ" % rank, b, q)) if rank else ()


def parse_datatype(b, p):
    cv, f0, f1, f2, size = struct.unpack_from("<BBBBI", b, p)
    cls, ver = cv & 0x0F, cv >> 4
    dt = {"class": cls, "size": size, "version": ver}
    if cls == 0:       # fixed point
        dt["np"] = np.dtype("%s%s%d" % (">" if f0 & 1 else "<", "i" if f0 & 8 else "u", size))
    elif cls == 1:     # floating point
        dt["np"] = np.dtype("%sf%d" % (">" if f0 & 1 else "<", size))
    elif cls == 3:     # fixed-length string
        dt["np"] = np.dtype("S%d" % size)
        dt["pad"] = f0 & 0x0F
    elif cls == 9:     # variable length
        dt["vlen_string"] = (f0 & 0x0F) == 1
        dt["base"] = parse_datatype(b, p + 8)
    elif cls == 8:     # enum (anndata writes booleans as an enum over int8)
        dt["base"] = parse_datatype(b, p + 8)
        dt["np"] = dt["base"]["np"]
    else:
        raise NotImplementedError("datatype class %d" % cls)
    return dt


def decode(f, dt, raw, n):
    if dt["class"] in (0, 1, 8):
        return np.frombuffer(raw, dtype=dt["np"], count=n).copy()
    if dt["class"] == 3:
        a = np.frombuffer(raw, dtype=dt["np"], count=n)
        return np.array([x.split(b"\0")[0].decode() for x in a], dtype=object)
    if dt["class"] == 9:
        out = np.empty(n, dtype=object)
        for i in range(n):
            length, addr, idx = struct.unpack_from("<IQI", raw, i * 16)
            data = f.gheap_object(addr, idx)[:length * dt["base"]["size"]] if addr not in (0, UNDEF) and length else b""
            out[i] = data.decode() if dt["vlen_string"] else np.frombuffer(data, dtype=dt["base"]["np"]).copy()
        return out
    raise NotImplementedError


class Dataset(Obj):
    def __init__(self, f, addr):
        super().__init__(f, addr)
        b = f.b
        self.shape = parse_dataspace(b, self.find(0x01)[0][0])
        self.dt = parse_datatype(b, self.find(0x03)[0][0])
        self.filters = []
        for body, _ in self.find(0x0B):
            ver, nf = b[body], b[body + 1]
            p = body + (8 if ver == 1 else 2)
            for _ in range(nf):
                if ver == 1:
                    fid, nlen, _fl, ncv = struct.unpack_from("<HHHH", b, p)
                    p += 8 + ((nlen + 7) & ~7)
                else:
                    fid = struct.unpack_from("<H", b, p)[0]
                    if fid < 256:
                        nlen = 0
                        _fl, ncv = struct.unpack_from("<HH", b, p + 2)
                        p += 6
                    else:
                        nlen, _fl, ncv = struct.unpack_from("<HHH", b, p + 2)
                        p += 8 + nlen
                cvals = struct.unpack_from("<%dI" % ncv, b, p)
                p += 4 * ncv
                if ver == 1 and ncv % 2:
                    p += 4
                self.filters.append((fid, cvals))

    def _unfilter(self, raw, mask=0):
        for i, (fid, cvals) in reversed(list(enumerate(self.filters))):
            if mask & (1 << i):
                continue
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:                      # shuffle
                es = cvals[0]
                n = len(raw) // es
                raw = np.frombuffer(raw[:n * es], dtype=np.uint8).reshape(es, n).T.tobytes() + raw[n * es:]
            else:
                raise NotImplementedError("HDF5 filter id %d" % fid)
        return raw

    def read(self):
        b = self.f.b
        body = self.find(0x08)[0][0]
        ver = b[body]
        n = int(np.prod(self.shape)) if self.shape else 1
        es = self.dt["size"]
        if ver != 3:
            raise NotImplementedError("data layout message version %d" % ver)
        lclass = b[body + 1]
        if lclass == 0:                         # compact
            size = struct.unpack_from("<H", b, body + 2)[0]
            raw = b[body + 4:body + 4 + size]
        elif lclass == 1:                       # contiguous
            addr, size = struct.unpack_from("<QQ", b, body + 2)
            raw = b"" if addr == UNDEF else b[addr:addr + size]
            if addr == UNDEF:
                raw = bytes(n * es)
        elif lclass == 2:                       # chunked
            rank = b[body + 2]
            btree = struct.unpack_from("<Q", b, body + 3)[0]
            cdims = struct.unpack_from("<%dI" % rank, b, body + 11)
            cshape = cdims[:-1]
            if cdims[-1] != es:
                raise ValueError("chunk element size mismatch")
            out = np.zeros(self.shape, dtype=np.dtype("V%d" % es))
            if btree != UNDEF:
                self._read_chunks(btree, rank, cshape, out)
            raw = out.tobytes()
        else:
            raise NotImplementedError("layout class %d" % lclass)
        arr = decode(self.f, self.dt, raw, n)
        return arr.reshape(self.shape) if self.shape else arr[0]

    def _read_chunks(self, node, rank, cshape, out):
        b = self.f.b
        if b[node:node + 4] != b"TREE":
            raise ValueError("bad chunk B-tree node")
        ntype, level, used = struct.unpack_from("<BBH", b, node + 4)
        if ntype != 1:
            raise ValueError("chunk B-tree expected")
        ksz = 8 + 8 * rank
        p = node + 24
        es = out.dtype.itemsize
        for i in range(used):
            key = p + i * (ksz + 8)
            csize, mask = struct.unpack_from("<II", b, key)
            offs = struct.unpack_from("<%dQ" % rank, b, key + 8)[:-1]
            child = struct.unpack_from("<Q", b, key + ksz)[0]
            if level > 0:
                self._read_chunks(child, rank, cshape, out)
                continue
            raw = self._unfilter(b[child:child + csize], mask)
            chunk = np.frombuffer(raw, dtype=out.dtype, count=int(np.prod(cshape))).reshape(cshape)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cshape, out.shape))
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]


# ---- anndata layout ------------------------------------------------------------------------------------------------------
def _read_column(node):
    """One obs / var column as anndata (>= 0.7) writes it: a plain dataset, or a categorical group (codes + categories)."""
    if isinstance(node, Dataset):
        return node.read()
    if "codes" in node and "categories" in node:
        codes, cats = node["codes"].read(), node["categories"].read()
        out = np.empty(len(codes), dtype=object)
        for i, c in enumerate(codes):
            out[i] = cats[c] if c >= 0 else np.nan
        return out, cats
    raise NotImplementedError("obs column layout %s" % node.keys())


def read_h5ad(path):
    """-> dict(X=ndarray (cells x vars), obs=dict column -> array (categoricals as (values, categories)), obs_names, var_names).
    Only dense X (what Kidney_IgAN_G.h5ad holds); a sparse X group raises NotImplementedError."""
    f = H5File(path)
    root = f.root
    X = root["X"]
    if not isinstance(X, Dataset):
        raise NotImplementedError("sparse X (group with %s)" % X.keys())
    out = {"X": X.read()}
    for frame in ("obs", "var"):
        g = root[frame]
        attrs = g.attrs
        index_key = attrs.get("_index", "_index")
        order = [str(c) for c in attrs["column-order"]] if "column-order" in attrs else [k for k in g.keys() if k != index_key and not k.startswith("__")]
        cols = {}
        for name in order:
            node = g[name]
            if isinstance(node, Dataset) and "__categories" in g and name in g["__categories"]:      # anndata 0.7 categoricals
                cats = g["__categories"][name].read()
                codes = node.read()
                cols[name] = (np.array([cats[c] if c >= 0 else np.nan for c in codes], dtype=object), cats)
            else:
                cols[name] = _read_column(node)
        out[frame] = cols
        out[frame + "_names"] = g[index_key].read()
    return out
