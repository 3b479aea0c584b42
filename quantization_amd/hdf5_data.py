"""`read_hdf5_data` of the reference (/root/reference/quantization/quantization.py:746-820): every dataset of an HDF5
archive -> one (tot_frames, dim) float16 matrix, rows shuffled, split into (train, valid).

The reference reads the archive with h5py.  h5py is used here too when it is importable; when it is not (the MI355X image
ships without it) the archive is read by the small pure-Python reader below, which understands what
`h5py.File(name, 'w').create_dataset(key, data=x)` writes (the layout of the reference's test_write_hdf5.py:25-31):
superblock version 0 or 1, old-style groups (symbol-table B-tree + local heap), version-1 object headers, fixed- and
floating-point little/big-endian element types, contiguous / compact / chunked-without-filters storage.  Anything else
(compression filters, new-style groups, superblock >= 2) raises with a message that says so.

Deviation, documented: the reference slices with `valid_frames = 0.05 * tot_frames`, a float, and therefore raises a
TypeError for archives of <= 200,000 frames (:812-820; only above that does the cap of 10,000 make it an int).  Here the
count is int(...) of the same expression, so small archives work; for archives the reference can read the split is
identical (same order of datasets, same np.random.shuffle call on the same matrix).
"""
import logging
import mmap
import struct
from typing import Dict, List, Tuple

import numpy as np
import torch

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5FormatError(RuntimeError):
    pass


class MiniHdf5File:
    """Read-only view of the datasets of the ROOT group of an HDF5 file (see the module docstring for the subset)."""

    def __init__(self, filename: str):
        with open(filename, "rb") as f:      # mapped, not read: a dataset is only paged in when it is copied out
            try:
                self.buf = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            except ValueError as e:          # (an empty file cannot be mapped)
                raise Hdf5FormatError(f"{filename}: not an HDF5 file ({e})") from None
        try:
            self._parse_superblock(filename)
        except BaseException:                  # (a junk file: the mapping must not outlive the failed constructor)
            self.close()
            raise

    def _parse_superblock(self, filename: str):
        b = self.buf
        base = -1
        off = 0
        while off + 8 <= len(b):                       # the superblock may sit at 0, 512, 1024, ...
            if b[off:off + 8] == _SIG:
                base = off
                break
            off = 512 if off == 0 else off * 2
        if base < 0:
            raise Hdf5FormatError(f"{filename}: not an HDF5 file (no superblock signature)")
        ver = b[base + 8]
        if ver > 1:
            raise Hdf5FormatError(f"{filename}: superblock version {ver} (written with libver='latest'?) is not supported "
                                  "by the built-in reader; install h5py")
        self.O, self.L = b[base + 13], b[base + 14]
        if self.O not in (4, 8) or self.L not in (4, 8):
            raise Hdf5FormatError("unsupported offset / length size")
        p = base + 24 + (4 if ver == 1 else 0)
        self.base_addr = self._off(p)
        p += 4 * self.O                                 # base, free-space, end-of-file, driver-info addresses
        # root group symbol table entry
        self.root_header = self._off(p + self.O)
        self._datasets = None

    # ---- primitive reads
    def _off(self, p: int) -> int:
        return int.from_bytes(self.buf[p:p + self.O], "little")

    def _len(self, p: int) -> int:
        return int.from_bytes(self.buf[p:p + self.L], "little")

    def _addr(self, a: int) -> int:
        return a + self.base_addr

    # ---- object headers (version 1)
    def _messages(self, addr: int) -> List[Tuple[int, int, int]]:
        """[(type, data offset, size)] of the object header at file address `addr`, continuation blocks included."""
        b, p = self.buf, self._addr(addr)
        if b[p:p + 4] == b"OHDR":
            raise Hdf5FormatError("version-2 object headers (libver='latest') are not supported by the built-in reader")
        if b[p] != 1:
            raise Hdf5FormatError(f"object header version {b[p]} not supported")
        nmsg = struct.unpack_from("<H", b, p + 2)[0]
        size = struct.unpack_from("<I", b, p + 8)[0]
        blocks = [(p + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            q, n = blocks.pop(0)
            end = q + n
            while q + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", b, q)
                data = q + 8
                if mtype == 0x10:                       # continuation
                    blocks.append((self._addr(self._off(data)), self._len(data + self.O)))
                out.append((mtype, data, msize))
                q = data + msize
        return out

    # ---- groups (symbol table: B-tree version 1 of type 0 + local heap)
    def _heap_data(self, addr: int) -> int:
        p = self._addr(addr)
        if self.buf[p:p + 4] != b"HEAP":
            raise Hdf5FormatError("bad local heap")
        return self._addr(self._off(p + 8 + 2 * self.L))

    def _name(self, heap_data: int, off: int) -> str:
        e = self.buf.find(b"\0", heap_data + off)
        if e < 0:
            raise Hdf5FormatError("unterminated name in a local heap")
        return self.buf[heap_data + off:e].decode("utf-8")

    def _group_entries(self, btree: int, heap_data: int, out: List[Tuple[str, int]]):
        b, p = self.buf, self._addr(btree)
        if b[p:p + 4] == b"SNOD":
            n = struct.unpack_from("<H", b, p + 6)[0]
            q = p + 8
            for _ in range(n):
                out.append((self._name(heap_data, self._off(q)), self._off(q + self.O)))
                q += 2 * self.O + 24
            return
        if b[p:p + 4] != b"TREE" or b[p + 4] != 0:
            raise Hdf5FormatError("bad group B-tree node")
        used = struct.unpack_from("<H", b, p + 6)[0]
        q = p + 8 + 2 * self.O + self.L                 # past the siblings and key 0
        for _ in range(used):
            self._group_entries(self._off(q), heap_data, out)
            q += self.O + self.L
        return

    def close(self):
        """unmap the file (views handed out by read() must be copied before)"""
        if self.buf is not None:
            try:
                self.buf.close()
            except BufferError:      # a caller still holds a view of the mapping: it goes with the last view
                pass
            self.buf = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def datasets(self) -> Dict[str, int]:
        """name -> object header address of the root group's members, in the group's (name-sorted) order."""
        if self._datasets is None:
            sym = [m for m in self._messages(self.root_header) if m[0] == 0x11]
            if not sym:
                raise Hdf5FormatError("the root group is not an old-style (symbol table) group; install h5py")
            data = sym[0][1]
            entries: List[Tuple[str, int]] = []
            self._group_entries(self._off(data), self._heap_data(self._off(data + self.O)), entries)
            self._datasets = dict(entries)
        return self._datasets

    def keys(self) -> List[str]:
        return list(self.datasets().keys())

    # ---- datasets
    def _dtype(self, p: int) -> np.dtype:
        b = self.buf
        cls, bits0 = b[p] & 0x0F, b[p + 1]
        size = struct.unpack_from("<I", b, p + 4)[0]
        order = ">" if (bits0 & 1) else "<"
        if cls == 1:
            kind = {2: "f2", 4: "f4", 8: "f8"}.get(size)
        elif cls == 0:
            kind = {1: "1", 2: "2", 4: "4", 8: "8"}.get(size)
            kind = (("i" if (bits0 & 8) else "u") + kind) if kind else None
        else:
            kind = None
        if kind is None:
            raise Hdf5FormatError(f"element type class {cls} of {size} bytes is not supported by the built-in reader")
        return np.dtype(order + kind if size > 1 else kind)

    def _describe(self, name: str):
        """(shape, element type, offset of the data layout message) of a dataset, from its object header alone."""
        b = self.buf
        shape, dtype, layout = None, None, None
        for mtype, data, _size in self._messages(self.datasets()[name]):
            if mtype == 0x01:                           # dataspace
                ver, rank = b[data], b[data + 1]
                q = data + (8 if ver == 1 else 4)
                shape = tuple(self._len(q + i * self.L) for i in range(rank))
            elif mtype == 0x03:
                dtype = self._dtype(data)
            elif mtype == 0x08:
                layout = data
            elif mtype == 0x0B:
                raise Hdf5FormatError(f"dataset {name!r} uses a filter pipeline (compression); install h5py to read it")
        if shape is None or dtype is None or layout is None:
            raise Hdf5FormatError(f"{name!r} is not a simple dataset")
        return shape, dtype, layout

    def shape(self, name: str) -> Tuple[int, ...]:
        return self._describe(name)[0]

    def read(self, name: str) -> np.ndarray:
        b = self.buf
        shape, dtype, layout = self._describe(name)
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if b[layout] != 3:
            raise Hdf5FormatError(f"data layout message version {b[layout]} not supported")
        cls = b[layout + 1]
        if cls == 1:                                    # contiguous
            addr = self._off(layout + 2)
            if addr == _UNDEF >> (64 - 8 * self.O):
                return np.zeros(shape, dtype=dtype.newbyteorder("="))
            return np.frombuffer(b, dtype=dtype, count=count, offset=self._addr(addr)).reshape(shape)
        if cls == 0:                                    # compact
            return np.frombuffer(b, dtype=dtype, count=count, offset=layout + 4).reshape(shape)
        if cls == 2:                                    # chunked, no filters
            rank1 = b[layout + 2]
            bt = self._off(layout + 3)
            cdims = struct.unpack_from("<%dI" % rank1, b, layout + 3 + self.O)
            chunk = tuple(cdims[:-1])
            if len(chunk) != len(shape) or cdims[-1] != dtype.itemsize:
                raise Hdf5FormatError("inconsistent chunk description")
            out = np.zeros(shape, dtype=dtype)
            if bt != _UNDEF >> (64 - 8 * self.O):
                self._read_chunks(bt, chunk, out)
            return out
        raise Hdf5FormatError(f"data layout class {cls} not supported")

    def _read_chunks(self, node: int, chunk: Tuple[int, ...], out: np.ndarray):
        b, p = self.buf, self._addr(node)
        if b[p:p + 4] != b"TREE" or b[p + 4] != 1:
            raise Hdf5FormatError("bad chunk B-tree node")
        level, used = b[p + 5], struct.unpack_from("<H", b, p + 6)[0]
        rank = len(chunk)
        key = 8 + 8 * (rank + 1)
        q = p + 8 + 2 * self.O
        n = int(np.prod(chunk))
        for _ in range(used):
            nbytes, mask = struct.unpack_from("<II", b, q)
            offs = struct.unpack_from("<%dQ" % rank, b, q + 8)
            child = self._off(q + key)
            if level > 0:
                self._read_chunks(child, chunk, out)
            else:
                if mask != 0 or nbytes != n * out.dtype.itemsize:
                    raise Hdf5FormatError("filtered chunk; install h5py to read this archive")
                blk = np.frombuffer(b, dtype=out.dtype, count=n, offset=self._addr(child)).reshape(chunk)
                sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, out.shape))
                out[sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]
            q += key + self.O


class _Archive:
    """The datasets of an archive in its iteration order: through h5py when it is importable, else MiniHdf5File."""

    def __init__(self, filename: str):
        try:
            import h5py
            self._h5, self._mini = h5py.File(filename, "r"), None
            self.names = list(self._h5.keys())
        except ImportError:
            self._h5, self._mini = None, MiniHdf5File(filename)
            self.names = self._mini.keys()

    def close(self):
        if self._h5 is not None:
            self._h5.close()
        if self._mini is not None:
            self._mini.close()
        self._h5 = self._mini = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def shape(self, name: str) -> Tuple[int, ...]:
        return tuple(self._h5[name].shape) if self._h5 is not None else self._mini.shape(name)

    def rows(self, name: str, dim: int) -> np.ndarray:
        """the dataset as a (frames, dim) matrix"""
        a = self._h5[name][:] if self._h5 is not None else self._mini.read(name)
        return np.ascontiguousarray(a).reshape(-1, dim)


def read_hdf5_data(filename: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """(train, valid) as the reference's helper of the same name returns them (quantization.py:746-820): all datasets of the
    archive stacked into one float16 matrix of frames, its rows shuffled with numpy's GLOBAL generator (so a caller's
    np.random.seed gives the reference's split), the first 5 % of the shuffled rows -- 10,000 at most -- as `valid`, the rest as
    `train`.  Both are CPU tensors viewing that one matrix."""
    logging.info(f"Opening file {filename}")
    with _Archive(filename) as archive:      # (closed -- file unmapped -- once the frames are copied out)
        # one pass over the headers: how many frames each dataset brings (everything but the last axis), one common feature dimension
        shapes = [archive.shape(name) for name in archive.names]
        dims = {sh[-1] for sh in shapes}
        assert len(dims) <= 1, "Dataset must have consistent dimension (last element of shape"
        dim = dims.pop() if dims else -1
        counts = [int(np.prod(sh[:-1], dtype=np.int64)) for sh in shapes]
        starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        total = int(starts[-1])
        logging.info(f"read_data: tot_frames = {total}")
        frames = np.empty((total, dim), dtype=np.float16)
        for name, lo, n in zip(archive.names, starts[:-1], counts):             # second pass: the data, dataset after dataset
            block = archive.rows(name, dim)
            assert block.shape[0] == n, (name, block.shape, n)
            frames[lo:lo + n] = block
            del block
    np.random.shuffle(frames)
    n_valid = min(int(0.05 * total), 10000)      # (the reference slices with the float itself: module docstring)
    logging.info(f"read_data: train_frames={total - n_valid}, valid_frames={n_valid}")
    rows = torch.from_numpy(frames)
    return rows[n_valid:], rows[:n_valid]
