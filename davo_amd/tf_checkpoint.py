"""Reader (and minimal writer) for TensorFlow V2 checkpoints ("tensor bundles") — row f3.

The reference restores its weights with ``tf.train.Saver(tf.trainable_variables()).restore(sess,
ckpt_file)`` (test_kitti_pose.py:129-131) from the files ``tf.train.Saver.save`` wrote
(davo.py:865-867,1572-1575): ``model-<step>.index`` + ``model-<step>.data-00000-of-00001``, found
through the ``checkpoint`` text file (run_inference.sh:28-40).  This module reads those files
without TensorFlow and returns ``{variable name: float32 ndarray}`` for ``DAVO.load_weights``.

Format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table = LevelDB's SSTable):
* ``.index``: SSTable.  48-byte footer = BlockHandle(metaindex) + BlockHandle(index), padded to
  40 bytes, + magic 0xdb4775248b80fb57 LE.  A block = entries + uint32 restarts[] + uint32
  n_restarts, followed on disk by a 5-byte trailer (compression type, masked crc32c).  An entry =
  varint shared, varint non_shared, varint value_len, key delta, value (prefix-compressed keys).
  Index-block values are BlockHandles of the data blocks.
* key "" -> BundleHeaderProto {num_shards=1, endianness=2, version=3}; key <var name> ->
  BundleEntryProto {dtype=1, shape=2 {dim=2 {size=1}}, shard_id=3, offset=4, size=5, crc32c=6}.
* ``.data-SSSSS-of-NNNNN``: raw little-endian tensor bytes at [offset, offset+size).

No TF checkpoint exists offline, so this reader is validated against bundles written by
``write_checkpoint`` below (tests/test_tf_checkpoint.py) — format knowledge, not a TF round trip.
"""
import os
import re
import struct

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DT = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 4: np.uint8, 10: np.bool_}
_DT_INV = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}


# ---- varints / protobuf wire format ---------------------------------------------------------
def _get_varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf):
    """-> {field number: [values]}; length-delimited values stay bytes, fixed32/64 are ints."""
    out, pos = {}, 0
    while pos < len(buf):
        key, pos = _get_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n]); pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(v)
    return out


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


# ---- crc32c (Castagnoli), as LevelDB masks it -------------------------------------------------
_CRC_TABLE = None


def crc32c(data):
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    c = 0xFFFFFFFF
    for b in bytes(data):
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- snappy (raw format) decoder: TF writes the index uncompressed, but tolerate type 1 ---------
def _snappy_decompress(src):
    n, pos = _get_varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]; pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], "little"); pos += nb
            ln += 1
            out += src[pos:pos + ln]; pos += ln
        else:
            if kind == 1:
                ln = ((tag >> 2) & 7) + 4
                off = ((tag >> 5) << 8) | src[pos]; pos += 1
            elif kind == 2:
                ln = (tag >> 2) + 1
                off = src[pos] | (src[pos + 1] << 8); pos += 2
            else:
                ln = (tag >> 2) + 1
                off = int.from_bytes(src[pos:pos + 4], "little"); pos += 4
            for _ in range(ln):
                out.append(out[-off])
    if len(out) != n:
        raise ValueError("corrupt snappy block")
    return bytes(out)


# ---- SSTable ----------------------------------------------------------------------------------
def _read_block(f, offset, size, verify):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) != size + 5:
        raise ValueError("truncated table block")
    body, ctype, crc = raw[:size], raw[size], struct.unpack_from("<I", raw, size + 1)[0]
    if verify and _mask_crc(crc32c(raw[:size + 1])) != crc:
        raise ValueError("table block checksum mismatch")
    if ctype == 1:
        body = _snappy_decompress(body)
    elif ctype != 0:
        raise ValueError("unknown block compression type %d" % ctype)
    return body


def _block_entries(block):
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _table_items(path, verify=True):
    with open(path, "rb") as f:
        f.seek(0, os.SEEK_END)
        size = f.tell()
        if size < 48:
            raise ValueError("%s: too small to be a checkpoint index" % path)
        f.seek(size - 48)
        footer = f.read(48)
        if struct.unpack_from("<Q", footer, 40)[0] != _MAGIC:
            raise ValueError("%s: not a TF checkpoint index (bad magic)" % path)
        pos = 0
        _, pos = _get_varint(footer, pos); _, pos = _get_varint(footer, pos)      # metaindex handle
        idx_off, pos = _get_varint(footer, pos); idx_size, pos = _get_varint(footer, pos)
        for _, handle in _block_entries(_read_block(f, idx_off, idx_size, verify)):
            off, p2 = _get_varint(handle, 0)
            sz, _ = _get_varint(handle, p2)
            for k, v in _block_entries(_read_block(f, off, sz, verify)):
                yield k, v


# ---- public API ---------------------------------------------------------------------------------
def resolve_checkpoint(path):
    """Accept 'dir' (reads dir/checkpoint, like run_inference.sh:28-40), 'prefix', 'prefix.index'."""
    if os.path.isdir(path):
        state = os.path.join(path, "checkpoint")
        if not os.path.exists(state):
            raise FileNotFoundError("%s has no `checkpoint' file" % path)
        m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', open(state).read())
        if not m:
            raise ValueError("%s: no model_checkpoint_path" % state)
        p = m.group(1)
        return p if os.path.isabs(p) else os.path.join(path, p)
    return path[:-len(".index")] if path.endswith(".index") else path


def list_variables(prefix):
    """[(name, shape, dtype)] like tf.train.list_variables."""
    prefix = resolve_checkpoint(prefix)
    out = []
    for k, v in _table_items(prefix + ".index"):
        if not k:
            continue
        e = _parse_proto(v)
        shape = [_signed64(_parse_proto(d).get(1, [0])[0]) for d in _parse_proto(e.get(2, [b""])[0]).get(2, [])]
        out.append((k.decode(), tuple(shape), _DT.get(e.get(1, [0])[0])))
    return out


def read_checkpoint(prefix, names=None, verify_crc=False):
    """{variable name: ndarray} (float tensors as float32).  ``names`` restricts the set (default:
    every variable; slot variables such as '.../Adam' are returned too if the file has them)."""
    prefix = resolve_checkpoint(prefix)
    entries, num_shards = {}, 1
    for k, v in _table_items(prefix + ".index"):
        msg = _parse_proto(v)
        if not k:
            num_shards = msg.get(1, [1])[0]
            if msg.get(2, [0])[0] != 0:
                raise ValueError("big-endian checkpoints are not supported")
            continue
        entries[k.decode()] = msg
    want = list(entries) if names is None else list(names)
    out, files = {}, {}
    try:
        for name in want:
            if name not in entries:
                raise KeyError("checkpoint %s has no variable `%s'" % (prefix, name))
            e = entries[name]
            dt = _DT.get(e.get(1, [0])[0])
            if names is None and (7 in e or dt is None):
                continue            # reading everything: leave out what the path never needs (string/sliced entries)
            if 7 in e:
                raise ValueError("`%s' is a sliced (partitioned) variable: not supported" % name)
            if dt is None:
                raise ValueError("`%s': unsupported dtype enum %s" % (name, e.get(1)))
            shape = [_signed64(_parse_proto(d).get(1, [0])[0]) for d in _parse_proto(e.get(2, [b""])[0]).get(2, [])]
            shard, off, size = e.get(3, [0])[0], e.get(4, [0])[0], e.get(5, [0])[0]
            if shard not in files:
                files[shard] = open("%s.data-%05d-of-%05d" % (prefix, shard, num_shards), "rb")
            f = files[shard]
            f.seek(off)
            raw = f.read(size)
            if len(raw) != size or size != int(np.prod(shape, dtype=np.int64)) * np.dtype(dt).itemsize:
                raise ValueError("`%s': data size %d does not match shape %s" % (name, len(raw), shape))
            if verify_crc and 6 in e and _mask_crc(crc32c(raw)) != e[6][0]:
                raise ValueError("`%s': crc32c mismatch" % name)
            a = np.frombuffer(raw, dtype=np.dtype(dt).newbyteorder("<")).reshape(shape)
            out[name] = a.astype(np.float32) if a.dtype.kind == "f" else a.copy()
    finally:
        for f in files.values():
            f.close()
    return out


def load_weights(path):
    """What run_kitti_pose --ckpt_file accepts: an .npz keyed by TF names, or a TF V2 checkpoint."""
    if path.endswith(".npz"):
        return dict(np.load(path))
    return read_checkpoint(path)


# ---- writer (tests / export): uncompressed, same on-disk structure as TF's ------------------------
def _build_block(items, restart_interval=16):
    buf, restarts, last, n = bytearray(), [], b"", 0
    for k, v in items:
        shared = 0
        if n % restart_interval == 0:
            restarts.append(len(buf))
        else:
            while shared < min(len(last), len(k)) and last[shared] == k[shared]:
                shared += 1
        buf += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last, n = k, n + 1
    if not restarts:
        restarts = [0]
    for r in restarts:
        buf += struct.pack("<I", r)
    buf += struct.pack("<I", len(restarts))
    return bytes(buf)


def write_checkpoint(prefix, variables, block_entries=4, num_shards=1):
    """Write {name: ndarray} as <prefix>.index / .data-SSSSS-of-NNNNN and a `checkpoint' state file.
    num_shards > 1 spreads the tensors round-robin over that many data files, as a sharded Saver does."""
    names = sorted(variables)
    data = [bytearray() for _ in range(num_shards)]
    items = [(b"", b"\x08" + _put_varint(num_shards) + b"\x1a\x02\x08\x01")]      # num_shards, version{producer=1}
    for i, name in enumerate(names):
        a = np.asarray(variables[name], order="C")          # (ascontiguousarray would turn a scalar into shape (1,))
        raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
        shape = b"".join(b"\x12" + _put_varint(len(d)) + d for d in
                         (b"\x08" + _put_varint(int(s)) for s in a.shape))
        e = b"\x08" + _put_varint(_DT_INV[a.dtype]) + b"\x12" + _put_varint(len(shape)) + shape
        shard = i % num_shards
        if shard:
            e += b"\x18" + _put_varint(shard)
        if len(data[shard]):
            e += b"\x20" + _put_varint(len(data[shard]))
        e += b"\x28" + _put_varint(len(raw)) + b"\x35" + struct.pack("<I", _mask_crc(crc32c(raw)))
        items.append((name.encode(), e))
        data[shard] += raw
    for sh in range(num_shards):
        with open("%s.data-%05d-of-%05d" % (prefix, sh, num_shards), "wb") as f:
            f.write(bytes(data[sh]))
    with open(prefix + ".index", "wb") as f:
        index = []

        def emit(block):
            off = f.tell()
            trailer = b"\x00"
            f.write(block + trailer + struct.pack("<I", _mask_crc(crc32c(block + trailer))))
            return _put_varint(off) + _put_varint(len(block))
        for i in range(0, len(items), block_entries):
            chunk = items[i:i + block_entries]
            index.append((chunk[-1][0], emit(_build_block(chunk))))
        meta = emit(_build_block([]))
        idx = emit(_build_block(index, restart_interval=1))
        footer = meta + idx
        f.write(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC))
    with open(os.path.join(os.path.dirname(prefix) or ".", "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n'
                % (os.path.basename(prefix), os.path.basename(prefix)))
