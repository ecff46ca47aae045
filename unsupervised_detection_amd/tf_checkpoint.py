"""TF-1 "V2" checkpoints (tf.train.Saver: `<prefix>.index` + `<prefix>.data-00000-of-0000N`) without TensorFlow.

The reference restores `pwcnet.ckpt-595000`, the recover net's `model-175` and `model.best` with tf.train.Saver
(train.py:19, test_generator.py:45-55, models/adversarial_learner.py:339-360).  This module reads (and writes) that on-disk
format directly, so released checkpoints load into the flat buffers of weights.py and trained weights can be handed back to
the reference.  SURVEY section 8f, row N3.

Format (tensorflow/core/util/tensor_bundle/tensor_bundle.cc, tensorflow/core/lib/io/{table_builder,block,format}.cc):
  * `.index` is an SSTable in LevelDB's table format, written by BundleWriter with compression off:
      data blocks | metaindex block | index block | 48-byte footer
    block   = entries, restart offsets (uint32 LE each), number of restarts (uint32 LE); followed by a 5-byte trailer
              (compression type byte: 0 none / 1 snappy; masked CRC32C of block + type byte, uint32 LE)
    entry   = varint32 shared key bytes, varint32 unshared key bytes, varint32 value length, key suffix, value
    footer  = metaindex BlockHandle, index BlockHandle (varint64 offset, varint64 size each), zero padding to 40 bytes,
              magic 0xdb4775248b80fb57 (LE)
    index block entries: key >= last key of a data block, value = that block's BlockHandle
  * key ""   -> BundleHeaderProto  {1: num_shards, 2: endianness (0 little), 3: VersionDef}
    key name -> BundleEntryProto   {1: dtype, 2: TensorShapeProto {2: Dim {1: size}}, 3: shard_id, 4: offset, 5: size,
                                    6: fixed32 masked CRC32C of the tensor bytes, 7: slices (partitioned variables: unsupported)}
  * `.data-*` shards hold the raw little-endian tensor bytes at [offset, offset + size).

NOT verified against a file written by TensorFlow (none exists in this environment, TF is not installable): the reader is
checked against the writer below and both check every CRC, so a layout mistake shows up as a checksum / parse error, not as
silently wrong weights."""
from __future__ import annotations

import os
import struct
from collections import OrderedDict

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
MASK_DELTA = 0xA282EAD8
# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_, 19: np.float16}
DTYPE_IDS = {np.dtype(v): k for k, v in DTYPES.items()}


# ------------------------------------------------------------------------------------------------ CRC32C (Castagnoli)
def _make_crc_table():
    tab = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab[i] = c
    return tab


_CRC_TABLE = _make_crc_table()


def _update_scalar(reg: int, data) -> int:
    tab = _CRC_TABLE
    for b in data:
        reg = int(tab[(reg ^ b) & 0xFF]) ^ (reg >> 8)
    return reg


def crc32c(data) -> int:
    """CRC-32C (Castagnoli, reflected, init / final xor 0xFFFFFFFF) of a bytes-like object or array.
    Large inputs are cut into K equal chunks whose registers advance in lock step (numpy lanes); the chunk results are
    chained with the linear map Z that advances a register through one chunk length of zero bytes:
        U(r, chunk) = U(0, chunk) xor Z(r)      (the table-driven update U is affine in the start register r)."""
    buf = np.frombuffer(bytes(data) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).tobytes(), np.uint8)
    n = buf.size
    K = 4096
    if n < 8 * K:
        return _update_scalar(0xFFFFFFFF, buf.tolist()) ^ 0xFFFFFFFF
    L = n // K
    tab = _CRC_TABLE
    lanes = buf[:K * L].reshape(K, L)
    reg = np.zeros(K, np.uint32)
    for i in range(L):  # every chunk from a zero register
        reg = tab[(reg ^ lanes[:, i]) & 0xFF] ^ (reg >> np.uint32(8))
    z = np.uint32(1) << np.arange(32, dtype=np.uint32)  # images of the 32 unit registers under L zero bytes
    for _ in range(L):
        z = tab[z & 0xFF] ^ (z >> np.uint32(8))
    zcols = [int(v) for v in z]
    r = 0xFFFFFFFF
    for a in reg.tolist():
        adv, bit = 0, 0
        while r:
            if r & 1:
                adv ^= zcols[bit]
            r >>= 1
            bit += 1
        r = adv ^ a
    r = _update_scalar(r, buf[K * L:].tolist())
    return r ^ 0xFFFFFFFF


def mask_crc(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked: int) -> int:
    rot = (masked - MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ varints / protobuf subset
def _get_varint(buf, pos):
    shift, val = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 63:
            raise ValueError("malformed varint")


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf: bytes):
    """{field number: [values]} for wire types 0 (varint), 1 (fixed64), 2 (bytes), 5 (fixed32)."""
    out, pos = {}, 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.setdefault(field, []).append(v)
    return out


def _field(field: int, wt: int, payload) -> bytes:
    tag = _put_varint(field << 3 | wt)
    if wt == 0:
        return tag + _put_varint(int(payload))
    if wt == 2:
        return tag + _put_varint(len(payload)) + payload
    if wt == 5:
        return tag + struct.pack("<I", payload)
    raise ValueError(wt)


# ------------------------------------------------------------------------------------------------ SSTable reading
def _read_block(f: bytes, offset: int, size: int) -> bytes:
    block, trailer = f[offset:offset + size], f[offset + size:offset + size + 5]
    if len(block) != size or len(trailer) != 5:
        raise ValueError("index file truncated")
    if trailer[0] != 0:
        raise NotImplementedError("compressed table block (type %d): BundleWriter writes uncompressed tables" % trailer[0])
    want = unmask_crc(struct.unpack("<I", trailer[1:])[0])
    if crc32c(bytes(block) + bytes(trailer[:1])) != want:
        raise ValueError("table block checksum mismatch at offset %d" % offset)
    return block


def _block_entries(block: bytes):
    if len(block) < 4:
        raise ValueError("table block too short")
    nrestarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        unshared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _handle(buf: bytes, pos: int = 0):
    off, pos = _get_varint(buf, pos)
    size, pos = _get_varint(buf, pos)
    return off, size, pos


def read_index(prefix: str):
    """-> (header dict, OrderedDict name -> {dtype, shape, shard, offset, size, crc})."""
    with open(prefix + ".index", "rb") as fh:
        f = fh.read()
    if len(f) < 48 or struct.unpack("<Q", f[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s.index is not a TensorBundle index (bad magic)" % prefix)
    footer = f[-48:]
    _, _, pos = _handle(footer, 0)                 # metaindex (empty)
    ioff, isize, _ = _handle(footer, pos)
    entries = OrderedDict()
    header = None
    for _, hv in _block_entries(_read_block(f, ioff, isize)):
        boff, bsize, _ = _handle(hv)
        for key, val in _block_entries(_read_block(f, boff, bsize)):
            msg = _parse_proto(val)
            if key == b"":
                header = {"num_shards": msg.get(1, [1])[0], "endianness": msg.get(2, [0])[0]}
                continue
            if 7 in msg:
                raise NotImplementedError("partitioned variable %s (tensor slices)" % key.decode())
            dims = []
            for shp in msg.get(2, []):
                for d in _parse_proto(shp).get(2, []):
                    dims.append(_parse_proto(d).get(1, [0])[0])
            entries[key.decode()] = {"dtype": msg.get(1, [0])[0], "shape": tuple(dims), "shard": msg.get(3, [0])[0],
                                     "offset": msg.get(4, [0])[0], "size": msg.get(5, [0])[0], "crc": msg.get(6, [0])[0]}
    if header is None:
        raise ValueError("bundle header entry missing")
    if header["endianness"] != 0:
        raise NotImplementedError("big-endian bundle")
    return header, entries


def read_checkpoint(prefix: str, names=None, verify: bool = True):
    """{variable name: numpy array} of a V2 checkpoint `prefix` (`prefix.index`, `prefix.data-0000k-of-0000n`)."""
    header, entries = read_index(prefix)
    n = header["num_shards"]
    shards = {}
    out = OrderedDict()
    for name, e in entries.items():
        if names is not None and name not in names:
            continue
        if e["dtype"] not in DTYPES:
            raise NotImplementedError("%s: dtype enum %d" % (name, e["dtype"]))
        if e["shard"] not in shards:
            shards[e["shard"]] = np.memmap("%s.data-%05d-of-%05d" % (prefix, e["shard"], n), dtype=np.uint8, mode="r")
        raw = np.asarray(shards[e["shard"]][e["offset"]:e["offset"] + e["size"]])
        dt = np.dtype(DTYPES[e["dtype"]])
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if raw.size != count * dt.itemsize:
            raise ValueError("%s: %d bytes on disk, shape %s needs %d" % (name, raw.size, e["shape"], count * dt.itemsize))
        if verify and unmask_crc(e["crc"]) != crc32c(raw.tobytes()):
            raise ValueError("%s: tensor checksum mismatch" % name)
        out[name] = raw.view(dt).reshape(e["shape"]).copy()
    return out


# ------------------------------------------------------------------------------------------------ writing
def _build_block(items, restart_interval: int = 16) -> bytes:
    out, restarts, last = bytearray(), [], b""
    for i, (key, val) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            m = min(len(last), len(key))
            while shared < m and last[shared] == key[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(val)) + key[shared:] + val
        last = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def write_checkpoint(prefix: str, tensors, block_entries: int = 64):
    """Write {name: array} as a single-shard V2 checkpoint (uncompressed table, CRCs as TensorFlow computes them)."""
    names = sorted(tensors)  # table keys are ordered bytewise
    data = bytearray()
    items = [(b"", _field(1, 0, 1) + _field(2, 0, 0) + _field(3, 2, _field(1, 0, 1)))]  # header: 1 shard, little endian, producer 1
    for name in names:
        a = np.asarray(tensors[name])  # (np.ascontiguousarray would turn a scalar into shape (1,))
        if a.dtype not in DTYPE_IDS:
            raise NotImplementedError("%s: dtype %s" % (name, a.dtype))
        raw = a.tobytes(order="C")
        shape = b"".join(_field(2, 2, _field(1, 0, d)) for d in a.shape)
        entry = _field(1, 0, DTYPE_IDS[a.dtype]) + _field(2, 2, shape)
        if len(data):
            entry += _field(4, 0, len(data))
        entry += _field(5, 0, len(raw)) + _field(6, 5, mask_crc(crc32c(raw)))
        items.append((name.encode(), entry))
        data += raw
    table, index_items = bytearray(), []

    def emit(block: bytes):
        off = len(table)
        table.extend(block + b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return _put_varint(off) + _put_varint(len(block))

    for i in range(0, len(items), block_entries):
        chunk = items[i:i + block_entries]
        index_items.append((chunk[-1][0], emit(_build_block(chunk))))
    meta = emit(_build_block([]))
    index = emit(_build_block(index_items, restart_interval=1))
    footer = meta + index
    table.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(table))
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))


def tf_variable_name(canonical: str) -> str:
    """Name under which the reference's Saver stores a variable of weights.param_table() (inverse of weights.canonical_name)."""
    from .weights import _GEN_TOP_BN_ORDER
    parts = canonical.split("/")
    if parts[0] == "MaskNet" and len(parts) == 4 and parts[2] == "bn":
        layer = parts[1]
        if layer.endswith("_upsample"):
            return "MaskNet//%s/batch_normalization/%s" % (layer, parts[3])
        k = _GEN_TOP_BN_ORDER.index(layer)
        return "MaskNet//batch_normalization%s/%s" % ("_%d" % k if k else "", parts[3])
    if parts[0] in ("MaskNet", "FlownetS"):
        return parts[0] + "//" + "/".join(parts[1:])
    return canonical
