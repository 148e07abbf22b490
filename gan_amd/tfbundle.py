"""TensorFlow TensorBundle container, written and read without TensorFlow: the on-disk format
of `tf.train.Checkpoint.save` / `CheckpointManager` that the reference uses (pix2pix.py:400-403,419-420;
cycle_gan.py:437-444,460-461):

  <prefix>.data-00000-of-00001   tensor bytes back to back (little endian)
  <prefix>.index                 an immutable sorted string table (the LevelDB/TF `table` format): key "" ->
                                 BundleHeaderProto, variable key -> BundleEntryProto {dtype, shape, shard, offset, size,
                                 masked crc32c}, plus `_CHECKPOINTABLE_OBJECT_GRAPH` -> serialized TrackableObjectGraph
                                 (a scalar DT_STRING tensor) that tf.train.Checkpoint.restore walks by child name.

Restated from the published formats (tensorflow/core/util/tensor_bundle, core/lib/io/{table,block,format},
core/protobuf/{tensor_bundle,trackable_object_graph}.proto, core/framework/{types,tensor_shape}.proto of the pinned
tensorflow==2.6.0).  Format parity is UNPINNED: the reference ships no TF-written checkpoint and TensorFlow is not
installable here, so this is checked by its own reader, by structural invariants (magic, block checksums, sorted keys)
and by the field numbers above - not against a file TensorFlow wrote.
"""
from __future__ import annotations

import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
BLOCK_SIZE, RESTART_INTERVAL = 262144, 16
# tensorflow/core/framework/types.proto
DT = {np.dtype('float32'): 1, np.dtype('float64'): 2, np.dtype('int32'): 3, np.dtype('uint8'): 4, np.dtype('int64'): 9,
      np.dtype('bool'): 10, np.dtype('float16'): 19}
DT_STRING = 7
DT_INV = {v: k for k, v in DT.items()}
OBJECT_GRAPH_KEY = '_CHECKPOINTABLE_OBJECT_GRAPH'


def _crc32c(data, crc=0):
    from . import _lib
    lib = _lib.load()
    mv = memoryview(data).cast('B')
    if len(mv) == 0:
        return crc
    buf = np.frombuffer(mv, dtype=np.uint8)
    return lib.gan_crc32c(crc, buf.ctypes.data, buf.size)


def _mask(crc):
    """crc32c::Mask: rotate right by 15 and add a constant (checksums of data that itself holds checksums)."""
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + 0xa282ead8) & 0xffffffff


def _unmask(m):
    rot = (m - 0xa282ead8) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- protobuf wire format (only what the three messages need) ---------------------------------------------------
def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _read_varint(buf, pos):
    v = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7f) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _field(num, wire, payload):
    return _varint((num << 3) | wire) + payload


def _f_varint(num, v):
    return _field(num, 0, _varint(v))


def _f_bytes(num, b):
    b = b.encode() if isinstance(b, str) else bytes(b)
    return _field(num, 2, _varint(len(b)) + b)


def _parse(buf):
    """-> [(field number, wire type, value)] of one message (value: int for varint / fixed, bytes for length-delimited)."""
    out, pos = [], 0
    while pos < len(buf):
        tag, pos = _read_varint(buf, pos)
        num, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v, pos = bytes(buf[pos:pos + n]), pos + n
        elif wire == 5:
            v, pos = struct.unpack_from('<I', buf, pos)[0], pos + 4
        elif wire == 1:
            v, pos = struct.unpack_from('<Q', buf, pos)[0], pos + 8
        else:
            raise ValueError(f"unsupported wire type {wire}")
        out.append((num, wire, v))
    return out


def _shape_proto(shape):
    """TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }."""
    return b''.join(_f_bytes(2, _f_varint(1, int(d))) for d in shape)


def _entry_proto(dtype_enum, shape, offset, size, crc_masked):
    """BundleEntryProto: dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6 (fixed32)."""
    msg = _f_varint(1, dtype_enum) + _f_bytes(2, _shape_proto(shape))
    if offset:
        msg += _f_varint(4, offset)
    msg += _f_varint(5, size) + _field(6, 5, struct.pack('<I', crc_masked))
    return msg


def _header_proto():
    """BundleHeaderProto: num_shards = 1, endianness = 2 (LITTLE = 0, omitted), version = 3 { producer = 1 }."""
    return _f_varint(1, 1) + _f_bytes(3, _f_varint(1, 1))


# ---- the table (sorted string table with prefix-compressed blocks) ---------------------------------------------
class _BlockBuilder:
    def __init__(self):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b''

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count % RESTART_INTERVAL == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last, self.count = key, self.count + 1

    def finish(self) -> bytes:
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4


def _write_block(f, contents: bytes):
    """block + trailer {compression type 0, masked crc32c(contents + type)}; -> (offset, size) handle."""
    off = f.tell()
    f.write(contents)
    f.write(b'\x00' + struct.pack('<I', _mask(_crc32c(contents + b'\x00'))))
    return off, len(contents)


def write_table(path, items):
    """items: iterable of (key bytes, value bytes), strictly increasing keys."""
    with open(path, 'wb') as f:
        index, data, last_key = _BlockBuilder(), _BlockBuilder(), None
        for k, v in items:
            assert last_key is None or k > last_key, "table keys must be strictly increasing"
            data.add(k, v)
            last_key = k
            if data.size() >= BLOCK_SIZE:
                off, size = _write_block(f, data.finish())
                index.add(last_key, _varint(off) + _varint(size))       # separator = the block's last key
                data = _BlockBuilder()
        if data.count:
            off, size = _write_block(f, data.finish())
            index.add(last_key, _varint(off) + _varint(size))
        meta = _write_block(f, _BlockBuilder().finish())               # no filter / properties: an empty metaindex block
        idx = _write_block(f, index.finish())
        footer = _varint(meta[0]) + _varint(meta[1]) + _varint(idx[0]) + _varint(idx[1])
        f.write(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC))


def _read_block(buf, off, size, verify=True):
    contents = bytes(buf[off:off + size])
    ctype, crc = buf[off + size], struct.unpack_from('<I', buf, off + size + 1)[0]
    if ctype != 0:
        raise ValueError("compressed table blocks are not supported (TensorBundle indexes are written uncompressed)")
    if verify and _unmask(crc) != _crc32c(contents + b'\x00'):
        raise ValueError("table block checksum mismatch")
    n_restarts = struct.unpack_from('<I', contents, len(contents) - 4)[0]
    end = len(contents) - 4 - 4 * n_restarts
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _read_varint(contents, pos)
        non_shared, pos = _read_varint(contents, pos)
        vlen, pos = _read_varint(contents, pos)
        key = key[:shared] + contents[pos:pos + non_shared]
        pos += non_shared
        out.append((key, contents[pos:pos + vlen]))
        pos += vlen
    return out


def read_table(path):
    buf = open(path, 'rb').read()
    if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError(f"{path}: not a TensorFlow table file (bad magic)")
    foot = buf[-48:]
    _, p = _read_varint(foot, 0)
    _, p = _read_varint(foot, p)
    ioff, p = _read_varint(foot, p)
    isize, p = _read_varint(foot, p)
    items = []
    for _, handle in _read_block(buf, ioff, isize):
        off, q = _read_varint(handle, 0)
        size, _ = _read_varint(handle, q)
        items += _read_block(buf, off, size)
    return items


# ---- the object graph ---------------------------------------------------------------------------------------------
def object_graph(variables, slots):
    """TrackableObjectGraph for `variables` = {checkpoint key: full variable name} and `slots` = [(optimizer object
    name, slot name, original variable key, slot variable key)].  Every path component of a key (up to `.ATTRIBUTES`)
    becomes a node reached from its parent by `local_name`, which is how tf.train.Checkpoint.restore matches a saved graph
    against live objects; the variable node carries SerializedTensor{name 'VARIABLE_VALUE', full_name, checkpoint_key}."""
    nodes = [{'children': {}, 'attrs': [], 'slots': []}]            # node 0 = the root Checkpoint object

    def node_for(path):
        cur = 0
        for comp in path:
            nxt = nodes[cur]['children'].get(comp)
            if nxt is None:
                nodes.append({'children': {}, 'attrs': [], 'slots': []})
                nxt = len(nodes) - 1
                nodes[cur]['children'][comp] = nxt
            cur = nxt
        return cur

    var_node = {}
    for key in sorted(variables):
        path = key.split('/.ATTRIBUTES/')[0].split('/')
        if '.OPTIMIZER_SLOT' in path:
            continue
        n = node_for(path)
        nodes[n]['attrs'].append(('VARIABLE_VALUE', variables[key], key))
        var_node[key] = n
    for opt, slot, orig_key, slot_key in slots:
        nodes.append({'children': {}, 'attrs': [('VARIABLE_VALUE', variables.get(slot_key, ''), slot_key)], 'slots': []})
        nodes[node_for([opt])]['slots'].append((var_node[orig_key], slot, len(nodes) - 1))
    out = b''
    for nd in nodes:
        msg = b''
        for name, child in nd['children'].items():          # ObjectReference: node_id = 1, local_name = 2
            msg += _f_bytes(1, _f_varint(1, child) + _f_bytes(2, name))
        for name, full, key in nd['attrs']:                  # SerializedTensor: name = 1, full_name = 2, checkpoint_key = 3
            msg += _f_bytes(2, _f_bytes(1, name) + _f_bytes(2, full) + _f_bytes(3, key))
        for orig, slot, sv in nd['slots']:                   # SlotVariableReference: original = 1, slot_name = 2, slot node = 3
            msg += _f_bytes(3, _f_varint(1, orig) + _f_bytes(2, slot) + _f_varint(3, sv))
        out += _f_bytes(1, msg)                              # TrackableObjectGraph.nodes = 1
    return out


def parse_object_graph(blob):
    """-> [{children: {name: id}, keys: [checkpoint_key], slots: [(orig, slot, node)]}] (reader side, tests)."""
    nodes = []
    for num, _, msg in _parse(blob):
        if num != 1:
            continue
        nd = {'children': {}, 'keys': [], 'slots': []}
        for fnum, _, sub in _parse(msg):
            f = {n: v for n, _, v in _parse(sub)}
            if fnum == 1:
                nd['children'][f.get(2, b'').decode()] = f.get(1, 0)
            elif fnum == 2:
                nd['keys'].append(f.get(3, b'').decode())
            elif fnum == 3:
                nd['slots'].append((f.get(1, 0), f.get(2, b'').decode(), f.get(3, 0)))
        nodes.append(nd)
    return nodes


# ---- bundle ----------------------------------------------------------------------------------------------------------
def _string_lengths_crc(lengths):
    """Running crc32c over the element lengths as TensorFlow's WriteStringTensor extends it (tensor_bundle.cc): a length that
    fits 32 bits enters as a little-endian uint32 (kept for files written before 64-bit lengths existed), a longer one as a
    uint64 - NOT the varint bytes that are stored."""
    crc = 0
    for ln in lengths:
        crc = _crc32c(struct.pack('<I', ln) if ln <= 0xffffffff else struct.pack('<Q', ln), crc)
    return crc


def _string_tensor_bytes(values):
    """DT_STRING tensor data: [varint64 length]... [fixed32 masked crc32c of the lengths] [bytes]...; the entry's checksum
    keeps running from the length crc over the 4 stored checksum bytes and then the string bytes."""
    lens = b''.join(_varint(len(v)) for v in values)
    crc = _string_lengths_crc(len(v) for v in values)
    chk = struct.pack('<I', _mask(crc))
    crc = _crc32c(chk, crc)
    for v in values:
        crc = _crc32c(v, crc)
    return lens + chk + b''.join(values), crc


def write_bundle(prefix, arrays, graph_blob=None):
    """arrays: {checkpoint key: numpy array}; graph_blob: serialized TrackableObjectGraph (stored under
    _CHECKPOINTABLE_OBJECT_GRAPH).  One shard."""
    entries, off = {}, 0
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for key in sorted(arrays):
            a = np.asarray(arrays[key])
            a = a if a.flags.c_contiguous else a.copy(order='C')       # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in DT:
                raise TypeError(f"{key}: dtype {a.dtype} has no TensorFlow DataType mapping here")
            b = a.tobytes()
            f.write(b)
            entries[key] = _entry_proto(DT[a.dtype], a.shape, off, len(b), _mask(_crc32c(b)))
            off += len(b)
        if graph_blob is not None:
            b, crc = _string_tensor_bytes([graph_blob])
            f.write(b)
            entries[OBJECT_GRAPH_KEY] = _entry_proto(DT_STRING, (), off, len(b), _mask(crc))
            off += len(b)
    items = [(b'', _header_proto())] + [(k.encode(), entries[k]) for k in sorted(entries, key=lambda s: s.encode())]
    write_table(prefix + '.index', items)
    return prefix


def read_bundle(prefix, verify=True):
    """-> ({checkpoint key: numpy array}, object-graph blob or None)."""
    items = read_table(prefix + '.index')
    if not items or items[0][0] != b'':
        raise ValueError("TensorBundle index without a header entry")
    data = np.memmap(prefix + '.data-00000-of-00001', dtype=np.uint8, mode='r')
    arrays, graph = {}, None
    for k, v in items[1:]:
        f = {}
        for num, _, val in _parse(v):
            f[num] = val
        dtype, off, size = f.get(1, 0), f.get(4, 0), f.get(5, 0)
        raw = bytes(data[off:off + size])
        shape = [dict((n, x) for n, _, x in _parse(dim)).get(1, 0) for n2, _, dim in _parse(f.get(2, b'')) if n2 == 2]
        key = k.decode()
        if dtype == DT_STRING:
            n = int(np.prod(shape)) if shape else 1
            pos, lens = 0, []
            for _ in range(n):
                ln, pos = _read_varint(raw, pos)
                lens.append(ln)
            crc = _string_lengths_crc(lens)
            stored = raw[pos:pos + 4]
            if verify and (len(stored) != 4 or struct.unpack('<I', stored)[0] != _mask(crc)):
                raise ValueError(f"{key}: string-length checksum mismatch")
            crc = _crc32c(stored, crc)
            pos += 4
            vals = []
            for ln in lens:
                vals.append(raw[pos:pos + ln])
                pos += ln
                crc = _crc32c(vals[-1], crc)
            if verify and (pos != len(raw) or _unmask(f.get(6, 0)) != crc):
                raise ValueError(f"{key}: tensor checksum mismatch")
            if key == OBJECT_GRAPH_KEY:
                graph = vals[0]
            continue
        if verify and _unmask(f.get(6, 0)) != _crc32c(raw):
            raise ValueError(f"{key}: tensor checksum mismatch")
        arrays[key] = np.frombuffer(raw, dtype=DT_INV[dtype]).reshape(shape)
    return arrays, graph
