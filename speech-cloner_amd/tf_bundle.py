"""Pure-Python reader/writer for TensorFlow "bundle" (V2) checkpoints.

The reference saves and restores its models with ``tf.train.Saver``
(/root/reference/encoder.py:207-253, decoder.py:276-324), i.e. files
``<prefix>.index`` (a LevelDB-style SSTable of ``BundleEntryProto``s) and
``<prefix>.data-00000-of-00001`` (raw little-endian tensors), plus the text file
``checkpoint`` that ``tf.train.latest_checkpoint`` reads.  TensorFlow is not available on
the target machine, so this module implements the on-disk format directly (layout described
in SURVEY.md section 5).  Nothing in a checkpoint is executed: the files are parsed as bytes.

Public API
----------
read_bundle(prefix, verify_crc=True) -> dict name -> np.ndarray
list_bundle(prefix)                  -> dict name -> BundleEntry (dtype, shape, offset, size, crc)
write_bundle(prefix, tensors)        -> writes .index / .data-00000-of-00001
latest_checkpoint(model_dir)         -> prefix or None    (tf.train.latest_checkpoint)
update_checkpoint_state(model_dir, prefix_basename)
"""
import os
import struct
from collections import namedtuple

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8

# TensorFlow DataType enum values used by the reference's checkpoints
_DT_TO_NP = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 9: np.dtype('<i8'),
             10: np.dtype('bool')}
_NP_TO_DT = {np.dtype('float32'): 1, np.dtype('float64'): 2, np.dtype('int32'): 3,
             np.dtype('int64'): 9, np.dtype('bool'): 10}

BundleEntry = namedtuple('BundleEntry', 'dtype shape shard_id offset size crc32c')


# --------------------------------------------------------------------------- crc32c
def _make_crc_table():
    poly = 0x82F63B78
    tab = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (poly if (c & 1) else 0)
        tab[i] = c
    return tab


_CRC_TAB = _make_crc_table()
_CRC_TAB_L = [int(v) for v in _CRC_TAB]


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli).  Table-driven; byte loop in Python (~3 MB/s), used only on
    checkpoint load/save."""
    tab = _CRC_TAB_L
    c = crc ^ 0xFFFFFFFF
    for b in bytes(data):
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c):
    """LevelDB/TF masked CRC: rotate right by 15 and add a constant."""
    return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(m):
    rot = (m - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# --------------------------------------------------------------------------- varint / proto
def _get_varint(buf, pos):
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not (b & 0x80):
            return result, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf):
    """Minimal protobuf wire parser -> list of (field_no, wire_type, value)."""
    pos = 0
    n = len(buf)
    out = []
    while pos < n:
        key, pos = _get_varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        out.append((fno, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_entry(buf):
    dtype, shape, shard, offset, size, crc = 0, [], 0, 0, 0, 0
    for fno, wt, v in _parse_proto(buf):
        if fno == 1:
            dtype = v
        elif fno == 2:                      # TensorShapeProto
            for f2, _, v2 in _parse_proto(v):
                if f2 == 2:                 # Dim
                    dsize = 0
                    for f3, _, v3 in _parse_proto(v2):
                        if f3 == 1:
                            dsize = _signed64(v3)
                    shape.append(dsize)
        elif fno == 3:
            shard = v
        elif fno == 4:
            offset = _signed64(v)
        elif fno == 5:
            size = _signed64(v)
        elif fno == 6:
            crc = v
        elif fno == 7:
            raise ValueError('sliced (partitioned) variables are not supported')
    return BundleEntry(dtype, tuple(shape), shard, offset, size, crc)


def _encode_entry(e):
    out = bytearray()
    out += b'\x08' + _put_varint(e.dtype)                       # field 1 varint
    dims = bytearray()
    for d in e.shape:
        dim = b'\x08' + _put_varint(d)
        dims += b'\x12' + _put_varint(len(dim)) + dim           # TensorShapeProto.dim (field 2)
    out += b'\x12' + _put_varint(len(dims)) + bytes(dims)       # field 2 (always written)
    if e.shard_id:
        out += b'\x18' + _put_varint(e.shard_id)
    if e.offset:
        out += b'\x20' + _put_varint(e.offset)
    out += b'\x28' + _put_varint(e.size)
    out += b'\x35' + struct.pack('<I', e.crc32c)                # field 6 fixed32
    return bytes(out)


# --------------------------------------------------------------------------- sstable read
def _read_block(data, offset, size, verify=True):
    contents = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise ValueError('compressed SSTable blocks are not supported (type %d)' % ctype)
    if verify:
        stored = struct.unpack_from('<I', data, offset + size + 1)[0]
        actual = mask_crc(crc32c(data[offset:offset + size + 1]))
        if stored != actual:
            raise ValueError('SSTable block checksum mismatch at offset %d' % offset)
    return contents


def _iter_block(block):
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * n_restarts
    pos = 0
    key = b''
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        value = bytes(block[pos:pos + vlen])
        pos += vlen
        yield key, value


def _read_index(path, verify=True):
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < 48:
        raise ValueError('%s: too short for an SSTable' % path)
    footer = data[-48:]
    if struct.unpack_from('<Q', footer, 40)[0] != _MAGIC:
        raise ValueError('%s: bad SSTable magic' % path)
    pos = 0
    _mi_off, pos = _get_varint(footer, pos)
    _mi_sz, pos = _get_varint(footer, pos)
    ix_off, pos = _get_varint(footer, pos)
    ix_sz, pos = _get_varint(footer, pos)
    index_block = _read_block(data, ix_off, ix_sz, verify)
    entries = {}
    header = None
    for _, handle in _iter_block(index_block):
        boff, p = _get_varint(handle, 0)
        bsz, p = _get_varint(handle, p)
        for key, value in _iter_block(_read_block(data, boff, bsz, verify)):
            if key == b'':
                header = value
            else:
                entries[key.decode('utf-8')] = _parse_entry(value)
    return header, entries


def list_bundle(prefix, verify=True):
    """name -> BundleEntry for every tensor in ``<prefix>.index``."""
    header, entries = _read_index(prefix + '.index', verify)
    if header is None:
        raise ValueError('%s.index: missing bundle header' % prefix)
    num_shards = 1
    for fno, _, v in _parse_proto(header):
        if fno == 1:
            num_shards = v
        elif fno == 2 and v != 0:
            raise ValueError('big-endian bundles are not supported')
    if num_shards != 1:
        raise ValueError('multi-shard bundles are not supported (num_shards=%d)' % num_shards)
    return entries


def read_bundle(prefix, verify_crc=True, names=None):
    """Load tensors of the checkpoint ``prefix`` into numpy arrays.

    ``names``: optional iterable restricting what is loaded.  Every loaded tensor's
    masked CRC32C is checked against the index when ``verify_crc``."""
    entries = list_bundle(prefix, verify_crc)
    with open(prefix + '.data-00000-of-00001', 'rb') as f:
        data = f.read()
    out = {}
    for name, e in entries.items():
        if names is not None and name not in names:
            continue
        if e.dtype not in _DT_TO_NP:
            raise ValueError('%s: unsupported dtype enum %d' % (name, e.dtype))
        raw = data[e.offset:e.offset + e.size]
        if len(raw) != e.size:
            raise ValueError('%s: data file truncated' % name)
        if verify_crc and mask_crc(crc32c(raw)) != e.crc32c:
            raise ValueError('%s: tensor checksum mismatch' % name)
        out[name] = np.frombuffer(raw, dtype=_DT_TO_NP[e.dtype]).reshape(e.shape).copy()
    return out


# --------------------------------------------------------------------------- sstable write
class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf = bytearray()
        self.restarts = [0]
        self.counter = 0
        self.last_key = b''
        self.interval = restart_interval

    def add(self, key, value):
        shared = 0
        if self.counter < self.interval:
            m = min(len(key), len(self.last_key))
            while shared < m and key[shared] == self.last_key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last_key = key
        self.counter += 1

    def finish(self):
        out = bytes(self.buf)
        for r in self.restarts:
            out += struct.pack('<I', r)
        out += struct.pack('<I', len(self.restarts))
        return out

    def empty(self):
        return len(self.buf) == 0

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4


def _emit_block(out, contents):
    offset = len(out)
    out += contents + b'\x00'
    out += struct.pack('<I', mask_crc(crc32c(contents + b'\x00')))
    return _put_varint(offset) + _put_varint(len(contents))


def _shortest_separator(a, b):
    """A key k with a <= k < b, as LevelDB's BytewiseComparator does."""
    m = min(len(a), len(b))
    i = 0
    while i < m and a[i] == b[i]:
        i += 1
    if i < m and a[i] < 0xFF and a[i] + 1 < b[i]:
        return a[:i] + bytes([a[i] + 1])
    return a


def write_bundle(prefix, tensors, block_size=4096):
    """Write ``tensors`` (dict name -> array; float32/int32/...) as a single-shard V2 bundle."""
    names = sorted(tensors.keys(), key=lambda s: s.encode('utf-8'))
    data = bytearray()
    entries = []
    for name in names:
        a = np.asarray(tensors[name], order='C')          # (ascontiguousarray would turn 0-d into 1-d)
        if a.dtype not in _NP_TO_DT:
            raise ValueError('%s: unsupported dtype %s' % (name, a.dtype))
        raw = a.astype(a.dtype.newbyteorder('<')).tobytes()
        entries.append((name.encode('utf-8'),
                        BundleEntry(_NP_TO_DT[a.dtype], tuple(int(d) for d in a.shape), 0,
                                    len(data), len(raw), mask_crc(crc32c(raw)))))
        data += raw
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(bytes(data))

    # BundleHeaderProto{num_shards=1, endianness=LITTLE(0), version{producer=1}}
    header = b'\x08\x01' + b'\x1a\x02\x08\x01'
    kv = [(b'', header)] + [(k, _encode_entry(e)) for k, e in entries]

    out = bytearray()
    index = _BlockBuilder(restart_interval=1)
    blk = _BlockBuilder()
    pending = None                              # (last_key, handle) awaiting its separator
    for key, value in kv:
        if pending is not None:
            index.add(_shortest_separator(pending[0], key), pending[1])
            pending = None
        blk.add(key, value)
        if blk.size() >= block_size:
            pending = (blk.last_key, _emit_block(out, blk.finish()))
            blk = _BlockBuilder()
    if not blk.empty():
        if pending is not None:
            index.add(pending[0], pending[1])
        pending = (blk.last_key, _emit_block(out, blk.finish()))
    if pending is not None:
        index.add(pending[0], pending[1])
    meta_handle = _emit_block(out, _BlockBuilder().finish())
    index_handle = _emit_block(out, index.finish())
    footer = meta_handle + index_handle
    footer += b'\x00' * (40 - len(footer))
    footer += struct.pack('<Q', _MAGIC)
    out += footer
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out))


# --------------------------------------------------------------------------- checkpoint state
def latest_checkpoint(model_dir):
    """tf.train.latest_checkpoint: parse ``<model_dir>/checkpoint`` (CheckpointState text
    proto) and return the prefix it names, or None."""
    path = os.path.join(model_dir, 'checkpoint')
    if not os.path.exists(path):
        return None
    with open(path, 'r') as f:
        for line in f:
            line = line.strip()
            if line.startswith('model_checkpoint_path:'):
                name = line.split(':', 1)[1].strip().strip('"')
                prefix = name if os.path.isabs(name) else os.path.join(model_dir, name)
                if os.path.exists(prefix + '.index'):
                    return prefix
                return None
    return None


def update_checkpoint_state(model_dir, basename):
    """Append ``basename`` to ``<model_dir>/checkpoint`` the way tf.train.Saver does
    (max_to_keep=9999 in the reference, so nothing is ever pruned)."""
    path = os.path.join(model_dir, 'checkpoint')
    allp = []
    if os.path.exists(path):
        with open(path, 'r') as f:
            for line in f:
                line = line.strip()
                if line.startswith('all_model_checkpoint_paths:'):
                    allp.append(line.split(':', 1)[1].strip().strip('"'))
    if basename in allp:
        allp.remove(basename)
    allp.append(basename)
    os.makedirs(model_dir, exist_ok=True)
    with open(path, 'w') as f:
        f.write('model_checkpoint_path: "%s"\n' % basename)
        for p in allp:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)
