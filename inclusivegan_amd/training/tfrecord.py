"""The on-disk data format of the reference, without TensorFlow.

`dataset_tool.py:30-96` (TFRecordExporter) writes one `<name>-rNN.tfrecords` file per level of detail -- the full
resolution 2^NN and every 2x2 box-filtered reduction down to 4x4 -- plus `<name>-rxx.labels` (np.save of float32 labels).
Each record is a serialized `tf.train.Example` with two features: 'shape' (int64 [3] = C, H, W) and 'data' (the uint8 CHW
bytes).  `training/dataset.py:171-176` parses exactly those two fields.

TFRecord framing (tensorflow/core/lib/io/record_writer.cc -- third-party format, restated): per record
    uint64 length | uint32 masked_crc32c(length bytes) | payload | uint32 masked_crc32c(payload)
    masked(c) = ((c >> 15) | (c << 17)) + 0xa282ead8  (mod 2^32), crc32c = CRC-32/Castagnoli.
Protobuf wire format of the Example (proto3): Example{1: Features{repeated 1: MapEntry{1: key string, 2: Feature}}},
Feature{1: BytesList{repeated 1: bytes} | 3: Int64List{repeated 1: varint, packed}}.

The reader walks records with the lengths alone (payload checksums are verified on request); images are returned as NumPy
uint8 views of the file bytes (no copy until stacked).
"""
import os
import struct

import numpy as np

_MASK_DELTA = 0xa282ead8


def _make_crc_table():
    poly = 0x82F63B78
    tab = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        tab[i] = c
    return tab


_CRC_TABLE = _make_crc_table()
_CRC_TABLE_LIST = [int(v) for v in _CRC_TABLE]


def crc32c(data):
    """CRC-32C of a bytes-like object (table driven; the record payloads are small)."""
    c = 0xFFFFFFFF
    tab = _CRC_TABLE_LIST
    for b in bytes(data):
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


# ---- protobuf wire helpers -------------------------------------------------------------------------------------------

def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = 0
    val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _field(num, payload):
    """length-delimited field"""
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def _fields(buf):
    """Iterate (field number, wire type, value) over a message; value = int for varints, memoryview for length-delimited."""
    pos = 0
    buf = memoryview(buf)
    n = len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 1:
            val = buf[pos:pos + 8]; pos += 8
        elif wt == 5:
            val = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield num, wt, val


def serialize_example(shape, data):
    """tf.train.Example{'shape': int64_list, 'data': bytes_list} as dataset_tool.py:80-83 builds it (map entries in key order,
    the order the C++ serializer emits for a map<string, Feature>)."""
    def entry(key, feature):
        return _field(1, _field(1, key.encode()) + _field(2, feature))
    int64_list = _field(1, b''.join(_varint(int(v)) for v in shape))           # packed repeated int64
    shape_feature = _field(3, int64_list)
    data_feature = _field(1, _field(1, bytes(data)))
    features = entry('data', data_feature) + entry('shape', shape_feature)
    return _field(1, features)


def parse_example(record):
    """-> uint8 array of the recorded shape (training/dataset.py:171-176 parse_tfrecord_np)."""
    shape = None
    data = None
    for num, wt, features in _fields(record):
        if num != 1:
            continue
        for fnum, _, entry in _fields(features):
            if fnum != 1:
                continue
            key = None
            feature = None
            for enum_, _, v in _fields(entry):
                if enum_ == 1:
                    key = bytes(v).decode()
                elif enum_ == 2:
                    feature = v
            if key == 'shape':
                for knum, _, lst in _fields(feature):
                    if knum == 3:
                        vals = []
                        for vnum, vwt, v in _fields(lst):
                            if vnum != 1:
                                continue
                            if vwt == 0:
                                vals.append(v)
                            else:                       # packed
                                p = 0
                                while p < len(v):
                                    x, p = _read_varint(v, p)
                                    vals.append(x)
                        shape = vals
            elif key == 'data':
                for knum, _, lst in _fields(feature):
                    if knum == 1:
                        for vnum, _, v in _fields(lst):
                            if vnum == 1:
                                data = v
    if shape is None or data is None:
        raise ValueError('record is not an Example with "shape" and "data" features')
    return np.frombuffer(data, dtype=np.uint8).reshape(shape)


# ---- record files ------------------------------------------------------------------------------------------------------

class TFRecordWriter:
    def __init__(self, path):
        self._f = open(path, 'wb')

    def write(self, payload):
        header = struct.pack('<Q', len(payload))
        self._f.write(header + struct.pack('<I', masked_crc32c(header)) + payload + struct.pack('<I', masked_crc32c(payload)))

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None


def read_records(path, verify=False, limit=None):
    """Yield the payload (memoryview) of every record of a .tfrecords file."""
    buf = memoryview(np.fromfile(path, dtype=np.uint8)) if os.path.getsize(path) else memoryview(b'')
    pos = 0
    n = len(buf)
    count = 0
    while pos < n and (limit is None or count < limit):
        (length,) = struct.unpack_from('<Q', buf, pos)
        if verify:
            (c,) = struct.unpack_from('<I', buf, pos + 8)
            if c != masked_crc32c(buf[pos:pos + 8]):
                raise IOError('%s: corrupt record length at byte %d' % (path, pos))
        payload = buf[pos + 12:pos + 12 + length]
        if len(payload) != length:
            raise IOError('%s: truncated record at byte %d' % (path, pos))
        if verify:
            (c,) = struct.unpack_from('<I', buf, pos + 12 + length)
            if c != masked_crc32c(payload):
                raise IOError('%s: corrupt record payload at byte %d' % (path, pos))
        yield payload
        pos += 12 + length + 4
        count += 1


class TFRecordExporter:
    """dataset_tool.py:30-96: one record file per level of detail, labels as '<prefix>-rxx.labels'."""

    def __init__(self, tfrecord_dir, expected_images, print_progress=True, progress_interval=10):
        self.tfrecord_dir = tfrecord_dir
        self.tfr_prefix = os.path.join(self.tfrecord_dir, os.path.basename(self.tfrecord_dir))
        self.expected_images = expected_images
        self.cur_images = 0
        self.shape = None
        self.resolution_log2 = None
        self.tfr_writers = []
        self.print_progress = print_progress
        self.progress_interval = progress_interval
        if self.print_progress:
            print('Creating dataset "%s"' % tfrecord_dir)
        os.makedirs(self.tfrecord_dir, exist_ok=True)

    def close(self):
        for w in self.tfr_writers:
            w.close()
        self.tfr_writers = []
        if self.print_progress:
            print('Added %d images.' % self.cur_images)

    def choose_shuffled_order(self):    # images and labels must be added in shuffled order (:59-62)
        order = np.arange(self.expected_images)
        np.random.RandomState(123).shuffle(order)
        return order

    def add_image(self, img):
        if self.shape is None:
            self.shape = img.shape
            self.resolution_log2 = int(np.log2(self.shape[1]))
            assert self.shape[0] in [1, 3]
            assert self.shape[1] == self.shape[2]
            assert self.shape[1] == 2 ** self.resolution_log2
            for lod in range(self.resolution_log2 - 1):
                self.tfr_writers.append(TFRecordWriter(self.tfr_prefix + '-r%02d.tfrecords' % (self.resolution_log2 - lod)))
        assert img.shape == self.shape
        for lod, w in enumerate(self.tfr_writers):
            if lod:
                img = img.astype(np.float32)
                img = (img[:, 0::2, 0::2] + img[:, 0::2, 1::2] + img[:, 1::2, 0::2] + img[:, 1::2, 1::2]) * 0.25
            quant = np.rint(img).clip(0, 255).astype(np.uint8)
            w.write(serialize_example(quant.shape, quant.tobytes()))
        self.cur_images += 1

    def add_labels(self, labels):
        assert labels.shape[0] == self.cur_images
        with open(self.tfr_prefix + '-rxx.labels', 'wb') as f:
            np.save(f, labels.astype(np.float32))

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()
