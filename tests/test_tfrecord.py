"""CPU: the reference's on-disk data format without TensorFlow: TFRecord framing, the
tf.train.Example encoding of dataset_tool.py:80-83, the record / label file set of TFRecordExporter, and the data set
object on top of it.  The protobuf encoding is cross-checked against Google's protobuf runtime (schema of
tensorflow/core/example/{example,feature}.proto declared on the fly), the checksum against the CRC-32C vectors of RFC 3720."""
import os

import numpy as np
import pytest


def test_crc32c_known_answers_and_masking():
    from inclusivegan_amd.training import tfrecord as T
    assert T.crc32c(b'123456789') == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    c = T.crc32c(b'abc')
    assert T.masked_crc32c(b'abc') == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def _example_classes():
    pb = pytest.importorskip('google.protobuf')
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name='tf_example_test.proto', package='tensorflow', syntax='proto3')
    def msg(name):
        m = fd.message_type.add(); m.name = name; return m
    L = descriptor_pb2.FieldDescriptorProto
    m = msg('BytesList'); f = m.field.add(name='value', number=1, type=L.TYPE_BYTES, label=L.LABEL_REPEATED)
    m = msg('FloatList'); f = m.field.add(name='value', number=1, type=L.TYPE_FLOAT, label=L.LABEL_REPEATED)
    m = msg('Int64List'); f = m.field.add(name='value', number=1, type=L.TYPE_INT64, label=L.LABEL_REPEATED)
    m = msg('Feature')
    m.oneof_decl.add(name='kind')
    m.field.add(name='bytes_list', number=1, type=L.TYPE_MESSAGE, type_name='.tensorflow.BytesList', label=L.LABEL_OPTIONAL, oneof_index=0)
    m.field.add(name='float_list', number=2, type=L.TYPE_MESSAGE, type_name='.tensorflow.FloatList', label=L.LABEL_OPTIONAL, oneof_index=0)
    m.field.add(name='int64_list', number=3, type=L.TYPE_MESSAGE, type_name='.tensorflow.Int64List', label=L.LABEL_OPTIONAL, oneof_index=0)
    m = msg('Features')
    e = m.nested_type.add(name='FeatureEntry'); e.options.map_entry = True
    e.field.add(name='key', number=1, type=L.TYPE_STRING, label=L.LABEL_OPTIONAL)
    e.field.add(name='value', number=2, type=L.TYPE_MESSAGE, type_name='.tensorflow.Feature', label=L.LABEL_OPTIONAL)
    m.field.add(name='feature', number=1, type=L.TYPE_MESSAGE, type_name='.tensorflow.Features.FeatureEntry', label=L.LABEL_REPEATED)
    m = msg('Example'); m.field.add(name='features', number=1, type=L.TYPE_MESSAGE, type_name='.tensorflow.Features', label=L.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, 'GetMessageClass', None)
    if get is None:
        return message_factory.MessageFactory(pool).GetPrototype(pool.FindMessageTypeByName('tensorflow.Example'))
    return get(pool.FindMessageTypeByName('tensorflow.Example'))


def test_example_encoding_against_protobuf_runtime():
    from inclusivegan_amd.training import tfrecord as T
    Example = _example_classes()
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(3, 8, 8)).astype(np.uint8)
    # what dataset_tool.py:80-83 builds, through the real protobuf runtime
    ex = Example()
    ex.features.feature['shape'].int64_list.value.extend(img.shape)
    ex.features.feature['data'].bytes_list.value.append(img.tobytes())
    theirs = ex.SerializeToString(deterministic=True)
    ours = T.serialize_example(img.shape, img.tobytes())
    assert ours == theirs                                               # byte-identical with deterministic (sorted-key) map order
    assert np.array_equal(T.parse_example(theirs), img)                 # and our parser reads the runtime's bytes
    back = Example(); back.ParseFromString(ours)                        # ... and the runtime reads ours
    assert list(back.features.feature['shape'].int64_list.value) == [3, 8, 8] and back.features.feature['data'].bytes_list.value[0] == img.tobytes()
    # non-deterministic map order (shape first) parses as well
    ex2 = Example()
    ex2.features.feature['data'].bytes_list.value.append(img.tobytes())
    ex2.features.feature['shape'].int64_list.value.extend(img.shape)
    assert np.array_equal(T.parse_example(ex2.SerializeToString()), img)


def test_exporter_files_and_dataset_round_trip(tmp_path):
    from inclusivegan_amd.training import tfrecord as T
    from inclusivegan_amd.training import dataset
    rng = np.random.RandomState(1)
    imgs = rng.randint(0, 256, size=(10, 3, 16, 16)).astype(np.uint8)
    labels = (rng.rand(10, 5) < 0.4).astype(np.float32)
    d = str(tmp_path / 'toy16')
    with T.TFRecordExporter(d, 10, print_progress=False) as e:
        order = np.arange(10); np.random.RandomState(123).shuffle(order)
        assert e.choose_shuffled_order().tolist() == order.tolist()                # dataset_tool.py:59-62
        for im in imgs:
            e.add_image(im)
        e.add_labels(labels)
    assert sorted(os.listdir(d)) == ['toy16-r02.tfrecords', 'toy16-r03.tfrecords', 'toy16-r04.tfrecords', 'toy16-rxx.labels']   # dataset_tool.py:71-72,90
    # level-of-detail files hold the 2x2 box-filtered, re-quantised images (:75-78)
    lod1 = [T.parse_example(r) for r in T.read_records(os.path.join(d, 'toy16-r03.tfrecords'), verify=True)]
    f = imgs[0].astype(np.float32)
    want = np.rint((f[:, 0::2, 0::2] + f[:, 0::2, 1::2] + f[:, 1::2, 0::2] + f[:, 1::2, 1::2]) * 0.25).clip(0, 255).astype(np.uint8)
    assert len(lod1) == 10 and np.array_equal(lod1[0], want)
    ds = dataset.load_dataset(data_dir=str(tmp_path), tfrecord_dir='toy16', max_label_size='full', shuffle_mb=0)
    assert isinstance(ds, dataset.TFRecordDataset) and ds.shape == [3, 16, 16] and ds.label_size == 5 and ds.data_size == 10 and ds.resolution_log2 == 4
    x, l = ds.get_minibatch_np(4)
    assert x.dtype == np.uint8 and np.array_equal(x, imgs[:4]) and np.array_equal(l, labels[:4])
    x, l = ds.get_minibatch_np(4)
    assert np.array_equal(x, imgs[4:8])
    x, l = ds.get_minibatch_np(4)                                        # wraps around (repeat)
    assert np.array_equal(x, np.concatenate([imgs[8:], imgs[:2]]))
    ds.configure(6)                                                       # new minibatch size: the iterator restarts (dataset.py:139-145)
    assert np.array_equal(ds.get_minibatch_np(6)[0], imgs[:6])
    ds3 = dataset.load_dataset(data_dir=str(tmp_path), tfrecord_dir='toy16', max_label_size=2, max_images=7)
    assert ds3.label_size == 2 and ds3.data_size == 7 and np.array_equal(ds3._labels, labels[:7, :2])
    assert dataset.load_dataset(data_dir=str(tmp_path), tfrecord_dir='toy16').label_size == 0           # max_label_size defaults to 0 (:23)
    np.random.seed(3)
    want_idx = np.random.randint(10, size=[6])
    np.random.seed(3)
    assert np.array_equal(ds.get_random_labels_np(6), labels[want_idx])                                  # global NumPy stream (:163-166)
    with pytest.raises(FileNotFoundError):
        dataset.load_dataset(data_dir=str(tmp_path), tfrecord_dir='nope')
    # a corrupted payload is caught when verification is on
    p = os.path.join(d, 'toy16-r04.tfrecords')
    raw = bytearray(open(p, 'rb').read()); raw[40] ^= 0xFF; open(p, 'wb').write(raw)
    with pytest.raises(IOError):
        list(T.read_records(p, verify=True))
