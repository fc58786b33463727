"""Row f3: TensorFlow V2 checkpoint (tensor bundle) reader.  No TF checkpoint exists offline, so the
reader is exercised on bundles produced by the module's own writer, which lays the files out as
TF's BundleWriter does (SSTable index with prefix-compressed keys in several data blocks + raw data
shard + `checkpoint' state file)."""
import os
import struct

import numpy as np
import pytest

from davo_amd import tf_checkpoint as T
from davo_amd import synth, parse_version, FLAGSHIP_VERSION
from davo_amd.version import weight_shapes


def test_crc32c_known_answers():
    # RFC 3720 test vectors
    assert T.crc32c(b"") == 0
    assert T.crc32c(b"123456789") == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43


def test_snappy_decoder_literals_and_copies():
    # "abcabcabcabc": literal "abc" then a copy of length 9 at offset 3 (2-byte-offset form)
    raw = bytes([12]) + bytes([(3 - 1) << 2]) + b"abc" + bytes([((9 - 1) << 2) | 2, 3, 0])
    assert T._snappy_decompress(raw) == b"abcabcabcabc"


def test_roundtrip_flagship_weights(tmp_path):
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    weights_plus = dict(weights)
    weights_plus["global_step"] = np.array(1600000, np.int64)           # davo.py:865-867 saves it too
    prefix = str(tmp_path / "model-1600000")
    T.write_checkpoint(prefix, weights_plus)
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    listed = {n: (s, d) for n, s, d in T.list_variables(prefix)}
    assert set(listed) == set(weights_plus)
    for n, shp in weight_shapes(cfg).items():
        assert listed[n] == (shp, np.float32)
    assert listed["global_step"] == ((), np.int64)
    got = T.read_checkpoint(prefix, verify_crc=True)
    for n, a in weights.items():
        assert got[n].dtype == np.float32 and np.array_equal(got[n], a)
    assert int(got["global_step"]) == 1600000
    # the three ways run_inference.sh / test_kitti_pose.py name a checkpoint
    for p in (prefix, prefix + ".index", str(tmp_path)):
        only = T.read_checkpoint(p, names=["pose_exp_net/cnv1/weights"])
        assert list(only) == ["pose_exp_net/cnv1/weights"]
    assert set(T.load_weights(prefix)) == set(weights_plus)


def test_npz_side_format(tmp_path):
    w = {"pose_exp_net/cnv1/biases": np.arange(16, dtype=np.float32)}
    np.savez(str(tmp_path / "w.npz"), **w)
    got = T.load_weights(str(tmp_path / "w.npz"))
    assert np.array_equal(got["pose_exp_net/cnv1/biases"], w["pose_exp_net/cnv1/biases"])


def test_errors(tmp_path):
    prefix = str(tmp_path / "model-1")
    T.write_checkpoint(prefix, {"a/b": np.ones((2, 3), np.float32), "a/c": np.zeros(4, np.float32)})
    with pytest.raises(KeyError, match="no variable `nope'"):
        T.read_checkpoint(prefix, names=["nope"])
    raw = bytearray(open(prefix + ".index", "rb").read())
    bad = bytearray(raw); bad[-1] ^= 0xFF
    open(prefix + ".index", "wb").write(bytes(bad))
    with pytest.raises(ValueError, match="bad magic"):
        T.read_checkpoint(prefix)
    bad = bytearray(raw); bad[3] ^= 0x01                                  # flip a bit inside the first block
    open(prefix + ".index", "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        T.read_checkpoint(prefix)
    open(prefix + ".index", "wb").write(bytes(raw))
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[0] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError, match="crc32c mismatch"):
        T.read_checkpoint(prefix, verify_crc=True)
    os.makedirs(str(tmp_path / "sub"))
    with pytest.raises(FileNotFoundError):
        T.resolve_checkpoint(str(tmp_path / "sub"))


def test_footer_layout(tmp_path):
    prefix = str(tmp_path / "m")
    T.write_checkpoint(prefix, {"x": np.zeros(1, np.float32)})
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57 and len(raw) >= 48


def test_entries_the_path_never_needs_are_skipped_not_fatal(tmp_path, monkeypatch):
    """A training checkpoint also holds optimizer slots, global_step and (newer TF) string entries: reading
    everything leaves out dtypes the reader does not know; asking for such an entry by name still raises."""
    prefix = str(tmp_path / "model-7")
    T.write_checkpoint(prefix, {"global_step": np.array(7, np.int64), "pose_exp_net/cnv1/biases": np.arange(16, dtype=np.float32)})
    monkeypatch.setitem(T._DT, 9, None)                                   # pretend int64 is an unknown dtype
    got = T.read_checkpoint(prefix)
    assert list(got) == ["pose_exp_net/cnv1/biases"]
    with pytest.raises(ValueError, match="unsupported dtype"):
        T.read_checkpoint(prefix, names=["global_step"])


def test_sharded_bundle_with_training_slots(tmp_path):
    """What a training run leaves behind (davo.py:865-867 saves trainable variables + global_step; Adam keeps two slot
    variables per weight; a sharded Saver writes data-0000N-of-0000M): the reader follows each entry's shard_id, and
    loading the weights of the path ignores everything the variant does not name."""
    cfg = parse_version(FLAGSHIP_VERSION)
    weights = synth.make_weights(cfg)
    everything = dict(weights)
    everything["global_step"] = np.array(123456, np.int64)
    everything["beta1_power"] = np.array(0.5, np.float32)
    for k, v in weights.items():
        everything[k + "/Adam"] = np.full_like(v, 0.25)
        everything[k + "/Adam_1"] = np.full_like(v, 4.0)
    prefix = str(tmp_path / "model-123456")
    T.write_checkpoint(prefix, everything, num_shards=3)
    for sh in range(3):
        assert os.path.getsize("%s.data-%05d-of-00003" % (prefix, sh)) > 0
    assert not os.path.exists(prefix + ".data-00000-of-00001")
    got = T.read_checkpoint(str(tmp_path), verify_crc=True)              # via the `checkpoint' state file
    assert set(got) == set(everything)
    for n, a in weights.items():
        assert np.array_equal(got[n], a), n
        assert float(got[n + "/Adam"].ravel()[0]) == 0.25 and float(got[n + "/Adam_1"].ravel()[0]) == 4.0
    # the engine-side filter: exactly the variant's 26 tensors are taken, by name
    want = weight_shapes(cfg)
    assert set(want) <= set(got) and len(want) == 26
    # a data shard that went missing is an error, not a silent partial restore
    os.remove(prefix + ".data-00002-of-00003")
    with pytest.raises(FileNotFoundError):
        T.read_checkpoint(prefix)
