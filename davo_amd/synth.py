"""Synthetic inputs and weights for the pose path.

No KITTI dump, FlowNet2/DeepLab output or checkpoint exists offline, so tests and
``bench.py`` run on synthetic tensors of the reference's shapes and statistics:

* ``img``  u8  ``[B,H,3W,3]``   uniform{0..255}; strip order src0|tgt|src1
  (reference ``data_loader.py:537-557``, ``data/preprocess.py:61-66``)
* ``flow`` f32 ``[B,4,H,W,2]``  ~ (0.32140523, 15.384229) — the dataset statistics the
  reference hard-codes (``davo.py:1090``)
* ``seg``  f32 ``[B,3,H,W,1]``  Cityscapes train ids 0..18 in 8x8 blocks, 3 % of blocks 255
  (ignore id, ``utils/seg_utils/labels.py:64-70``)

The PRNG is splitmix64 and every float is produced by integer arithmetic plus IEEE
add/mul/sqrt only (the "normal" is an Irwin-Hall sum of four uniforms), so the same
seed gives bit-identical tensors on every host — the GPU box regenerates the inputs
that the committed golden outputs under tests/golden/ were computed from.
Seed 8964 is the reference's own (``train.py:34``).
"""
import math

import numpy as np

from .version import parse_version, weight_shapes, NUM_SEG_CLASSES

SEED = 8964
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GAMMA = 0x9E3779B97F4A7C15


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def splitmix64(seed, stream, n):
    """n 64-bit outputs of the splitmix64 sequence whose state starts at
    mix(seed, stream); counter-based, so it vectorises."""
    with np.errstate(over="ignore"):
        base = _mix(np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(stream) * np.uint64(_GAMMA))
        idx = np.arange(1, n + 1, dtype=np.uint64)
        return _mix(base + idx * np.uint64(_GAMMA))


def uniform01(seed, stream, n):
    """float64 in [0,1) with 53 random bits."""
    return (splitmix64(seed, stream, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normalish(seed, stream, n):
    """zero-mean unit-variance Irwin-Hall(4) variate; add/mul only."""
    u = uniform01(seed, stream, 4 * n).reshape(4, n)
    s = (u[0] + u[1]) + (u[2] + u[3])
    return (s - 2.0) * 1.7320508075688772          # sqrt(12/4)


def _stream(name):
    """stable 32-bit stream id from a tensor name (FNV-1a)."""
    h = 0x811C9DC5
    for ch in name.encode():
        h = ((h ^ ch) * 0x01000193) & 0xFFFFFFFF
    return h


def make_inputs(B, H=128, W=416, seed=SEED, first_window=0):
    """Synthetic batch of B triplets; window w of a sequence is reproducible on its own
    (``first_window`` offsets the per-window streams), which is what lets ranks generate
    just their shard."""
    img = np.empty((B, H, 3 * W, 3), np.uint8)
    flow = np.empty((B, 4, H, W, 2), np.float32)
    seg = np.empty((B, 3, H, W, 1), np.float32)
    hb, wb = (H + 7) // 8, (W + 7) // 8
    for b in range(B):
        w = first_window + b
        r = splitmix64(seed, _stream("img") + 7919 * w, (H * 3 * W * 3 + 7) // 8)
        img[b] = r.view(np.uint8)[: H * 3 * W * 3].reshape(H, 3 * W, 3)
        f = normalish(seed, _stream("flow") + 7919 * w, 4 * H * W * 2)
        flow[b] = (f * 15.384229 + 0.32140523).astype(np.float32).reshape(4, H, W, 2)
        r = splitmix64(seed, _stream("seg") + 7919 * w, 3 * hb * wb)
        ids = (r % np.uint64(NUM_SEG_CLASSES)).astype(np.float32)
        ign = ((r >> np.uint64(32)) % np.uint64(100)) < np.uint64(3)
        ids[ign] = 255.0
        blk = ids.reshape(3, hb, wb)
        seg[b, :, :, :, 0] = np.repeat(np.repeat(blk, 8, axis=1), 8, axis=2)[:, :H, :W]
    return img, flow, seg


def make_weights(version_or_cfg, seed=SEED):
    """dict TF-name -> float32 array (HWIO conv kernels, [in,out] dense kernels).

    conv ~ U(+-sqrt(6/fan_in)) (He-uniform keeps activations O(1) through eight ReLU
    layers, so the 6-DoF outputs are large enough for the 1e-4 absolute bar to bite),
    conv biases U(+-0.05), SE kernels variance-scaling as nets/attention_module.py:60,
    SE biases U(+-0.5), static seg weights stddev 0.05 as nets/posenn.py:387-388."""
    cfg = parse_version(version_or_cfg) if isinstance(version_or_cfg, str) else version_or_cfg
    out = {}
    for name, shape in weight_shapes(cfg).items():
        n = int(np.prod(shape))
        st = _stream(name)
        if name.endswith("/weights"):
            fan_in = shape[0] * shape[1] * shape[2]
            a = (uniform01(seed, st, n) * 2.0 - 1.0) * math.sqrt(6.0 / fan_in)
            if "/pred/" in name:
                a = a * 16.0        # lifts the 6-DoF outputs to O(0.1), the scale of real KITTI motion
        elif name.endswith("/biases"):
            a = (uniform01(seed, st, n) * 2.0 - 1.0) * 0.05
        elif name.endswith("/kernel"):
            a = normalish(seed, st, n) * math.sqrt(2.0 / shape[0])
        elif name.endswith("/bias"):
            a = (uniform01(seed, st, n) * 2.0 - 1.0) * 0.5
        elif name.endswith("seg_channel_weight/weight"):
            a = normalish(seed, st, n) * 0.05
        else:
            raise KeyError(name)
        out[name] = a.astype(np.float32).reshape(shape)
    return out
