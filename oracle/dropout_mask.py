"""numpy restatement of the counter-based dropout mask of libglnn_hip.so (csrc/glnn_common.h: drop_hash /
drop_threshold / drop_keep) and of the per-step seed schedule of glnn_amd.student.StudentEngine._seed.

TEST INFRASTRUCTURE ONLY.  torch's Philox dropout stream cannot be reproduced by a custom kernel, so parity
for dropout > 0 is stated as: GIVEN the keep-mask, the step equals the reference's step (SURVEY.md 8c "one
case with an explicit mask applied outside").  tests/golden/make_student_golden.py feeds masks from THIS
file into the reference's own MLP (its nn.Dropout swapped for a mask multiply) and the GPU test checks both
that glnn_dropout_mask_u8 equals this restatement bit for bit and that the fused step reproduces the
reference's numbers under those masks (reference models.py:52, train_and_eval.py:74-85)."""
import numpy as np

U32 = np.uint32


def _mix32(x):
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def drop_hash(seed, row, col):
    """lowbias32 of seed ^ row*0x9E3779B1 ^ (col*0x85EBCA77 + 0x632BE5AB), all mod 2^32."""
    row = np.asarray(row, dtype=np.uint64)
    col = np.asarray(col, dtype=np.uint64)
    h = (np.uint64(int(seed) & 0xFFFFFFFF) ^ ((row * 0x9E3779B1) & 0xFFFFFFFF)
         ^ ((col * 0x85EBCA77 + 0x632BE5AB) & 0xFFFFFFFF))
    return _mix32(h)


def drop_threshold(p):
    return int(np.float32(p) * np.float32(65536.0))


def keep_mask(rows, h, p, seed):
    """uint8 [rows, h]: 1 = kept.  One hash per PAIR of adjacent columns, 16 bits of it per element."""
    r = np.arange(rows, dtype=np.uint64)[:, None]
    c = np.arange(h, dtype=np.uint64)[None, :]
    hv = drop_hash(seed, r, c >> np.uint64(1))
    bits = np.where((c & np.uint64(1)) == 1, hv >> np.uint64(16), hv & np.uint64(0xFFFF))
    return (bits >= np.uint64(drop_threshold(p))).astype(np.uint8)


def engine_seed(base_seed, step, layer):
    """StudentEngine._seed: mix32(base ^ mix32(step*131 + layer + 1)), step = 1-based optimiser step."""
    return int(_mix32((int(base_seed) & 0xFFFFFFFF) ^ int(_mix32(step * 131 + layer + 1))))
