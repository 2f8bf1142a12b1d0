"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of the element-wise dropout mask generator of
mr-mt3_amd/csrc/common.h (`make_drop`, `drop_mix`, `drop_mask4`).

The reference drops with torch's generator (`nn.Dropout(config.dropout_rate)` inside HF T5Block, models/t5.py:487-490),
whose stream cannot be reproduced on another device; what has to hold is the DISTRIBUTION (keep probability 1 - p,
scale 1 / (1 - p), independence across elements, sites and steps) and that backward regenerates forward's mask.  This
file pins the generator the kernels use so both can be tested: bit-for-bit against the GPU, statistically on the CPU.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _mul24(a, c):
    return ((a & np.uint64(0xFFFFFF)) * np.uint64(c & 0xFFFFFF)) & M32


def drop_mix(x):
    """xor-shift / 24-bit multiply / xor-shift / 24-bit multiply / xor-shift on uint32 values held in uint64."""
    x = x.astype(np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = _mul24(x, 0x7FEB35)
    x ^= x >> np.uint64(15)
    x = _mul24(x, 0x6CA68B)
    x ^= x >> np.uint64(16)
    return x


def step_salt(step):
    """common.h step_salt(): what the kernels add to the key when a device step counter is attached (None -> 0)."""
    if step is None:
        return 0
    x = np.array([(int(step) * 0x9E3779B1 + 0x7F4A7C15) & 0xFFFFFFFF], dtype=np.uint64)
    return int(drop_mix(x)[0])


def make_drop(p, seed, stream, step=None):
    """-> (key, thresh16, scale) exactly as make_drop() in common.h (+ the in-kernel DROP_STEP salt)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    key = ((seed & 0xFFFFFFFF) ^ (((seed >> 32) * 0x9E3779B1) & 0xFFFFFFFF)) + (int(stream) * 0x85EBCA6B)
    key = (key + (step_salt(step) if p > 0.0 else 0)) & 0xFFFFFFFF
    key &= 0xFFFFFFFF
    if p <= 0.0:
        return key, 0, 1.0
    t = int(float(p) * 65536.0 + 0.5)
    t = min(max(t, 1), 65535)
    return key, t, np.float32(65536.0) / (np.float32(65536.0) - np.float32(t))


def keep_mask(n, p, seed, stream, step=None):
    """Boolean keep mask of the first n elements (n % 4 == 0) of a flat tensor, and the keep scale."""
    assert n % 4 == 0
    key, thresh, scale = make_drop(p, seed, stream, step)
    if thresh == 0:
        return np.ones(n, dtype=bool), 1.0
    idx4 = np.arange(n // 4, dtype=np.uint64)
    c = ((idx4 << np.uint64(1)) & M32) ^ (((idx4 >> np.uint64(31)) * np.uint64(0xC2B2AE35)) & M32)
    h0 = drop_mix((np.uint64(key) + c * np.uint64(0x9E3779B1)) & M32)
    h1 = drop_mix((np.uint64(key) + ((c + np.uint64(1)) & M32) * np.uint64(0x9E3779B1)) & M32)
    u = np.stack([h0 & np.uint64(0xFFFF), h0 >> np.uint64(16), h1 & np.uint64(0xFFFF), h1 >> np.uint64(16)], 1).reshape(-1)
    return u >= np.uint64(thresh), float(scale)


# ---- attention-probability dropout (csrc/attention.hip: make_attn_drop, mix24, drop_sel) ---------------------------
def _attn_mix24(x):
    x = x.astype(np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = _mul24(x, 0x7FEB35)
    x ^= x >> np.uint64(15)
    x = _mul24(x, 0x6CA68B)
    return x


def attn_keep_mask(B, H, Lq, Lk, p, seed, stream, step=None):
    """-> (keep[B, H, Lq, Lk] bool, scale).  Element (q, k) of head-matrix (b, h) takes byte (k & 3) of the word
    mix24(seed' + bh * CB + q * CQ + (k >> 2) * CK): one mix per query row and group of four consecutive keys; keep iff
    byte >= thresh8 with thresh8 = round(256 p) and scale = 256 / (256 - thresh8)  (csrc/attention.hip, round 3)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    s32 = (((seed & 0xFFFFFFFF) ^ (seed >> 32)) + int(stream) * 0x27D4EB2F + step_salt(step)) & 0xFFFFFFFF
    if p <= 0.0:
        return np.ones((B, H, Lq, Lk), dtype=bool), 1.0
    t8 = min(int(float(p) * 256.0 + 0.5), 255)
    scale = 256.0 / (256.0 - t8)
    bh = (np.arange(B * H, dtype=np.uint64) * np.uint64(0xC2B2AE35)).reshape(B, H, 1, 1)
    q = np.arange(Lq, dtype=np.uint64).reshape(1, 1, Lq, 1)
    k = np.arange(Lk, dtype=np.uint64).reshape(1, 1, 1, Lk)
    x = (np.uint64(s32) + bh + q * np.uint64(0x9E3779B1) + (k >> np.uint64(2)) * np.uint64(0x85EBCA6B)) & M32
    h = _attn_mix24(x)
    byte = (h >> (np.uint64(8) * (k & np.uint64(3)))) & np.uint64(0xFF)
    return byte >= np.uint64(t8), scale
