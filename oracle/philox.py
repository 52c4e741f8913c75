"""Test infrastructure: numpy restatement of the device noise generator (dvq_exp1_noise, d-vqvae_amd/csrc/misc.hip):
Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; the constants are the published ones), counter = (column quad, row low,
row high, stream), key = (seed low, seed high); u = (top 24 bits + 0.5) 2^-24; q = -log(u).  The reference draws with
torch.multinomial (network/pixelcnn/models.py:195), which cannot be reproduced across devices; the port replaces it by the
exponential race argmax p/q with THIS q so that sharded runs reproduce unsharded ones.  Only tests import this file."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c, k0, k1):
    """c: uint64 array [..., 4] holding 32-bit words; returns the ten-round Philox output (uint32 words in uint64)."""
    c = c.astype(np.uint64).copy()
    k0, k1 = np.uint64(k0), np.uint64(k1)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(M0) * c[..., 0]
        p1 = np.uint64(M1) * c[..., 2]
        n0 = ((p1 >> np.uint64(32)) ^ c[..., 1] ^ k0) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c[..., 3] ^ k1) & mask
        n3 = p0 & mask
        c = np.stack([n0, n1, n2, n3], axis=-1)
        k0 = (k0 + np.uint64(W0)) & mask
        k1 = (k1 + np.uint64(W1)) & mask
    return c


def exp1_noise(rows, cols, seed, row0=0, stream_id=0):
    """float32 [rows, cols], the values dvq_exp1_noise(seed, stream_id, row0, rows, cols) produces (up to libm's log)."""
    assert cols % 4 == 0
    r = np.arange(rows, dtype=np.uint64)[:, None] + np.uint64(row0)
    q = np.arange(cols // 4, dtype=np.uint64)[None, :]
    c = np.stack([np.broadcast_to(q, (rows, cols // 4)), np.broadcast_to(r & np.uint64(0xFFFFFFFF), (rows, cols // 4)),
                  np.broadcast_to(r >> np.uint64(32), (rows, cols // 4)), np.full((rows, cols // 4), stream_id, dtype=np.uint64)], axis=-1)
    x = philox4x32_10(c, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = ((x >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    return (-np.log(u.astype(np.float64))).astype(np.float32).reshape(rows, cols)
