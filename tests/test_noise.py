"""The sampling noise that replaces torch.multinomial's draws (network/pixelcnn/models.py:190-197): the numpy restatement
of the device Philox generator against the published known-answer vectors (CPU), the device kernel against the
restatement, and the property the multi-GPU contract needs: the noise of a row depends on its GLOBAL index only."""
import numpy as np
import pytest
import torch

from oracle import philox


def test_philox_known_answer_vectors():
    """Random123's kat_vectors for philox4x32-10 (counter, key) -> output."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox.philox4x32_10(np.array([ctr], dtype=np.uint64), key[0], key[1])[0]
        assert tuple(int(v) for v in got) == want


def test_noise_restatement_is_shard_invariant_and_exponential():
    full = philox.exp1_noise(64, 16, seed=1234, row0=0, stream_id=3)
    part = philox.exp1_noise(24, 16, seed=1234, row0=40, stream_id=3)
    assert np.array_equal(full[40:], part)
    assert not np.array_equal(full, philox.exp1_noise(64, 16, seed=1234, row0=0, stream_id=4))
    big = philox.exp1_noise(4096, 64, seed=7)
    assert big.min() > 0 and abs(big.mean() - 1.0) < 0.01 and abs(big.var() - 1.0) < 0.03      # Exp(1): mean 1, variance 1


@pytest.mark.gpu
def test_device_noise_matches_restatement_and_shards():
    from dvqvae_amd import ops
    dev = "cuda:0"
    rows, cols = 300, 9 * 512
    q = ops.exp1_noise(rows, cols, seed=(5 << 32) | 99, row0=1 << 33, stream_id=11, device=dev).cpu().numpy()
    ref = philox.exp1_noise(rows, cols, seed=(5 << 32) | 99, row0=1 << 33, stream_id=11)
    np.testing.assert_allclose(q, ref, rtol=2e-6, atol=1e-7)                 # same integers; libm's log differs by an ulp or two
    for R in (2, 4, 8):                                                      # contiguous shards reproduce the whole, bit for bit
        parts = []
        for r in range(R):
            lo, hi = r * rows // R, (r + 1) * rows // R
            parts.append(ops.exp1_noise(hi - lo, cols, seed=(5 << 32) | 99, row0=(1 << 33) + lo, stream_id=11, device=dev))
        assert torch.equal(torch.cat(parts).cpu(), torch.from_numpy(q))
    # cols % 4 != 0 (a prior with n_in % 4 != 0): the generator's quads are drawn on a padded row and the columns asked for kept
    odd = ops.exp1_noise(4, 6, seed=0, device=dev)
    assert tuple(odd.shape) == (4, 6) and torch.equal(odd, ops.exp1_noise(4, 8, seed=0, device=dev)[:, :6])
