"""Counter-based noise generator (csrc/rng_kernels.hip, dge_randn) and its oracle (oracle/philox_ref.py).

CPU: the oracle's Philox4x32-10 against the Random123 known-answer vectors (the published test vectors of the algorithm),
and the slicing property the data-parallel step relies on.  GPU: the kernel against the oracle and the same property on the
device, bit-exact: the rows a rank draws are the rows of the global-batch draw."""
import numpy as np
import pytest
import torch

from oracle import philox_ref as PR

KAT = [   # counter (4 words), key (2 words), expected output: Random123 kat_vectors, philox4x32 10 rounds
    ([0, 0, 0, 0], (0, 0), [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, (0xffffffff, 0xffffffff), [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0), [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def test_oracle_philox_known_answers():
    for ctr, key, want in KAT:
        got = PR.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert [int(v) for v in got] == want


def test_oracle_slices_are_rows_of_the_global_draw():
    n_row, B, world = 37 * 3, 2, 3           # odd row length: quads straddle the rank boundaries
    full = PR.randn(1234, 5, 0, world * B * n_row)
    for r in range(world):
        part = PR.randn(1234, 5, r * B * n_row, B * n_row)
        assert np.array_equal(part, full[r * B * n_row:(r + 1) * B * n_row])
    assert not np.array_equal(PR.randn(1234, 6, 0, 64), PR.randn(1234, 5, 0, 64))        # another draw number
    assert not np.array_equal(PR.randn(1235, 5, 0, 64), PR.randn(1234, 5, 0, 64))        # another seed
    x = PR.randn(7, 0, 0, 1 << 18)
    assert abs(float(x.mean())) < 0.01 and abs(float(x.std()) - 1.0) < 0.01 and float(np.abs(x).max()) < 7.0


@pytest.mark.gpu
def test_device_noise_matches_oracle_and_slices_bit_exactly():
    from dge_amd import ops
    dev = "cuda"
    shapes = [(2, 1, 64, 64), (2, 1, 32, 32), (2, 3, 5, 7), (2, 512)]
    try:
        ops.noise_dp(0, 1)
        ops.noise_seed(77)
        # the global batch of a 3-rank run drawn by ONE process: 6 rows
        glob = [t.cpu().numpy() for t in ops.randn_rows([(6,) + s[1:] for s in shapes], dev)]
        for k, gt in enumerate(glob):
            want = PR.randn(77, k, 0, gt.size).reshape(gt.shape)
            assert np.abs(gt - want).max() < 2e-6, k                    # libm differences of logf / sincospif only
        for rank in range(3):
            ops.noise_dp(rank, 3)
            ops.noise_seed(77)
            part = [t.cpu().numpy() for t in ops.randn_rows(shapes, dev)]
            for k, (pt, gt) in enumerate(zip(part, glob)):
                assert np.array_equal(pt, gt[2 * rank:2 * rank + 2]), (rank, k)      # bit-exact rows of the global draw
        # consecutive calls continue the draw numbering; re-seeding restarts it
        ops.noise_dp(0, 1)
        ops.noise_seed(5)
        a = ops.randn((4, 100), dev).cpu()
        b = ops.randn((4, 100), dev).cpu()
        ops.noise_seed(5)
        a2 = ops.randn((4, 100), dev).cpu()
        assert torch.equal(a, a2) and not torch.equal(a, b)
        big = ops.randn((1 << 22,), dev)
        assert abs(float(big.mean())) < 3e-3 and abs(float(big.std()) - 1) < 3e-3
    finally:
        ops.noise_dp(0, 1)
