"""TEST INFRASTRUCTURE ONLY.  numpy restatement of the counter-based noise generator of csrc/rng_kernels.hip
(dge_randn): Philox4x32-10 (Salmon et al., SC'11 - a published algorithm, third-party to the reference, which draws its noise
with torch.randn: model/E/E.py:60,73, model/stylegan2_generator.py:187,911-913) + Box-Muller.

Pinned: `philox4x32_10` reproduces the Random123 known-answer vectors (tests/test_rng.py, kat_vectors of the Random123
distribution).  The normals are a floating-point map of those integers; the device result is compared within 2e-6.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(ctr, key):
    """ctr: uint32 [..., 4], key: (k0, k1) python ints -> uint32 [..., 4]"""
    c = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = key
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        n0 = (p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)
        c = [n0 & mask, p1 & mask, n2 & mask, p0 & mask]
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


def randn(seed, subseq, goff, count):
    """elements [goff, goff+count) of draw `subseq` under `seed` (float32 array)"""
    q0, q1 = goff >> 2, (goff + count + 3) >> 2
    q = np.arange(q0, q1, dtype=np.uint64)
    ctr = np.zeros((q.size, 4), dtype=np.uint32)
    ctr[:, 0] = (q & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[:, 1] = (q >> np.uint64(32)).astype(np.uint32)
    ctr[:, 2] = subseq
    x = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    u = ((x.astype(np.float32) + np.float32(0.5)) * np.float32(2.3283064365386963e-10)).astype(np.float32)
    out = np.empty((q.size, 4), dtype=np.float32)
    for a in (0, 2):
        r = np.sqrt(np.float32(-2.0) * np.log(u[:, a])).astype(np.float32)
        ang = (np.float32(2.0) * u[:, a + 1]).astype(np.float64) * np.pi
        out[:, a] = r * np.cos(ang).astype(np.float32)
        out[:, a + 1] = r * np.sin(ang).astype(np.float32)
    flat = out.reshape(-1)
    lo = goff - (q0 << 2)
    return flat[lo:lo + count]
