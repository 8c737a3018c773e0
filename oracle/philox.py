"""Oracle: Philox4x32-10 + Box-Muller, numpy restatement of the in-kernel noise generator
(test infrastructure only).

The reference has no counter-based RNG (it draws from torch's global generator,
diffusion/gaussian_diffusion.py:532,694,773); this generator is the build's own, used for
throughput runs so that results do not depend on how a batch is sharded.  The oracle restates
the published Philox4x32-10 algorithm (Salmon et al., SC'11; same constants as Random123 /
cuRAND) and is pinned by the Random123 known-answer vectors in tests/test_philox.py.
Keying: counter = (element_group, draw_step, sample_lo32, sample_hi32), key = (seed_lo32, seed_hi32).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(counter, key):
    """counter: uint32 [..., 4], key: (k0, k1) uint32 scalars -> uint32 [..., 4]."""
    c = [counter[..., i].astype(np.uint32) for i in range(4)]
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c[0].astype(np.uint64)
            p1 = M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return np.stack(c, axis=-1)


def normal(batch, per_sample, seed, sample_offset=0, step=0):
    """[batch, per_sample] float32 N(0,1), element e of sample b from group e//4, slot e%4."""
    groups = (per_sample + 3) // 4
    out = np.empty((batch, groups * 4), dtype=np.float32)
    grp = np.arange(groups, dtype=np.uint32)
    for b in range(batch):
        sample = np.uint64(sample_offset + b)
        ctr = np.stack([grp, np.full(groups, step, np.uint32), np.full(groups, np.uint32(sample & np.uint64(0xFFFFFFFF))),
                        np.full(groups, np.uint32(sample >> np.uint64(32)))], axis=-1)
        r = philox4x32_10(ctr, (np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)))
        inv24 = np.float32(1.0 / 16777216.0)
        u0 = ((r[:, 0] >> np.uint32(8)) + np.uint32(1)).astype(np.float32) * inv24
        u1 = (r[:, 1] >> np.uint32(8)).astype(np.float32) * inv24
        u2 = ((r[:, 2] >> np.uint32(8)) + np.uint32(1)).astype(np.float32) * inv24
        u3 = (r[:, 3] >> np.uint32(8)).astype(np.float32) * inv24
        r0 = np.sqrt(np.float32(-2.0) * np.log(u0)).astype(np.float32)
        r1 = np.sqrt(np.float32(-2.0) * np.log(u2)).astype(np.float32)
        a0 = np.float32(6.28318530717958647692) * u1
        a1 = np.float32(6.28318530717958647692) * u3
        z = np.stack([r0 * np.cos(a0), r0 * np.sin(a0), r1 * np.cos(a1), r1 * np.sin(a1)], axis=-1)
        out[b] = z.reshape(-1).astype(np.float32)
    return out[:, :per_sample]
