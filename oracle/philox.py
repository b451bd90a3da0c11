"""Philox4x32 (Salmon, Moraes, Dror, Shaw - "Parallel random numbers: as easy as 1, 2, 3", SC'11)
in NumPy, and the EMA stick-noise profile built on it, as a float64 checker of the in-kernel
generator (fpyv_amd/csrc/fpv_math.h: fpv_philox4x32<R> / fpv_normal_from_word / fpv_normal4 / fpv_stick_noise).

TEST INFRASTRUCTURE ONLY.  The integer generator is pinned by the known-answer vectors of the
Random123 distribution for 10 AND for 7 rounds (tests/test_stick_noise.py); the generator runs 7.  The float part
follows /root/reference/tests/noise_smooth_test.py:6-12 (x ~ N(0,1); x_s <- (1-tau) x_s + tau x); a normal is
the exact inverse CDF (scipy.special.ndtri, float64) of the tail probability the kernel reads off a random word - the
kernel's piecewise-cubic table must land within 3e-6 of it.
"""
import numpy as np

NOISE_ROUNDS = 7

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32(ctr, key, rounds=10):
    """ctr: [..., 4] uint32, key: [..., 2] uint32 -> [..., 4] uint32"""
    c = [np.asarray(ctr[..., i], dtype=np.uint32).copy() for i in range(4)]
    k0, k1 = np.asarray(key[..., 0], dtype=np.uint32).copy(), np.asarray(key[..., 1], dtype=np.uint32).copy()
    for _ in range(rounds):
        p0 = M0 * c[0].astype(np.uint64)
        p1 = M1 * c[2].astype(np.uint64)
        n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[1] ^ k0
        n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[3] ^ k1
        c = [n0, p1.astype(np.uint32), n2, p0.astype(np.uint32)]
        k0 = (k0 + W0).astype(np.uint32)
        k1 = (k1 + W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def philox4x32_10(ctr, key):
    return philox4x32(ctr, key, 10)


def normal_from_words(w):
    """float64 standard normals of 32-bit words: sign = top bit, tail probability p = float32((w & 0x7fffffff) | 1) / 2^32
    (the conversion's rounding is part of the definition), z = -+Phi^-1(p) exactly."""
    from scipy.special import ndtri
    w = np.asarray(w, dtype=np.uint32)
    k = (w & np.uint32(0x7FFFFFFF)) | np.uint32(1)
    p = k.astype(np.float32).astype(np.float64) * 2.0 ** -32
    z = -ndtri(np.minimum(p, 0.5))
    return np.where(w >> np.uint32(31), -z, z)


def normal4(seed, drone_ids, step):
    """[n, 4] float64 standard normals for global drone ids `drone_ids` at the 64-bit step index `step`
    (Philox counter = drone id low, high, step low, high; the high word is 0 below 2^32 steps)."""
    ids = np.asarray(drone_ids, dtype=np.uint64)
    n = ids.shape[0]
    step = int(step) & (2 ** 64 - 1)
    ctr = np.stack([(ids & np.uint64(0xFFFFFFFF)).astype(np.uint32), (ids >> np.uint64(32)).astype(np.uint32),
                    np.full(n, step & 0xFFFFFFFF, dtype=np.uint32), np.full(n, step >> 32, dtype=np.uint32)], axis=-1)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32), (n, 2))
    return normal_from_words(philox4x32(ctr, key, NOISE_ROUNDS))


def ema_sticks(seed, drone_ids, steps, tau=0.1, gain=1.0, base_action=None, step0=0):
    """[steps, n, 4] applied sticks and the final EMA state [n, 4] (float64)."""
    n = len(drone_ids)
    s = np.zeros((n, 4))
    out = np.empty((steps, n, 4))
    for t in range(steps):
        s = s * (1 - tau) + normal4(seed, drone_ids, step0 + t) * tau
        a = 0.0 if base_action is None else base_action[t]
        out[t] = np.clip(a + gain * s, -1.0, 1.0)
    return out, s
