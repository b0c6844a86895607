"""Host-side threefry2x32 for the few scalar PRNG uses outside the device kernel:
the adaptive seed chain (cpflow/main.py:798-799) and `Ansatz.learn`'s repeat chain
(cpflow/optimization.py:306-311).  The bulk draw of initial angles for the raw stage
happens on the GPU (`k_initial_angles`, cpflow_amd/csrc/cpf_device.h)."""
import numpy as np

_M = 0xFFFFFFFF
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _threefry(k0, k1, x0, x1):
    ks = (k0, k1, k0 ^ k1 ^ 0x1BD11BDA)
    x0, x1 = (x0 + ks[0]) & _M, (x1 + ks[1]) & _M
    for i in range(5):
        for r in _ROT[i & 1]:
            x0 = (x0 + x1) & _M
            x1 = ((x1 << r) | (x1 >> (32 - r))) & _M
            x1 ^= x0
        x0 = (x0 + ks[(i + 1) % 3]) & _M
        x1 = (x1 + ks[(i + 2) % 3] + i + 1) & _M
    return x0, x1


def _bits(key, n):
    """threefry_2x32(key, iota(n)) as a python list."""
    count = list(range(n)) + ([0] if n % 2 else [])
    h = len(count) // 2
    pairs = [_threefry(key[0], key[1], count[i], count[h + i]) for i in range(h)]
    return ([p[0] for p in pairs] + [p[1] for p in pairs])[:n]


def prng_key(seed):
    return (0, int(seed) & _M)


def split(key, num=2):
    b = _bits(key, 2 * num)
    return [(b[2 * i], b[2 * i + 1]) for i in range(num)]


def uniform_angles(key, n):
    bits = np.array(_bits(key, n), dtype=np.uint32)
    f = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    return np.maximum(np.float32(0), f * np.float32(2 * np.pi)).astype(np.float32)


def next_adaptive_seed(seed):
    """seed <- int(split(PRNGKey(seed))[1][1])."""
    return int(split(prng_key(seed), 2)[1][1])


def repeats_angles(num_repeats, num_angles):
    key = prng_key(0)
    out = np.empty((num_repeats, num_angles), dtype=np.float32)
    for i in range(num_repeats):
        key, sub = split(key, 2)
        out[i] = uniform_angles(sub, num_angles)
    return out
