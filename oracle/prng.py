"""ORACLE (test infrastructure, never imported by the product path).

NumPy restatement of the JAX 0.3.x threefry2x32 PRNG calls the reference makes
when it samples initial angles.  jax/jaxlib (pinned `jax>=0.3.0,<0.3.5` in
/root/reference/setup.py:19-20) are third-party and absent from this image, so
the published Threefry-2x32 algorithm (Salmon et al., "Parallel random numbers:
as easy as 1, 2, 3", 20 rounds, as used by jax.random) is restated here and
pinned bit-exactly against the initial-angle rows stored in the reference's own
result files (tests/golden/*.npz, tests/test_oracle_golden.py).

Reference call sites restated:
  random.PRNGKey / random.split / random.uniform
      /root/reference/cpflow/main.py:541-548   (Synthesize._generate_initial_angles)
      /root/reference/cpflow/cp_utils.py:13-42 (random_cp_angles)
      /root/reference/cpflow/trigonometric_utils.py:35-38 (random_angles)
      /root/reference/cpflow/main.py:798-799   (adaptive seed chain)
      /root/reference/cpflow/optimization.py:306-311 (num_repeats chain)
"""
import numpy as np

_U32 = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def prng_key(seed):
    """random.PRNGKey(seed) with x64 disabled: uint32[2] = {0, seed mod 2^32}."""
    return np.array([0, int(seed) & 0xFFFFFFFF], dtype=_U32)


def _rotl(x, r):
    return (x << _U32(r)) | (x >> _U32(32 - r))


def threefry2x32(key, x0, x1):
    """One Threefry-2x32 (20 rounds) evaluation on uint32 arrays x0, x1."""
    with np.errstate(over='ignore'):
        k0, k1 = _U32(key[0]), _U32(key[1])
        ks = (k0, k1, _U32(k0 ^ k1 ^ _U32(0x1BD11BDA)))
        x0 = (x0.astype(_U32) + ks[0]).astype(_U32)
        x1 = (x1.astype(_U32) + ks[1]).astype(_U32)
        for i in range(5):
            for r in _ROT[i % 2]:
                x0 = (x0 + x1).astype(_U32)
                x1 = _rotl(x1, r)
                x1 = x1 ^ x0
            x0 = (x0 + ks[(i + 1) % 3]).astype(_U32)
            x1 = (x1 + ks[(i + 2) % 3] + _U32(i + 1)).astype(_U32)
    return x0, x1


def random_bits(key, n):
    """threefry_2x32(key, iota(n)): counts padded to even length, split in halves."""
    count = np.arange(n, dtype=_U32)
    if n % 2:
        count = np.concatenate([count, np.zeros(1, dtype=_U32)])
    h = count.size // 2
    a, b = threefry2x32(key, count[:h], count[h:])
    return np.concatenate([a, b])[:n]


def split(key, num=2):
    """random.split(key, num) -> uint32[num, 2]."""
    return random_bits(key, 2 * num).reshape(num, 2)


def uniform(key, n, minval=0.0, maxval=2 * np.pi):
    """random.uniform(key, (n,), float32, minval, maxval)."""
    bits = random_bits(key, n)
    f = ((bits >> _U32(9)) | _U32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(minval), np.float32(maxval)
    return np.maximum(lo, f * (hi - lo) + lo).astype(np.float32)


def initial_angles(seed, num_samples, num_angles, cp_mask=None, cp_dist='uniform'):
    """Synthesize._generate_initial_angles (main.py:541-548) for 'uniform' / '0'.

    keys = split(PRNGKey(seed), S+1)[1:]; per sample random_cp_angles does
    key, subkey = split(k); angles = uniform(subkey, P) * (1 - mask if '0').
    """
    keys = split(prng_key(seed), num_samples + 1)[1:]
    out = np.empty((num_samples, num_angles), dtype=np.float32)
    for j, k in enumerate(keys):
        sub = split(k, 2)[1]
        out[j] = uniform(sub, num_angles)
    if cp_dist == '0':
        out = out * (1 - np.asarray(cp_mask, dtype=np.float32))
    elif cp_dist != 'uniform':
        raise NotImplementedError(cp_dist)
    return out


def next_adaptive_seed(seed):
    """main.py:798-799: _, subkey = split(PRNGKey(seed)); seed = int(subkey[1])."""
    return int(split(prng_key(seed), 2)[1][1])


def repeats_angles(num_repeats, num_angles):
    """optimization.py:306-311: key=PRNGKey(0); per repeat key, subkey = split(key)."""
    key = prng_key(0)
    out = np.empty((num_repeats, num_angles), dtype=np.float32)
    for i in range(num_repeats):
        key, sub = split(key, 2)
        out[i] = uniform(sub, num_angles)
    return out
