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


def erfinv_f32(x):
    """lax.erf_inv in float32 as XLA lowers it (xla/client/lib/math.cc ErfInv32, Giles' single-precision polynomial in
    w = -log((1 - x)(1 + x))).  Third-party arithmetic, restated from the published algorithm; PARITY UNPINNED: no result
    file of the reference uses cp_distribution='normal'."""
    f = np.float32
    x = np.asarray(x, dtype=f)
    w = -np.log1p(-x * x).astype(f)
    lt = w < f(5)
    w = np.where(lt, w - f(2.5), np.sqrt(w, dtype=f) - f(3)).astype(f)
    ca = [2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503, -0.00417768164,
          0.246640727, 1.50140941]
    cb = [-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613, 0.00943887047,
          1.00167406, 2.83297682]
    p = np.where(lt, f(ca[0]), f(cb[0])).astype(f)
    for a, b in zip(ca[1:], cb[1:]):
        p = (np.where(lt, f(a), f(b)) + p * w).astype(f)
    return (p * x).astype(f)


def normal(key, n):
    """random.normal(key, (n,), float32) of jax 0.3.x (_normal_real): sqrt(2) * erf_inv(uniform(key, nextafter(-1, 0), 1))."""
    lo = np.nextafter(np.float32(-1), np.float32(0))
    return (np.float32(np.sqrt(2)) * erfinv_f32(uniform(key, n, lo, np.float32(1)))).astype(np.float32)


def initial_angles(seed, num_samples, num_angles, cp_mask=None, cp_dist='uniform'):
    """Synthesize._generate_initial_angles (main.py:541-548).

    keys = split(PRNGKey(seed), S+1)[1:]; per sample random_cp_angles (cp_utils.py:31-40) does
    key, subkey = split(k); angles = uniform(subkey, P); '0': * (1 - mask); 'normal': key, subkey = split(key) again and
    angles * (1 - mask) + 1.5 * normal(subkey, P) * mask.
    """
    keys = split(prng_key(seed), num_samples + 1)[1:]
    out = np.empty((num_samples, num_angles), dtype=np.float32)
    mask = None if cp_mask is None else np.asarray(cp_mask, dtype=np.float32)
    for j, k in enumerate(keys):
        key, sub = split(k, 2)
        out[j] = uniform(sub, num_angles)
        if cp_dist == 'normal':
            sub2 = split(key, 2)[1]
            out[j] = out[j] * (1 - mask) + np.float32(1.5) * normal(sub2, num_angles) * mask
    if cp_dist == '0':
        out = out * (1 - mask)
    elif cp_dist not in ('uniform', 'normal'):
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
