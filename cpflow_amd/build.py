"""Build the HIP library in-tree:  python -m cpflow_amd.build

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels to the
GPU box with the source snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'cpflow_amd.hip')
DEPS = [SRC] + [os.path.join(HERE, 'csrc', h) for h in ('cpf_device.h', 'cpf_blocks.h', 'cpf_layered.h', 'cpf_split.h', 'cpf_dense.h', 'cpf_dense_math.h', 'cpf_dense_asm.h', 'cpf_dense_aot.h', 'cpf_dense_aot.hip', 'cpf_zyz.h', 'cpf_zyzl.h', 'cpf_zyz3.h', 'cpf_zyz_math.h', 'cpf_zyz_asm.h', 'cpf_zyz_aot.h', 'cpf_zyz_aot.hip', 'cpf_zyz3_aot.h', 'cpf_zyz3_aot.hip', 'cpf_jit.h')
                if os.path.exists(os.path.join(HERE, 'csrc', h))] + [os.path.join(os.path.dirname(HERE), 'include', 'cpflow_amd.h')]
JIT_DIR = os.path.join(HERE, 'jit_cache')      # shipped code objects of runtime-specialised kernels (git-ignored, travels with gpurun)
OUT = os.path.join(HERE, 'libcpflow_amd.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -fno-slp-vectorize: hipcc otherwise packs the complex arithmetic into v_pk_*_f32 with extra moves,
# which measured 5 % slower than the scalar f32 stream and blocks folding DPP reads into the VALU ops
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize', '-fno-signed-zeros', '-fPIC', '-shared', '-Wall',
         '-Wno-unused-function', '-Wno-pass-failed']        # pass-failed: '#pragma unroll' on the run-time-length variants of the 4x4 walks


def up_to_date():
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)


# the dense-block kernels are their own translation unit: 32 amplitudes per thread only stay in registers with raised
# alloca-promotion limits (cpf_dense_aot.h), which the other kernels are not built with
DENSE_SRC = os.path.join(HERE, 'csrc', 'cpf_dense_aot.hip')
ZYZ_SRC = os.path.join(HERE, 'csrc', 'cpf_zyz_aot.hip')
ZYZ3_SRC = os.path.join(HERE, 'csrc', 'cpf_zyz3_aot.hip')
DENSE_FLAGS = ['-mllvm', '-amdgpu-promote-alloca-to-vector-max-regs=128', '-mllvm', '-amdgpu-promote-alloca-to-vector-vgpr-ratio=1']


def build(force=False, verbose=False):
    if not force and up_to_date():
        return OUT
    cflags = [f for f in FLAGS if f != '-shared']
    objs = [(SRC, os.path.join(HERE, 'csrc', 'cpflow_amd.o'), []), (DENSE_SRC, os.path.join(HERE, 'csrc', 'cpf_dense_aot.o'), DENSE_FLAGS),
            (ZYZ_SRC, os.path.join(HERE, 'csrc', 'cpf_zyz_aot.o'), DENSE_FLAGS), (ZYZ3_SRC, os.path.join(HERE, 'csrc', 'cpf_zyz3_aot.o'), [])]
    procs = []
    for src, obj, extra in objs:          # the two units compile side by side
        cmd = [HIPCC] + cflags + extra + ['-c', '-o', obj, src]
        if verbose:
            print(' '.join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait():
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + [o for _, o, _ in objs] + ['-lhiprtc']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return OUT


def build_variant(name, defines=(), verbose=False):
    """A/B build: the library compiled with extra -D flags as cpflow_amd/libcpflow_amd.<name>.so (same directory, so that
    its runtime-specialised kernels find csrc/; select it with CPF_LIB; tools/gpu.py ablib compares two builds)."""
    out = os.path.join(HERE, f'libcpflow_amd.{name}.so')
    cflags = [f for f in FLAGS if f != '-shared'] + [f'-D{d}' for d in defines]
    objs = [(SRC, os.path.join(HERE, 'csrc', f'cpflow_amd.{name}.o'), []), (DENSE_SRC, os.path.join(HERE, 'csrc', f'cpf_dense_aot.{name}.o'), DENSE_FLAGS),
            (ZYZ_SRC, os.path.join(HERE, 'csrc', f'cpf_zyz_aot.{name}.o'), DENSE_FLAGS), (ZYZ3_SRC, os.path.join(HERE, 'csrc', f'cpf_zyz3_aot.{name}.o'), [])]
    procs = [subprocess.Popen([HIPCC] + cflags + extra + ['-c', '-o', obj, src]) for src, obj, extra in objs]
    for p in procs:
        if p.wait():
            raise RuntimeError(f'variant {name} failed to compile')
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + [o for _, o, _ in objs] + ['-lhiprtc'])
    for _, o, _ in objs:
        os.unlink(o)
    return out


def precompile_shapes():
    """(layer, rotation strings, block counts, sample counts) of the reference's tutorial / paper runs whose kernels are
    runtime-specialised: the 4-qubit star of tutorial/results/toff4_star, connected / chain layers on 4 and 5 qubits
    (paper/results/benchmarks, 'xyz' and 'xz'), twisted 3-qubit layers.  The kernel a launch uses depends on the
    rotation string, the layer, the angle count per owner thread (bucketed) and on whether the samples fill the chip."""
    star4, conn4, chain4 = [[0, 1], [0, 2], [0, 3]], [[i, j] for i in range(4) for j in range(i + 1, 4)], [[0, 1], [1, 2], [2, 3]]
    conn5, chain5 = [[i, j] for i in range(5) for j in range(i + 1, 5)], [[i, i + 1] for i in range(4)]
    shapes = []
    for layer in (star4, conn4, chain4):
        shapes += [(layer, rot, K, S, 'cp') for rot in ('xyz', 'xz') for K in (14, 23, 29, 40) for S in (100, 2048)]
    for layer in (conn5, chain5):
        shapes += [(layer, rot, K, S, 'cp') for rot in ('xyz', 'xz') for K in (30, 57, 86) for S in (200, 2000)]
    # Ansatz(3, 'cz', fill_layers(connected_layer(3), k)).learn(...) with 1000 repeats (paper/results/local_minimums)
    conn3, chain3 = [[0, 1], [0, 2], [1, 2]], [[0, 1], [1, 2]]
    shapes += [(layer, 'xyz', K, 1000, 'cz') for layer in (conn3, chain3) for K in (2, 4, 6, 8, 10, 12, 14, 16)]
    # 3 qubits, the block counts an adaptive search over the reference's layers visits (tutorial/results/toff3_chain:
    # 10..20): the two-ended ZYZ-form kernel (cpf_zyz3.h) is specialised on block count, cut and layer, and
    # cpf_jit_precompile builds all four launch regimes' cuts of a shape whatever the sample count named here, so one
    # entry per shape ships everything a launch of that shape can ask for (nothing compiles on a GPU box).  A few
    # seconds of hipcc each.  (The 'cz' template shapes of the local-minimum study are in the list above.)
    shapes += [(layer, rot, K, 1000, 'cp') for layer in (chain3, conn3) for rot in ('xyz', 'xz') for K in range(6, 25)]
    return shapes


def _headers_tag():
    """Content hash of the kernel headers (a checkout or a touch changes modification times, not this)."""
    import hashlib
    h = hashlib.sha1()
    for d in sorted(d for d in DEPS if d.endswith('.h') and os.sep + 'csrc' + os.sep in d):
        with open(d, 'rb') as f:
            h.update(os.path.basename(d).encode() + b'\0' + f.read())
    return h.hexdigest()


def precompile(shapes=None, workers=None, verbose=True):
    """Fill cpflow_amd/jit_cache/ with the runtime-specialised kernels of `shapes` (no GPU needed: hipcc
    cross-compiles), so that a GPU box never compiles them.  One process per shape, `workers` at a time."""
    import json
    from concurrent.futures import ThreadPoolExecutor
    build()
    os.makedirs(JIT_DIR, exist_ok=True)
    os.chmod(JIT_DIR, 0o700)
    # code objects are keyed by a hash of the kernel headers: after a header edit every shipped object is stale
    # (the kernel headers under csrc/: the public header include/cpflow_amd.h is not part of any kernel)
    stamp, tag = os.path.join(JIT_DIR, '.stamp'), _headers_tag()
    if not os.path.exists(stamp) or open(stamp).read() != tag:
        for f in os.listdir(JIT_DIR):
            if f.endswith('.hsaco') or '.tmp' in f:
                os.unlink(os.path.join(JIT_DIR, f))
        with open(stamp, 'w') as f:
            f.write(tag)
    shapes = precompile_shapes() if shapes is None else shapes
    env = dict(os.environ, CPF_JIT_CACHE=JIT_DIR)

    def batch(chunk):           # one process per few shapes (start-up costs as much as a cached look-up); a chunk shares its tuning
        code = ('import json, sys; from cpflow_amd.build import _precompile_one; '
                'print(json.dumps([_precompile_one(*s) for s in json.loads(sys.argv[1])]))')
        r = subprocess.run([sys.executable, '-c', code, json.dumps(chunk)], env=env, capture_output=True, text=True,
                           cwd=os.path.dirname(HERE))
        if r.returncode:
            raise RuntimeError(f'precompile of {chunk} failed: {r.stderr[-400:]}')
        return json.loads(r.stdout.strip().splitlines()[-1])
    groups = {}
    for shape in shapes:
        groups.setdefault(json.dumps(shape[5] if len(shape) > 5 else None, sort_keys=True), []).append(shape)
    chunks = [g[i:i + 4] for g in groups.values() for i in range(0, len(g), 4)]
    seen = set()
    with ThreadPoolExecutor(max_workers=workers or max(1, (os.cpu_count() or 2) - 1)) as ex:
        for per_shape in ex.map(batch, chunks):
            for line in (x for lines in per_shape for x in lines):
                if verbose and line not in seen:
                    print(line)
                seen.add(line)
    # The offline compiler builds every shipped kernel.  cpf_jit falls back to hiprtc when hipcc fails - right for a user's machine,
    # wrong here: round 6 shipped two hiprtc objects (2 x slower code) for instances hipcc's backend aborted on, and nothing said so.
    slow = sorted(x for x in seen if 'jit(hiprtc)' in x)
    if slow and os.environ.get('CPF_JIT_BACKEND') != 'hiprtc':
        raise RuntimeError('hipcc failed on shipped kernels (see the *.tmp.log files in %s): %s' % (JIT_DIR, '; '.join(slow)))
    return sorted(seen)


def _precompile_one(layer, rot, K, S, entangler='cp', tuning=None):
    import ctypes as C
    import numpy as np
    from . import _ffi
    from .engine import placements_for, problem_struct
    for name, value in (tuning or {}).items():
        _ffi.set_tuning(name, value)
    n = max(max(p) for p in layer) + 1
    pl, target, rotb = placements_for(layer, K), np.ascontiguousarray(np.eye(1 << n, dtype=np.complex128)), rot.encode()
    pb = problem_struct(n, pl, rotb, entangler, 'hst', target)
    buf = C.create_string_buffer(4096)
    rc = _ffi.lib().cpf_jit_precompile(C.byref(pb), int(S), buf, 4096)
    if rc < 0:
        raise RuntimeError(_ffi.lib().cpf_last_error().decode())
    return [x for x in buf.value.decode().splitlines() if x]


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
    if '--precompile' in sys.argv:
        precompile()
