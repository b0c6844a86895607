"""ctypes binding of the C ABI declared in include/cpflow_amd.h.

The HIP library is the product; there is no CPU fallback.  Importing this module
loads `cpflow_amd/libcpflow_amd.so` (built in-tree by `__graft_entry__.build()` /
`python -m cpflow_amd.build`) and fails loudly when it is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CPF_LIB') or os.path.join(_HERE, 'libcpflow_amd.so')   # CPF_LIB: A/B builds

ENTANGLER = {'cp': 0, 'cz': 1, 'cx': 2}
PENALTY = {'linear': 0, 'L1': 1}
LOSS = {'hst': 0, 'state': 1, 'relative_phase': 2}


class Penalty(C.Structure):
    _fields_ = [('kind', C.c_int32), ('xmax', C.c_double), ('ymax', C.c_double),
                ('plato_0', C.c_double), ('plato_1', C.c_double), ('plato_2', C.c_double)]


class Problem(C.Structure):
    _fields_ = [('n_qubits', C.c_int32), ('num_blocks', C.c_int32), ('placements', C.c_void_p),
                ('n_rot', C.c_int32), ('rot_axes', C.c_char_p), ('entangler', C.c_int32),
                ('loss_kind', C.c_int32), ('target', C.c_void_p), ('penalty', Penalty),
                ('use_reg', C.c_int32), ('r', C.c_double),
                ('lr', C.c_double), ('b1', C.c_double), ('b2', C.c_double), ('eps', C.c_double),
                ('num_iterations', C.c_int32), ('threshold_cp', C.c_double), ('entry_loss', C.c_double)]


class Result(C.Structure):
    _fields_ = [('params', C.c_void_p), ('regloss', C.c_void_p), ('loss', C.c_void_p), ('reg', C.c_void_p),
                ('cz', C.c_void_p), ('passed', C.c_void_p)]


# every symbol include/cpflow_amd.h declares: name -> (restype, argtypes)
_VP, _I32, _I64, _U32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
SYMBOLS = {
    'cpf_version': (_I32, []),
    'cpf_device_count': (_I32, []),
    'cpf_last_error': (C.c_char_p, []),
    'cpf_plan_create': (_I32, [C.POINTER(Problem), _I32, C.POINTER(_VP)]),
    'cpf_plan_destroy': (_I32, [_VP]),
    'cpf_plan_num_params': (_I32, [_VP]),
    'cpf_plan_num_gates': (_I32, [_VP]),
    'cpf_plan_cp_mask': (_I32, [_VP, _VP]),
    'cpf_plan_flops_per_sample_step': (C.c_double, [_VP]),
    'cpf_initial_angles_device': (_I32, [_VP, _U32, _I64, _I64, _I64, _I32, _VP, _VP]),
    'cpf_optimize_device': (_I32, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    'cpf_loss_grad_device': (_I32, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP]),
    'cpf_unitary_device': (_I32, [_VP, _I64, _VP, _VP, _VP]),
    'cpf_reg_device': (_I32, [_VP, _I64, _VP, _VP, _VP]),
    'cpf_plan_last_kernel_ms': (_I32, [_VP, C.POINTER(C.c_float)]),
    'cpf_plan_last_variant': (_I32, [_VP, C.c_char_p, C.c_size_t]),
    'cpf_jit_compiles': (_I32, []),
    'cpf_set_tuning': (_I32, [C.c_char_p, C.c_char_p]),
    'cpf_jit_precompile': (_I32, [C.POINTER(Problem), _I64, C.c_char_p, C.c_size_t]),
    'cpf_run_static': (_I32, [C.POINTER(Problem), _I32, _U32, _I64, _I64, _I64, _I32, _VP, _VP, C.POINTER(Result)]),
}

_lib = None


class CpfError(RuntimeError):
    pass


def lib():
    """The loaded HIP library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CpfError(
                f'{LIB_PATH} not found: the HIP extension is not built. '
                'Run `python -c "import __graft_entry__ as g; g.build()"` (or `python -m cpflow_amd.build`). '
                'cpflow_amd has no CPU fallback.')
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)      # AttributeError if the ABI and the header drifted apart
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().cpf_last_error().decode(errors='replace')
        if rc < 0:
            raise ValueError(f'cpflow_amd: {msg} (code {rc})')
        raise CpfError(f'cpflow_amd: {msg} (hip error {rc})')


def set_tuning(name, value=None):
    """Kernel-selection switch of the library (tests, A/B tools): value None restores the default."""
    check(lib().cpf_set_tuning(name.encode(), None if value is None else str(value).encode()))
