"""Inert reader for the dill pickles the reference ships (fixture mining only).

The reference stores its `Results` objects with dill; they reference classes
from cpflow, jax, qiskit and hyperopt, none of which are importable here.
Nothing from such a pickle may be executed, so this reader rebuilds numpy
arrays and builtin containers only and turns every other global into an inert
`Stub` that merely records how it was constructed.

This is a test-infrastructure tool (used by tools/extract_golden.py, run once
in the build container where /root/reference exists). It is not part of the
product path and never travels to the GPU box as a dependency of anything.
"""
import importlib
import pickle

import numpy as np


class Stub:
    """Stands for any class / function / instance found in the pickle."""

    def __init__(self, name, args=(), kw=None):
        self._name, self._args, self._kw = name, tuple(args), kw or {}
        self._state = None
        self._items = []
        self._kv = {}

    def __call__(self, *a, **k):
        return Stub(self._name + '()', a, k)

    def append(self, x):
        self._items.append(x)

    def extend(self, xs):
        self._items.extend(xs)

    def add(self, x):
        self._items.append(x)

    def update(self, d):
        self._kv.update(d)

    def __setitem__(self, k, v):
        self._kv[k] = v

    def __hash__(self):
        return id(self)

    def __eq__(self, o):
        return self is o

    def __repr__(self):
        return f'<Stub {self._name}>'


_REAL = {
    ('numpy.core.multiarray', '_reconstruct'): ('numpy._core.multiarray', '_reconstruct'),
    ('numpy._core.multiarray', '_reconstruct'): ('numpy._core.multiarray', '_reconstruct'),
    ('numpy.core.multiarray', 'scalar'): ('numpy._core.multiarray', 'scalar'),
    ('numpy._core.multiarray', 'scalar'): ('numpy._core.multiarray', 'scalar'),
    ('numpy', 'ndarray'): ('numpy', 'ndarray'),
    ('numpy', 'dtype'): ('numpy', 'dtype'),
    ('collections', 'OrderedDict'): ('collections', 'OrderedDict'),
    ('builtins', 'complex'): ('builtins', 'complex'),
    ('builtins', 'set'): ('builtins', 'set'),
    ('builtins', 'frozenset'): ('builtins', 'frozenset'),
    ('builtins', 'slice'): ('builtins', 'slice'),
    ('builtins', 'range'): ('builtins', 'range'),
}


def _is_stub(x):
    return isinstance(x, Stub)


class InertUnpickler(pickle._Unpickler):
    dispatch = dict(pickle._Unpickler.dispatch)

    def find_class(self, module, name):
        if (module, name) in _REAL:
            m, n = _REAL[(module, name)]
            return getattr(importlib.import_module(m), n)
        return Stub(f'{module}.{name}')

    def load_reduce(self):
        args = self.stack.pop()
        func = self.stack[-1]
        if not _is_stub(func) and not any(_is_stub(a) for a in args):
            self.stack[-1] = func(*args)
        else:
            self.stack[-1] = Stub(f'{getattr(func, "_name", func)}()', args)
    dispatch[pickle.REDUCE[0]] = load_reduce

    def load_newobj(self):
        args = self.stack.pop()
        cls = self.stack.pop()
        if _is_stub(cls):
            self.append(Stub('new:' + cls._name, args))
        else:
            self.append(cls.__new__(cls, *args))
    dispatch[pickle.NEWOBJ[0]] = load_newobj

    def load_newobj_ex(self):
        kw = self.stack.pop()
        args = self.stack.pop()
        cls = self.stack.pop()
        if _is_stub(cls):
            self.append(Stub('new:' + cls._name, args, kw))
        else:
            self.append(cls.__new__(cls, *args, **kw))
    dispatch[pickle.NEWOBJ_EX[0]] = load_newobj_ex

    def load_build(self):
        state = self.stack.pop()
        obj = self.stack[-1]
        if _is_stub(obj):
            obj._state = state
        elif hasattr(obj, '__setstate__'):
            obj.__setstate__(state)
    dispatch[pickle.BUILD[0]] = load_build

    def load_obj(self):
        args = self.pop_mark()
        cls = args.pop(0)
        self.append(Stub(f'obj:{getattr(cls, "_name", cls)}', args))
    dispatch[pickle.OBJ[0]] = load_obj

    def load_inst(self):
        module = self.readline()[:-1].decode('ascii')
        name = self.readline()[:-1].decode('ascii')
        args = self.pop_mark()
        self.append(Stub(f'inst:{module}.{name}', args))
    dispatch[pickle.INST[0]] = load_inst


def load(path):
    with open(path, 'rb') as f:
        return InertUnpickler(f).load()


def to_array(x):
    """jax/numpy arrays surface as called stubs carrying the 5-tuple ndarray state."""
    if isinstance(x, np.ndarray):
        return x
    if isinstance(x, (int, float, complex, np.generic)):
        return np.asarray(x)
    _ver, shape, dt, fortran, raw = x._state
    assert not _is_stub(dt) and not dt.hasobject
    return np.frombuffer(raw, dtype=dt).reshape(shape, order='F' if fortran else 'C').copy()
