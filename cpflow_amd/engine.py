"""Device engine: a CP-template problem compiled to a `cpf_plan` plus the calls that run it.

This is the thin host layer between the reference-shaped front end
(`cpflow_amd.main`) and the C ABI (`include/cpflow_amd.h`).  PyTorch is used for
device memory and streams only; every computation happens in the HIP library.

Reference counterparts:
    TemplateEngine.initial_angles  <- Synthesize._generate_initial_angles   cpflow/main.py:541-548
    TemplateEngine.optimize        <- mynimize_repeated + evaluate_cp_result cpflow/optimization.py:269-382,
                                                                            cpflow/cp_utils.py:144-164
    TemplateEngine.loss_and_grad   <- value_and_grad(regloss_func)           cpflow/optimization.py:331-340
    TemplateEngine.unitary         <- Ansatz.unitary                         cpflow/main.py:186-191
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _ffi


def placements_for(layer, num_blocks):
    """Block b sits on layer[b % len(layer)]: complete layers followed by the truncated one
    (fill_layers, cpflow/topology.py:15-20; Ansatz.all_placements, cpflow/main.py:175)."""
    layer = [[int(p[0]), int(p[1])] for p in layer]
    if num_blocks and not layer:
        raise ValueError('empty layer')
    return np.array([layer[b % len(layer)] for b in range(int(num_blocks))], dtype=np.int32).reshape(-1, 2)


def num_qubits_from_layer(layer):
    return max(int(q) for pair in layer for q in pair) + 1


def problem_struct(num_qubits, placements, rot, entangler, loss, target, penalty='linear', xmax=math.pi / 2, ymax=2.0,
                   plato_0=0.05, plato_1=0.05, plato_2=0.05, use_reg=True, r=0.00055, learning_rate=0.1, num_gd_iterations=2000,
                   threshold_cp=0.2, entry_loss=1e-3):
    """The `cpf_problem` of include/cpflow_amd.h from host arrays (the caller keeps placements / rot / target alive)."""
    pb = _ffi.Problem()
    pb.n_qubits, pb.num_blocks = int(num_qubits), len(placements)
    pb.placements = placements.ctypes.data
    pb.n_rot, pb.rot_axes = len(rot), rot
    pb.entangler, pb.loss_kind = _ffi.ENTANGLER[entangler], _ffi.LOSS[loss]
    pb.target = target.ctypes.data
    pb.penalty = _ffi.Penalty(_ffi.PENALTY[penalty], xmax, ymax, plato_0, plato_1, plato_2)
    pb.use_reg, pb.r = int(bool(use_reg)), float(r)
    pb.lr, pb.b1, pb.b2, pb.eps = float(learning_rate), 0.9, 0.999, 1e-8
    pb.num_iterations = int(num_gd_iterations)
    pb.threshold_cp, pb.entry_loss = float(threshold_cp), float(entry_loss)
    return pb


class TemplateEngine:
    def __init__(self, layer, num_cp_gates, target_unitary, rotation_gates='xyz', r=0.00055,
                 entangler='cp', use_reg=True, penalty='linear', xmax=math.pi / 2, ymax=2.0,
                 plato_0=0.05, plato_1=0.05, plato_2=0.05, learning_rate=0.1, num_gd_iterations=2000,
                 threshold_cp=0.2, entry_loss=1e-3, num_qubits=None, device=None, placements=None, loss='hst'):
        if not torch.cuda.is_available():
            raise _ffi.CpfError('cpflow_amd needs a ROCm GPU: torch.cuda.is_available() is False '
                                '(there is no CPU fallback).')
        self._lib = _ffi.lib()
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else int(device))
        self.layer = [[int(p[0]), int(p[1])] for p in layer]
        self.num_qubits = int(num_qubits) if num_qubits is not None else num_qubits_from_layer(self.layer)
        self.num_cp_gates = int(num_cp_gates)
        self.rotation_gates = str(rotation_gates)
        self.entangler = entangler
        N = 1 << self.num_qubits
        target = np.asarray(target_unitary, dtype=np.complex128)
        if loss == 'state' and target.ndim == 1:         # target state -> column 0 of an otherwise zero matrix
            assert target.shape == (N,), 'Number of qubits in target state and layer do not match.'
            target = np.concatenate([target[:, None], np.zeros((N, N - 1))], axis=1)
        target = np.ascontiguousarray(target)
        assert target.shape == (N, N), 'Number of qubits in target unitary and layer do not match.'
        if loss not in _ffi.LOSS:
            raise ValueError(f"loss '{loss}' not in {sorted(_ffi.LOSS)}")
        self._target = target
        if placements is not None:      # explicit block placements (Ansatz); otherwise the layer is cycled
            self._placements = np.ascontiguousarray(np.asarray(placements, dtype=np.int32).reshape(-1, 2))
            assert len(self._placements) == self.num_cp_gates
        else:
            self._placements = placements_for(self.layer, self.num_cp_gates)
        self._rot = self.rotation_gates.encode()
        if entangler not in _ffi.ENTANGLER:
            raise TypeError(f"Gate '{entangler}' not implemented.")
        pb = problem_struct(self.num_qubits, self._placements, self._rot, entangler, loss, self._target, penalty,
                            xmax, ymax, plato_0, plato_1, plato_2, use_reg, r, learning_rate, num_gd_iterations, threshold_cp, entry_loss)
        self._problem = pb
        self.num_gd_iterations = int(num_gd_iterations)
        handle = C.c_void_p()
        _ffi.check(self._lib.cpf_plan_create(C.byref(pb), self.device.index, C.byref(handle)))
        self._plan = handle
        self.num_angles = int(self._lib.cpf_plan_num_params(self._plan))
        self.num_gates = int(self._lib.cpf_plan_num_gates(self._plan))
        mask = np.zeros(self.num_angles, dtype=np.uint8)
        _ffi.check(self._lib.cpf_plan_cp_mask(self._plan, mask.ctypes.data))
        self.cp_mask = mask
        self.flops_per_sample_step = float(self._lib.cpf_plan_flops_per_sample_step(self._plan))
        self.event_log = None      # set to [] to collect (start, end) stream events around every optimize launch

    def close(self):
        if getattr(self, '_plan', None):
            self._lib.cpf_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _empty(self, shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _as_device(self, x, dtype=torch.float32):
        t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
        return t.to(device=self.device, dtype=dtype).contiguous()

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    # ------------------------------------------------------------------ calls
    def initial_angles(self, seed, num_samples_total, first=0, count=None, cp_dist='uniform'):
        """float32[count, P] on the device; row j is what jax draws for global sample first+j."""
        modes = {'uniform': 0, '0': 1, 'normal': 2}
        if cp_dist not in modes:           # the reference prints "cp_dist ... not supported" and returns None (cp_utils.py:41-42)
            raise ValueError(f"cp_dist '{cp_dist}' not supported")
        count = int(num_samples_total) - int(first) if count is None else int(count)
        out = self._empty((count, self.num_angles))
        _ffi.check(self._lib.cpf_initial_angles_device(
            self._plan, int(seed) & 0xFFFFFFFF, int(num_samples_total), int(first), count,
            modes[cp_dist], self._ptr(out), self._stream()))
        return out

    def optimize(self, init, frozen=None, keep_history=False):
        """Multi-start Adam + per-sample evaluation.  Returns device tensors:
        params [S,2,P], regloss/loss/reg [S,2], cz [S] int32, passed [S] bool (+ history)."""
        init = self._as_device(init)
        if init.dim() == 1:
            init = init[None]
        S, P, T = init.shape[0], self.num_angles, self.num_gd_iterations
        assert init.shape[1] == P, f'initial angles have {init.shape[1]} columns, ansatz has {P} angles'
        fz = None
        if frozen is not None:
            fz = self._as_device(frozen, torch.uint8)
            assert fz.shape == (S, P)
        out = dict(params=self._empty((S, 2, P)), regloss=self._empty((S, 2)), loss=self._empty((S, 2)),
                   reg=self._empty((S, 2)), cz=self._empty((S,), torch.int32), passed=self._empty((S,), torch.uint8))
        hp = self._empty((S, T, P)) if keep_history else None
        hl = self._empty((S, T)) if keep_history else None
        if self.event_log is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream(self.device))
        _ffi.check(self._lib.cpf_optimize_device(
            self._plan, S, self._ptr(init), self._ptr(fz), self._ptr(out['params']), self._ptr(out['regloss']),
            self._ptr(out['loss']), self._ptr(out['reg']), self._ptr(out['cz']), self._ptr(out['passed']),
            self._ptr(hp), self._ptr(hl), self._stream()))
        if self.event_log is not None:
            ev[1].record(torch.cuda.current_stream(self.device))
            self.event_log.append(ev)
        out['passed'] = out['passed'].bool()
        if keep_history:
            out['history_params'], out['history_regloss'] = hp, hl
            out['history_reg'] = self.reg(hp)
        self._keepalive = (init, fz)
        return out

    def mean_kernel_ms(self):
        """Average duration of the optimize launches logged in event_log (synchronises)."""
        torch.cuda.synchronize(self.device)
        times = [a.elapsed_time(b) for a, b in self.event_log]
        return sum(times) / len(times), times

    def last_variant(self):
        """Name of the kernel variant the last launch used (fast3/aot, layered/jit, generic/aot)."""
        buf = C.create_string_buffer(512)
        _ffi.check(self._lib.cpf_plan_last_variant(self._plan, buf, 512))
        return buf.value.decode()

    def last_kernel_ms(self):
        ms = C.c_float(0)
        _ffi.check(self._lib.cpf_plan_last_kernel_ms(self._plan, C.byref(ms)))
        return float(ms.value)

    def loss_and_grad(self, theta, with_grad=True):
        theta = self._as_device(theta)
        if theta.dim() == 1:
            theta = theta[None]
        S, P = theta.shape
        assert P == self.num_angles
        loss, reg, regloss = self._empty((S,)), self._empty((S,)), self._empty((S,))
        grad = self._empty((S, P)) if with_grad else None
        _ffi.check(self._lib.cpf_loss_grad_device(self._plan, S, self._ptr(theta), self._ptr(loss), self._ptr(reg),
                                                  self._ptr(regloss), self._ptr(grad), self._stream()))
        self._keepalive = (theta,)
        return dict(loss=loss, reg=reg, regloss=regloss, grad=grad)

    def reg(self, theta):
        """r * sum of CP penalties for angle rows [..., P] -> float32 [...] (device tensor)."""
        theta = self._as_device(theta)
        shape = theta.shape[:-1]
        flat = theta.reshape(-1, self.num_angles).contiguous()
        out = self._empty((flat.shape[0],))
        _ffi.check(self._lib.cpf_reg_device(self._plan, flat.shape[0], self._ptr(flat), self._ptr(out), self._stream()))
        self._keepalive = (flat,)
        return out.reshape(shape)

    def unitary(self, theta):
        """complex64 [S, N, N] (or [N, N] for a single angle vector)."""
        theta = self._as_device(theta)
        single = theta.dim() == 1
        if single:
            theta = theta[None]
        S, P = theta.shape
        assert P == self.num_angles
        N = 1 << self.num_qubits
        out = self._empty((S, N, N, 2))
        _ffi.check(self._lib.cpf_unitary_device(self._plan, S, self._ptr(theta), self._ptr(out), self._stream()))
        self._keepalive = (theta,)
        u = torch.view_as_complex(out)
        return u[0] if single else u
