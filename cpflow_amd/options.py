"""Option records of the synthesis front end.

Field names, defaults, required-field behaviour (`TypeError`) and `get_static`
mirror the reference's dataclasses so option objects print and pickle the same:
    RegularizationOptions  cpflow/main.py:328-335
    BasicOptions           cpflow/main.py:338-367
    StaticOptions          cpflow/main.py:370-388
    AdaptiveOptions        cpflow/main.py:391-426
"""
import math
from dataclasses import asdict, dataclass, fields

_REQUIRED = -1   # sentinel the reference uses for "must be provided"


@dataclass
class RegularizationOptions:
    function: str = 'linear'
    ymax: float = 2
    xmax: float = math.pi / 2
    plato_0: float = 0.05
    plato_1: float = 0.05
    plato_2: float = 0.05


@dataclass
class BasicOptions:
    """Options shared by static and adaptive synthesis.

    num_samples: number of initial conditions in the multi-start optimisation.
    method: optimiser; the device path implements 'adam' (the reference's only tested one).
    learning_rate, num_gd_iterations: Adam step size / number of updates at the raw stage.
    cp_distribution: 'uniform' or '0' initialisation of the CP angles.
    entry_loss: loss below which a CP template counts as prospective.
    target_loss: loss below which a verified CZ circuit counts as a decomposition.
    threshold_cp: CP angles within this distance of 0 / pi are projected to identity / CZ.
    learning_rate_at_verification, num_gd_iterations_at_verification: verification-stage Adam.
    random_seed: seed of the initial angles (and of the adaptive seed chain).
    rotation_gates: letters of the 1q rotations inside each entangling block, e.g. 'xyz', 'xz'.
    """
    num_samples: int = 100
    method: str = 'adam'
    learning_rate: float = 0.1
    num_gd_iterations: int = 2000
    cp_distribution: str = 'uniform'
    entry_loss: float = 1e-3
    target_loss: float = 1e-6
    threshold_cp: float = 0.2
    learning_rate_at_verification: float = 0.01
    num_gd_iterations_at_verification: int = 5000
    random_seed: int = 0
    rotation_gates: str = 'xyz'


@dataclass
class StaticOptions(BasicOptions):
    """num_cp_gates: CP gates in the template; r: regularisation weight;
    accepted_num_cz_gates: verify prospective results whose CZ count is not above this."""
    num_cp_gates: int = _REQUIRED
    r: float = 0.00055
    accepted_num_cz_gates: int = _REQUIRED

    def __post_init__(self):
        for name in ('num_cp_gates', 'accepted_num_cz_gates'):
            if getattr(self, name) == _REQUIRED:
                raise TypeError(f"Missing required argument '{name}'")


@dataclass
class AdaptiveOptions(BasicOptions):
    """min/max_num_cp_gates: template-length search range; r_mean, r_variance: lognormal prior of r;
    max_evals: hyper-parameter configurations to evaluate; target_num_cz_gates /
    stop_if_target_reached: early stop; keep_logs: keep raw evaluations."""
    min_num_cp_gates: int = _REQUIRED
    max_num_cp_gates: int = _REQUIRED
    r_mean: float = 0.00055
    r_variance: float = 0.5
    max_evals: int = 100
    target_num_cz_gates: int = 0
    stop_if_target_reached: bool = False
    keep_logs: bool = False

    def __post_init__(self):
        for name in ('min_num_cp_gates', 'max_num_cp_gates'):
            if getattr(self, name) == _REQUIRED:
                raise TypeError(f"Missing required argument '{name}'")

    def get_static(self, num_cp_gates, r):
        """StaticOptions carrying this object's basic fields (accepted_num_cz_gates = None)."""
        basic_names = {f.name for f in fields(BasicOptions)}
        kwargs = {k: v for k, v in asdict(self).items() if k in basic_names}
        return StaticOptions(num_cp_gates=num_cp_gates, r=r, accepted_num_cz_gates=None, **kwargs)
