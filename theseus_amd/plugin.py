"""Adapters that plug the HIP back end into the REAL ``theseus`` (the drop-in boundary of SURVEY.md §8b).

    import theseus as th
    import theseus_amd.plugin as thp
    opt = th.LevenbergMarquardt(objective, linear_solver_cls=thp.HipCholeskySolver,
                                linearization_cls=thp.HipLinearization, vectorize=True)
    layer = th.TheseusLayer(opt)

The reference's optimiser loop (nonlinear_least_squares.py:100-215) stays untouched and talks to the two ABCs
``Linearization`` (theseus/optimizer/linearization.py:16-87) and ``LinearSolver``
(theseus/optimizer/linear/linear_solver.py:15-37); everything behind them runs in libtheseus_hip.so.
``theseus`` is imported lazily: this module is only usable where the reference is installed (it is not on the
GPU box of this build; tests/test_plugin_reference.py exercises it in the container that has /root/reference).
"""
from typing import Any, Dict, Optional, Type, Union

import torch

import theseus as th
from theseus.optimizer import Linearization as _RefLinearization
from theseus.optimizer.linear import CholeskyDenseSolver as _RefCholeskyDenseSolver
from theseus.optimizer.linear import LinearSolver as _RefLinearSolver

from .linear_solver import HipCholeskyCore
from .linearization import HipLinearizationCore


class HipLinearization(HipLinearizationCore, _RefLinearization):
    """Replaces ``th.DenseLinearization`` (theseus/optimizer/dense_linearization.py:15-77)."""

    def __init__(self, objective: th.Objective, ordering=None, kernels=None, **kwargs):
        _RefLinearization.__init__(self, objective, ordering)
        self._core_init(objective, kernels)

    def _linearize_jacobian_impl(self):
        self._materialize_A_b()

    def _linearize_hessian_impl(self, _detach_hessian: bool = False):
        # the Hessian is built by a kernel outside autograd, i.e. it is always "detached"
        # (dense_linearization.py:61 detaches it in the implicit step; UNROLL backward is not supported)
        self._assemble()

    def hessian_approx(self):
        return self._full_AtA()

    def _ata_impl(self) -> torch.Tensor:
        return self._full_AtA()

    def _atb_impl(self) -> torch.Tensor:
        return self.g.unsqueeze(2)


class HipCholeskySolver(HipCholeskyCore, _RefCholeskyDenseSolver):
    """Replaces ``th.CholeskyDenseSolver`` (theseus/optimizer/linear/dense_solver.py:20-161).

    It subclasses the reference's class so that ``LevenbergMarquardt``'s isinstance whitelists for ellipsoidal /
    adaptive damping accept it (levenberg_marquardt.py:21-48,82-87), but calls ``LinearSolver.__init__`` directly:
    ``DenseSolver.__init__`` rejects every linearization class that is not literally ``DenseLinearization``
    (dense_solver.py:28-32)."""

    def __init__(self, objective: th.Objective, linearization_cls: Optional[Type[_RefLinearization]] = None,
                 linearization_kwargs: Optional[Dict[str, Any]] = None, check_singular: bool = False, **kwargs):
        linearization_cls = linearization_cls or HipLinearization
        if not (isinstance(linearization_cls, type) and issubclass(linearization_cls, HipLinearization)):
            raise RuntimeError("HipCholeskySolver only works with theseus_amd.plugin.HipLinearization, "
                               f"but {linearization_cls} was provided.")
        _RefLinearSolver.__init__(self, objective, linearization_cls, linearization_kwargs)
        self._check_singular = check_singular
        self._core_init()

    def solve(self, damping: Optional[Union[float, torch.Tensor]] = None, ellipsoidal_damping: bool = True,
              damping_eps: float = 1e-8, **kwargs) -> torch.Tensor:
        # failure = RuntimeError, which the reference loop turns into FAIL status under no_grad
        # (nonlinear_least_squares.py:138-152)
        return self._solve(damping, ellipsoidal_damping, damping_eps, check_info=True)

    def _solve_sytem(self, Atb: torch.Tensor, AtA: torch.Tensor) -> torch.Tensor:  # abstract in DenseSolver
        raise NotImplementedError("HipCholeskySolver.solve() factorises its linearization's packed Hessian")
