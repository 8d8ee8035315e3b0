"""Host-side mirror of the part of the Theseus modelling API that feeds the GN/LM hot path.

Same names, argument meaning and error behaviour as the reference so that code written against
``theseus`` reads the same against ``theseus_amd`` for this path:
  Variable              theseus/core/variable.py:14-112
  SE3 / SE2             theseus/geometry/se3.py:20-300, se2.py:20-340, theseus/geometry/lie_group.py:19-260
  Scale/DiagonalCostWeight  theseus/core/cost_weight.py:60-139
  Between / Difference  theseus/embodied/measurements/between.py:16-60, theseus/embodied/misc/local_cost_fn.py:16-75
  Objective             theseus/core/objective.py:42-956 (add / update / error / error_metric / retract)
All arithmetic is done by the HIP kernels (theseus_amd/csrc); nothing here computes on the CPU.
"""
import abc
from collections import OrderedDict
from itertools import count
from typing import Dict, List, Optional, Sequence, Union

import torch

from .kernels import default_kernels


class Variable:
    """Named tensor with a leading batch dimension (theseus/core/variable.py:14-112)."""

    _ids = count(0)
    _global_updates = 0  # bumped by every update()/to(): lets packed buffers skip per-variable stamp scans

    def __init__(self, tensor: torch.Tensor, name: Optional[str] = None):
        self._id = next(Variable._ids)
        self.name = name if name else f"{self.__class__.__name__}__{self._id}"
        self._tensor = tensor
        self._num_updates = 0

    @property
    def tensor(self) -> torch.Tensor:
        return self._tensor

    @tensor.setter
    def tensor(self, value: torch.Tensor):
        # a plain attribute in the reference (core/variable.py:21): assigning it is allowed, and may change the batch size or the
        # device -- counted like update(), so Objective.update()'s "nothing changed" fast path re-resolves them.  The package's own
        # re-pointing of variables at views of its packed buffers (same shape, same device) writes ``_tensor`` directly.
        self._tensor = value
        Variable._global_updates += 1

    def update(self, data: Union[torch.Tensor, "Variable"], batch_ignore_mask: Optional[torch.Tensor] = None):
        if isinstance(data, Variable):
            data = data.tensor
        if data.ndim != self.tensor.ndim or data.shape[1:] != self.tensor.shape[1:]:
            raise ValueError(
                f"Tried to update tensor {self.name} with data incompatible with original tensor shape. "
                f"Given {tuple(data.shape[1:])}. Expected: {tuple(self.tensor.shape[1:])}")
        if data.dtype != self.dtype:
            raise ValueError(f"Tried to update used tensor of dtype {data.dtype} but Variable "
                             f"{self.name} has dtype {self.dtype}.")
        if batch_ignore_mask is not None and batch_ignore_mask.any():
            mask = batch_ignore_mask.view([-1] + [1] * (data.ndim - 1))
            self._tensor = torch.where(mask, self._tensor, data)  # core/variable.py:65-69
        else:
            self._tensor = data
        self._num_updates += 1
        Variable._global_updates += 1

    @property
    def shape(self):
        return self.tensor.shape

    @property
    def dtype(self):
        return self.tensor.dtype

    @property
    def device(self):
        return self.tensor.device

    @property
    def ndim(self):
        return self.tensor.ndim

    def to(self, *args, **kwargs):
        self._tensor = self._tensor.to(*args, **kwargs)
        self._num_updates += 1
        Variable._global_updates += 1

    def copy(self, new_name: Optional[str] = None) -> "Variable":
        return self.__class__(tensor=self.tensor.clone(), name=new_name or f"{self.name}_copy")

    def __getitem__(self, item):
        return self.tensor[item]


class Vector(Variable):
    """Euclidean variable (theseus/geometry/vector.py): auxiliary data of the fused objectives, optimisation variable of the
    generic path (theseus_amd/euclidean.py)."""

    def __init__(self, dof: Optional[int] = None, tensor: Optional[torch.Tensor] = None,
                 name: Optional[str] = None, dtype: Optional[torch.dtype] = None):
        if tensor is None:
            tensor = torch.zeros(1, dof, dtype=dtype or torch.get_default_dtype())
        if tensor.ndim == 1:
            tensor = tensor.view(1, -1)
        super().__init__(tensor, name)

    def dof(self) -> int:
        return self.tensor.shape[1]

    # theseus/geometry/vector.py:150-178: local(a, b) = b - a, retraction = addition
    def local(self, other: "Vector") -> torch.Tensor:
        return other.tensor - self.tensor

    def retract(self, delta: torch.Tensor) -> "Vector":
        return Vector(tensor=self.tensor + delta)

    def copy(self, new_name: Optional[str] = None) -> "Vector":
        return Vector(tensor=self.tensor.clone(), name=new_name or f"{self.name}_copy")


class Point3(Vector):
    """World point (theseus/geometry/point_types.py:97-170): a 3-vector; retraction = addition
    (theseus/geometry/vector.py:177-178), local(a, b) = b - a (:150-160)."""

    def __init__(self, tensor: Optional[torch.Tensor] = None, name: Optional[str] = None,
                 dtype: Optional[torch.dtype] = None):
        if tensor is not None and tensor.shape[-1] != 3:
            raise ValueError("Provided tensor must have shape (batch_size, 3).")
        super().__init__(3, tensor=tensor, name=name, dtype=dtype)

    def local(self, other: "Vector") -> torch.Tensor:
        return other.tensor - self.tensor

    def retract(self, delta: torch.Tensor) -> "Point3":
        return Point3(tensor=self.tensor + delta)

    def copy(self, new_name: Optional[str] = None) -> "Point3":
        return Point3(tensor=self.tensor.clone(), name=new_name or f"{self.name}_copy")


class Point2(Vector):
    """Image point (theseus/geometry/point_types.py:18-94)."""

    def __init__(self, tensor: Optional[torch.Tensor] = None, name: Optional[str] = None,
                 dtype: Optional[torch.dtype] = None):
        if tensor is not None and tensor.shape[-1] != 2:
            raise ValueError("Provided tensor must have shape (batch_size, 2).")
        super().__init__(2, tensor=tensor, name=name, dtype=dtype)

    def copy(self, new_name: Optional[str] = None) -> "Point2":
        return Point2(tensor=self.tensor.clone(), name=new_name or f"{self.name}_copy")


class SE3(Variable):
    """SE3 group element batch, tensor (B,3,4) = [R | t]; tangent [v, w]; right perturbations."""

    def __init__(self, x_y_z_quaternion: Optional[torch.Tensor] = None, tensor: Optional[torch.Tensor] = None,
                 name: Optional[str] = None, dtype: Optional[torch.dtype] = None):
        if x_y_z_quaternion is not None and tensor is not None:
            raise ValueError("Please provide only one of x_y_z_quaternion or tensor.")
        if x_y_z_quaternion is not None:
            tensor = SE3._from_x_y_z_quaternion(x_y_z_quaternion)
        if tensor is None:
            tensor = torch.eye(3, 4, dtype=dtype or torch.get_default_dtype()).view(1, 3, 4)
        if tensor.ndim == 2:
            tensor = tensor.unsqueeze(0)
        if tensor.ndim != 3 or tensor.shape[1:] != (3, 4):
            raise ValueError("SE3 data tensors can only be 3x4 matrices.")
        if dtype is not None and tensor.dtype != dtype:
            tensor = tensor.to(dtype)
        super().__init__(tensor, name)

    @staticmethod
    def _from_x_y_z_quaternion(q: torch.Tensor) -> torch.Tensor:
        # theseus/geometry/se3.py:128-145 ; quaternion order (w, x, y, z) after the translation
        if q.ndim == 1:
            q = q.unsqueeze(0)
        t, w, x, y, z = q[:, :3], q[:, 3], q[:, 4], q[:, 5], q[:, 6]
        R = torch.stack([
            1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1).view(-1, 3, 3)
        return torch.cat([R, t.unsqueeze(2)], dim=2)

    @staticmethod
    def dof() -> int:
        return 6

    # ---- group operations: all evaluated by the HIP elementwise kernels -------------------------
    @staticmethod
    def exp_map(tangent_vector: torch.Tensor, jacobians: Optional[List[torch.Tensor]] = None) -> "SE3":
        if tangent_vector.ndim != 2 or tangent_vector.shape[1] != 6:
            raise ValueError("Tangent vectors of SE3 should be 6-D vectors.")
        K = default_kernels()
        if jacobians is not None:
            X, J = K.se3_exp(tangent_vector, jac=True)
            jacobians.append(J)
        else:
            X = K.se3_exp(tangent_vector)
        return SE3(tensor=X)

    def log_map(self, jacobians: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        K = default_kernels()
        if jacobians is not None:
            xi, J = K.se3_log(self.tensor, jac=True)
            jacobians.append(J)
            return xi
        return K.se3_log(self.tensor)

    def adjoint(self) -> torch.Tensor:
        return default_kernels().se3_adjoint(self.tensor)

    def inverse(self) -> "SE3":
        return SE3(tensor=default_kernels().se3_inverse(self.tensor))

    def compose(self, other: "SE3") -> "SE3":
        a, b = _broadcast_pair(self.tensor, other.tensor)
        return SE3(tensor=default_kernels().se3_compose(a, b))

    def between(self, other: "SE3") -> "SE3":
        return self.inverse().compose(other)

    def local(self, other: "SE3") -> torch.Tensor:
        return self.between(other).log_map()

    def retract(self, delta: torch.Tensor) -> "SE3":
        return self.compose(SE3.exp_map(delta))

    @staticmethod
    def rand(*size: int, generator=None, dtype=None, device=None) -> "SE3":
        # theseus/geometry/se3.py:45-62 (uniform quaternion + uniform translation); here: exp of a
        # random tangent, adequate for synthetic data
        if len(size) != 1:
            raise ValueError("The size should be 1D.")
        xi = torch.rand(size[0], 6, generator=generator, dtype=dtype, device=device) * 2 - 1
        xi[:, 3:] *= 3.0
        return SE3.exp_map(xi)

    def copy(self, new_name: Optional[str] = None) -> "SE3":
        return SE3(tensor=self.tensor.clone(), name=new_name or f"{self.name}_copy")


class SE2(Variable):
    """SE2 group element batch, tensor (B,4) = [x, y, cos, sin]; tangent [u_x, u_y, theta]; right perturbations
    (theseus/geometry/se2.py:20-120)."""

    def __init__(self, x_y_theta: Optional[torch.Tensor] = None, tensor: Optional[torch.Tensor] = None,
                 name: Optional[str] = None, dtype: Optional[torch.dtype] = None):
        if x_y_theta is not None and tensor is not None:
            raise ValueError("Please provide only one of x_y_theta or tensor.")
        if x_y_theta is not None:
            if x_y_theta.ndim == 1:
                x_y_theta = x_y_theta.unsqueeze(0)
            th = x_y_theta[:, 2:]
            tensor = torch.cat([x_y_theta[:, :2], th.cos(), th.sin()], dim=1)  # se2.py:40-44,93-106
        if tensor is None:
            tensor = torch.tensor([[0.0, 0.0, 1.0, 0.0]], dtype=dtype or torch.get_default_dtype())
        if tensor.ndim == 1:
            tensor = tensor.unsqueeze(0)
        if tensor.ndim != 2 or tensor.shape[1] != 4:
            raise ValueError("SE2 data tensors can only be 4D vectors.")
        if dtype is not None and tensor.dtype != dtype:
            tensor = tensor.to(dtype)
        super().__init__(tensor, name)

    @staticmethod
    def dof() -> int:
        return 3

    @staticmethod
    def exp_map(tangent_vector: torch.Tensor, jacobians: Optional[List[torch.Tensor]] = None) -> "SE2":
        if tangent_vector.ndim != 2 or tangent_vector.shape[1] != 3:
            raise ValueError("Tangent vectors of SE2 should be 3-D vectors.")
        K = default_kernels()
        if jacobians is not None:
            X, J = K.se2_exp(tangent_vector, jac=True)
            jacobians.append(J)
        else:
            X = K.se2_exp(tangent_vector)
        return SE2(tensor=X)

    def log_map(self, jacobians: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        K = default_kernels()
        if jacobians is not None:
            xi, J = K.se2_log(self.tensor, jac=True)
            jacobians.append(J)
            return xi
        return K.se2_log(self.tensor)

    def adjoint(self) -> torch.Tensor:
        return default_kernels().se2_adjoint(self.tensor)

    def inverse(self) -> "SE2":
        return SE2(tensor=default_kernels().se2_inverse(self.tensor))

    def compose(self, other: "SE2") -> "SE2":
        a, b = _broadcast_pair(self.tensor, other.tensor)
        return SE2(tensor=default_kernels().se2_compose(a, b))

    def between(self, other: "SE2") -> "SE2":
        return self.inverse().compose(other)

    def local(self, other: "SE2") -> torch.Tensor:
        return self.between(other).log_map()

    def retract(self, delta: torch.Tensor) -> "SE2":
        return self.compose(SE2.exp_map(delta))

    def theta(self) -> torch.Tensor:
        return torch.atan2(self.tensor[:, 3], self.tensor[:, 2])

    def copy(self, new_name: Optional[str] = None) -> "SE2":
        return SE2(tensor=self.tensor.clone(), name=new_name or f"{self.name}_copy")


class SO3(Variable):
    """SO3 group element batch, tensor (B,3,3); tangent w (3); right perturbations (theseus/geometry/so3.py:20-186 over
    torchlie/torchlie/functional/so3_impl.py)."""

    def __init__(self, quaternion: Optional[torch.Tensor] = None, tensor: Optional[torch.Tensor] = None,
                 name: Optional[str] = None, dtype: Optional[torch.dtype] = None):
        if quaternion is not None and tensor is not None:
            raise ValueError("Please provide only one of quaternion or tensor.")
        if quaternion is not None:
            if quaternion.ndim == 1:
                quaternion = quaternion.unsqueeze(0)
            tensor = SE3._from_x_y_z_quaternion(torch.cat([quaternion.new_zeros(quaternion.shape[0], 3), quaternion], 1))[:, :, :3]
        if tensor is None:
            tensor = torch.eye(3, dtype=dtype or torch.get_default_dtype()).unsqueeze(0)
        if tensor.ndim == 2:
            tensor = tensor.unsqueeze(0)
        if tensor.ndim != 3 or tensor.shape[1:] != (3, 3):
            raise ValueError("SO3 data tensors can only be 3x3 matrices.")
        if dtype is not None and tensor.dtype != dtype:
            tensor = tensor.to(dtype)
        super().__init__(tensor, name)

    @staticmethod
    def dof() -> int:
        return 3

    @staticmethod
    def exp_map(tangent_vector: torch.Tensor, jacobians: Optional[List[torch.Tensor]] = None) -> "SO3":
        if tangent_vector.ndim != 2 or tangent_vector.shape[1] != 3:
            raise ValueError("Tangent vectors of SO3 should be 3-D vectors.")
        K = default_kernels()
        if jacobians is not None:
            X, J = K.so3_exp(tangent_vector, jac=True)
            jacobians.append(J)
        else:
            X = K.so3_exp(tangent_vector)
        return SO3(tensor=X)

    def log_map(self, jacobians: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        K = default_kernels()
        if jacobians is not None:
            w, J = K.so3_log(self.tensor, jac=True)
            jacobians.append(J)
            return w
        return K.so3_log(self.tensor)

    def adjoint(self) -> torch.Tensor:
        return self.tensor.clone()   # so3.py:99-100

    def inverse(self) -> "SO3":
        return SO3(tensor=default_kernels().so3_inverse(self.tensor))

    def compose(self, other: "SO3") -> "SO3":
        a, b = _broadcast_pair(self.tensor, other.tensor)
        return SO3(tensor=default_kernels().so3_compose(a, b))

    def between(self, other: "SO3") -> "SO3":
        return self.inverse().compose(other)

    def local(self, other: "SO3") -> torch.Tensor:
        return self.between(other).log_map()

    def retract(self, delta: torch.Tensor) -> "SO3":
        return self.compose(SO3.exp_map(delta))

    def copy(self, new_name: Optional[str] = None) -> "SO3":
        return SO3(tensor=self.tensor.clone(), name=new_name or f"{self.name}_copy")


class SO2(Variable):
    """SO2 group element batch, tensor (B,2) = [cos, sin]; tangent theta (1); right perturbations
    (theseus/geometry/so2.py:20-120; every Jacobian of the group is the scalar 1: :116-117,167-223)."""

    def __init__(self, theta: Optional[torch.Tensor] = None, tensor: Optional[torch.Tensor] = None,
                 name: Optional[str] = None, dtype: Optional[torch.dtype] = None):
        if theta is not None and tensor is not None:
            raise ValueError("Please provide only one of theta or tensor.")
        if theta is not None:
            if theta.ndim == 1:
                theta = theta.unsqueeze(1)
            if theta.ndim != 2 or theta.shape[1] != 1:
                raise ValueError("Argument theta must be have ndim = 1, or ndim=2 and shape[1] = 1.")
            tensor = torch.cat([theta.cos(), theta.sin()], dim=1)   # so2.py:96-100
        if tensor is None:
            tensor = torch.tensor([[1.0, 0.0]], dtype=dtype or torch.get_default_dtype())
        if tensor.ndim == 1:
            tensor = tensor.unsqueeze(0)
        if tensor.ndim != 2 or tensor.shape[1] != 2:
            raise ValueError("SO2 data tensors can only be 2D vectors.")
        if dtype is not None and tensor.dtype != dtype:
            tensor = tensor.to(dtype)
        super().__init__(tensor, name)

    @staticmethod
    def dof() -> int:
        return 1

    @staticmethod
    def exp_map(tangent_vector: torch.Tensor, jacobians: Optional[List[torch.Tensor]] = None) -> "SO2":
        if tangent_vector.ndim != 2 or tangent_vector.shape[1] != 1:
            raise ValueError("Tangent vectors of SO2 should be 1-D vectors.")
        K = default_kernels()
        if jacobians is not None:
            X, J = K.so2_exp(tangent_vector, jac=True)
            jacobians.append(J)
        else:
            X = K.so2_exp(tangent_vector)
        return SO2(tensor=X)

    def log_map(self, jacobians: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        K = default_kernels()
        if jacobians is not None:
            th, J = K.so2_log(self.tensor, jac=True)
            jacobians.append(J)
            return th
        return K.so2_log(self.tensor)

    def adjoint(self) -> torch.Tensor:
        return default_kernels().so2_adjoint(self.tensor)

    def inverse(self) -> "SO2":
        return SO2(tensor=default_kernels().so2_inverse(self.tensor))

    def compose(self, other: "SO2") -> "SO2":
        a, b = _broadcast_pair(self.tensor, other.tensor)
        return SO2(tensor=default_kernels().so2_compose(a, b))

    def between(self, other: "SO2") -> "SO2":
        return self.inverse().compose(other)

    def local(self, other: "SO2") -> torch.Tensor:
        return self.between(other).log_map()

    def retract(self, delta: torch.Tensor) -> "SO2":
        return self.compose(SO2.exp_map(delta))

    def theta(self) -> torch.Tensor:
        return self.log_map()   # so2.py:113-114

    def copy(self, new_name: Optional[str] = None) -> "SO2":
        return SO2(tensor=self.tensor.clone(), name=new_name or f"{self.name}_copy")


def _broadcast_pair(a, b):
    if a.shape[0] != b.shape[0]:
        if a.shape[0] == 1:
            a = a.expand_as(b)
        elif b.shape[0] == 1:
            b = b.expand_as(a)
        else:
            raise ValueError("Batch sizes must match or be 1.")
    return a.contiguous(), b.contiguous()


# ------------------------------------------------------------------------------------------------
# cost weights
# ------------------------------------------------------------------------------------------------
class CostWeight(abc.ABC):
    def __init__(self, name: Optional[str] = None):
        self.name = name

    @abc.abstractmethod
    def sqrt_diag(self, dim: int) -> torch.Tensor:
        """(Bw, dim) sqrt-information diagonal (what the kernels consume)."""

    def diagonal6(self) -> torch.Tensor:
        return self.sqrt_diag(6)

    def to(self, *args, **kwargs):  # theseus/core/theseus_function.py:74-77
        for v in self.aux_vars():
            v.to(*args, **kwargs)


class ScaleCostWeight(CostWeight):
    """e <- e * s, J <- J * s (theseus/core/cost_weight.py:60-90)."""

    def __init__(self, scale: Union[float, torch.Tensor, Variable], name: Optional[str] = None):
        super().__init__(name)
        if not isinstance(scale, Variable):
            if not isinstance(scale, torch.Tensor):
                scale = torch.tensor(float(scale))
            scale = Variable(scale.view(-1, 1) if scale.ndim < 2 else scale)
        if scale.tensor.ndim != 2 or scale.tensor.shape[1] != 1:
            raise ValueError("ScaleCostWeight only accepts 0-dim or 1-dim tensors, or (batch, 1) tensors.")
        self.scale = scale

    def aux_vars(self):
        return [self.scale]

    def sqrt_diag(self, dim):
        return self.scale.tensor.expand(-1, dim)


class DiagonalCostWeight(CostWeight):
    """e <- e * w, J <- diag(w) J (theseus/core/cost_weight.py:93-139)."""

    def __init__(self, diagonal: Union[Sequence[float], torch.Tensor, Variable], name: Optional[str] = None):
        super().__init__(name)
        if not isinstance(diagonal, Variable):
            if not isinstance(diagonal, torch.Tensor):
                diagonal = torch.tensor(diagonal)
            diagonal = Variable(diagonal.view(1, -1) if diagonal.ndim == 1 else diagonal)
        if diagonal.tensor.ndim != 2:
            raise ValueError("DiagonalCostWeight only accepts tensors of shape (batch, dim) or (dim,).")
        self.diagonal = diagonal

    def aux_vars(self):
        return [self.diagonal]

    def sqrt_diag(self, dim):
        if self.diagonal.tensor.shape[1] != dim:
            raise ValueError(f"This cost needs a {dim}-dimensional DiagonalCostWeight.")
        return self.diagonal.tensor


# ------------------------------------------------------------------------------------------------
# cost functions
# ------------------------------------------------------------------------------------------------
class CostFunction(abc.ABC):
    _ids = count(0)

    def __init__(self, cost_weight: CostWeight, name: Optional[str] = None):
        self.weight = cost_weight
        self.name = name if name else f"{self.__class__.__name__}__{next(CostFunction._ids)}"

    @abc.abstractmethod
    def optim_vars(self) -> List[Variable]:
        pass

    @abc.abstractmethod
    def aux_vars(self) -> List[Variable]:
        pass

    @abc.abstractmethod
    def error(self) -> torch.Tensor:
        pass

    @abc.abstractmethod
    def dim(self) -> int:
        pass

    def weighted_error(self) -> torch.Tensor:
        return self.error() * self.weight.sqrt_diag(self.dim())

    def weighted_jacobians_error(self):
        """(theseus/core/cost_function.py:107-122) for cost functions that implement ``jacobians()`` -- the fused ones never
        evaluate their Jacobians in torch."""
        jacobians, err = self.jacobians()
        w = self.weight.sqrt_diag(self.dim())
        return [j * w.unsqueeze(2) for j in jacobians], err * w


class AutoDiffCostFunction(CostFunction):
    """User error function, Jacobians by automatic differentiation (theseus/core/cost_function.py:203-393, the default
    ``AutogradMode.VMAP``: ``vmap(jacrev(err_fn))`` over the batch, :298-341).  ``err_fn(optim_vars, aux_vars)`` -> (B, dim)
    receives tuples of variables and reads their ``.tensor``.  Optimisation variables must be Euclidean here (the Jacobian
    w.r.t. the tensor IS the Jacobian w.r.t. the tangent: vector.py ``project`` is the identity)."""

    def __init__(self, optim_vars: Sequence[Variable], err_fn, dim: int, cost_weight: Optional[CostWeight] = None,
                 aux_vars: Optional[Sequence[Variable]] = None, name: Optional[str] = None, **autograd_kwargs):
        super().__init__(cost_weight if cost_weight is not None else ScaleCostWeight(1.0), name)
        if len(optim_vars) < 1:
            raise ValueError("AutodiffCostFunction must receive at least one optimization variable.")
        self._optim_vars, self._aux_vars = list(optim_vars), list(aux_vars or [])
        self._err_fn, self._dim = err_fn, int(dim)

    def optim_vars(self):
        return list(self._optim_vars)

    def aux_vars(self):
        return list(self._aux_vars)

    def dim(self) -> int:
        return self._dim

    def error(self) -> torch.Tensor:
        err = self._err_fn(optim_vars=tuple(self._optim_vars), aux_vars=tuple(self._aux_vars))
        if err.shape[1] != self._dim:
            raise ValueError("Output dimension of given error function doesn't match self.dim().")
        return err

    def jacobians(self):
        import copy
        from torch.func import jacrev, vmap
        err = self.error()
        for v in self._optim_vars:
            if "Vector" not in {c.__name__ for c in type(v).__mro__}:
                raise NotImplementedError("AutoDiffCostFunction: optimisation variables must be Euclidean (Vector / Point2 / "
                                          f"Point3) on this back end; got {type(v).__name__} ({v.name}).")
        B = max(t.shape[0] for t in (v.tensor for v in self._optim_vars + self._aux_vars))
        full = lambda t: t if t.shape[0] == B else t.expand(B, *t.shape[1:])  # noqa: E731
        opt_t = tuple(full(v.tensor) for v in self._optim_vars)
        aux_t = tuple(full(v.tensor) for v in self._aux_vars)
        # shallow copies hold the per-sample tensors inside the transform (the reference's _tmp_optim_vars / _tmp_aux_vars)
        tmp_opt = tuple(copy.copy(v) for v in self._optim_vars)
        tmp_aux = tuple(copy.copy(v) for v in self._aux_vars)

        def one(opt_tensors, aux_tensors):
            for h, t in zip(tmp_opt, opt_tensors):
                h._tensor = t.unsqueeze(0)
            for h, t in zip(tmp_aux, aux_tensors):
                h._tensor = t.unsqueeze(0)
            return self._err_fn(optim_vars=tmp_opt, aux_vars=tmp_aux)[0]

        jac = vmap(jacrev(one, argnums=0))(opt_t, aux_t)
        return [j.reshape(B, self._dim, -1) for j in jac], err


class Between(CostFunction):
    """e = log(meas^-1 (v0^-1 v1)) (theseus/embodied/measurements/between.py:16-60)."""

    def __init__(self, v0: SE3, v1: SE3, measurement: SE3, cost_weight: CostWeight, name: Optional[str] = None):
        super().__init__(cost_weight, name)
        if not isinstance(v0, v1.__class__) or not isinstance(v0, measurement.__class__):
            raise ValueError("Inconsistent types between variables and measurement.")
        self.v0, self.v1, self.measurement = v0, v1, measurement

    def optim_vars(self):
        return [self.v0, self.v1]

    def aux_vars(self):
        return [self.measurement] + self.weight.aux_vars()

    def error(self):
        return self.measurement.local(self.v0.between(self.v1))

    def dim(self):
        return self.v0.dof()


class Difference(CostFunction):
    """e = log(target^-1 var) (theseus/embodied/misc/local_cost_fn.py:16-75; ``Local`` is an alias)."""

    def __init__(self, var: SE3, target: SE3, cost_weight: CostWeight, name: Optional[str] = None):
        super().__init__(cost_weight, name)
        if not isinstance(var, target.__class__):
            raise ValueError("Variable for the Local inconsistent with the given target.")
        self.var, self.target = var, target

    def optim_vars(self):
        return [self.var]

    def aux_vars(self):
        return [self.target] + self.weight.aux_vars()

    def error(self):
        return self.target.local(self.var)

    def dim(self):
        return self.var.dof()


Local = Difference


class Reprojection(CostFunction):
    """Pinhole reprojection with two radial terms (theseus/embodied/measurements/reprojection.py:13-94): optimises
    ``camera_pose`` (SE3) and ``world_point`` (Point3); auxiliary ``image_feature_point``, ``focal_length``,
    ``calib_k1``, ``calib_k2``.  Evaluated by the fused bundle-adjustment kernels (csrc/ba_kernels.hip)."""

    def __init__(self, camera_pose: SE3, world_point: Point3, image_feature_point: Point2, focal_length: Vector,
                 calib_k1: Optional[Vector] = None, calib_k2: Optional[Vector] = None, weight: Optional[CostWeight] = None,
                 name: Optional[str] = None):
        dt, dev = camera_pose.dtype, camera_pose.device
        if weight is None:
            weight = ScaleCostWeight(torch.tensor(1.0, dtype=dt, device=dev))
        super().__init__(weight, name)
        zero = lambda n: Vector(tensor=torch.zeros(1, 1, dtype=dt, device=dev), name=f"{self.name}__{n}")  # noqa: E731
        self.camera_pose, self.world_point = camera_pose, world_point
        self.image_feature_point, self.focal_length = image_feature_point, focal_length
        self.calib_k1 = calib_k1 if calib_k1 is not None else zero("calib_k1")
        self.calib_k2 = calib_k2 if calib_k2 is not None else zero("calib_k2")

    def optim_vars(self):
        return [self.camera_pose, self.world_point]

    def aux_vars(self):
        return [self.focal_length, self.image_feature_point, self.calib_k1, self.calib_k2] + self.weight.aux_vars()

    def dim(self):
        return 2

    def error(self):
        raise NotImplementedError("Reprojection is evaluated by the fused bundle-adjustment kernels "
                                  "(objective.error_metric() after an optimizer / linearization was built on it)")


# ------------------------------------------------------------------------------------------------
# robust costs (theseus/core/robust_loss.py:13-52, theseus/core/robust_cost_function.py:52-135)
# ------------------------------------------------------------------------------------------------
class RobustLoss(abc.ABC):
    """rho(x) / rho'(x) of the squared norm x of a weighted error.  The fused kernels evaluate these on device
    (csrc/robust.cuh); the torch forms below serve ``RobustCostFunction.weighted_error`` outside the packed path."""
    _LOSS_EPS = 1e-20

    @classmethod
    def evaluate(cls, x: torch.Tensor, log_radius: torch.Tensor) -> torch.Tensor:
        return cls._evaluate_impl(x, log_radius.exp())

    @classmethod
    def linearize(cls, x: torch.Tensor, log_radius: torch.Tensor) -> torch.Tensor:
        return cls._linearize_impl(x, log_radius.exp())


class WelschLoss(RobustLoss):
    @staticmethod
    def _evaluate_impl(x, radius):
        return radius - radius * torch.exp(-x / (radius + RobustLoss._LOSS_EPS))

    @staticmethod
    def _linearize_impl(x, radius):
        return torch.exp(-x / (radius + RobustLoss._LOSS_EPS))


class HuberLoss(RobustLoss):
    @staticmethod
    def _evaluate_impl(x, radius):
        return torch.where(x > radius, 2 * torch.sqrt(radius * x.max(radius) + RobustLoss._LOSS_EPS) - radius, x)

    @staticmethod
    def _linearize_impl(x, radius):
        return torch.sqrt(radius / torch.max(x, radius) + RobustLoss._LOSS_EPS)


class HingeLoss(RobustLoss):
    """robust_loss.py:55-62: no cost inside the radius, sqrt(x) - sqrt(radius) beyond it."""

    @staticmethod
    def _evaluate_impl(x, radius):
        return torch.where(x > radius, torch.sqrt(x) - torch.sqrt(radius), RobustLoss._LOSS_EPS)

    @staticmethod
    def _linearize_impl(x, radius):
        return torch.where(x > radius, 1.0 / (2 * torch.sqrt(x) + RobustLoss._LOSS_EPS), 0.0)


class GNCRobustLoss(RobustLoss, abc.ABC):
    """robust_loss.py:64-89: a loss with a graduated-non-convexity control value ``mu`` next to the radius."""

    @classmethod
    def evaluate(cls, x: torch.Tensor, log_radius: torch.Tensor, mu: torch.Tensor) -> torch.Tensor:   # type: ignore[override]
        return cls._evaluate_impl(x, log_radius.exp(), mu)

    @classmethod
    def linearize(cls, x: torch.Tensor, log_radius: torch.Tensor, mu: torch.Tensor) -> torch.Tensor:   # type: ignore[override]
        return cls._linearize_impl(x, log_radius.exp(), mu)


class GemanMcClureLoss(GNCRobustLoss):
    """robust_loss.py:92-113: mu = 1 the Geman-McClure loss, mu -> infinity the quadratic."""

    @staticmethod
    def _evaluate_impl(x, radius, mu):
        return mu * radius * x / (mu * radius + x + RobustLoss._LOSS_EPS)

    @staticmethod
    def _linearize_impl(x, radius, mu):
        return (mu * radius) ** 2 / ((mu * radius + x) ** 2 + RobustLoss._LOSS_EPS)


class RobustCostFunction(CostFunction):
    """Wraps a cost function: its weighted error / Jacobians are rescaled by sqrt(rho'(|w e|^2) + eps) in the
    linearization and its contribution to the objective is rho(|w e|^2) (robust_cost_function.py:52-135)."""
    _EPS = 1e-20

    def __init__(self, cost_function: CostFunction, loss_cls, log_loss_radius: Variable, flatten_dims: bool = False,
                 name: Optional[str] = None):
        self.cost_function = cost_function
        super().__init__(cost_function.weight, name=name)
        self.log_loss_radius = log_loss_radius
        self.loss = loss_cls()
        self.flatten_dims = flatten_dims

    def optim_vars(self):
        return self.cost_function.optim_vars()

    def aux_vars(self):
        return self.cost_function.aux_vars() + [self.log_loss_radius]

    def dim(self):
        return self.cost_function.dim()

    def weighted_error(self):
        we = self.cost_function.weighted_error()
        if self.flatten_dims:
            we = we.reshape(-1, 1)
        rho = self._evaluate_loss((we ** 2).sum(dim=1, keepdim=True))
        if self.flatten_dims:
            return (rho.reshape(-1, self.dim()) + RobustCostFunction._EPS).sqrt()
        return torch.ones_like(we) * (rho / self.dim() + RobustCostFunction._EPS).sqrt()

    def error(self):
        return self.weighted_error()

    def _evaluate_loss(self, squared_norm):
        return self.loss.evaluate(squared_norm, self.log_loss_radius.tensor)


class GNCRobustCostFunction(RobustCostFunction):
    """robust_cost_function.py:173-222: a RobustCostFunction whose loss takes the graduated-non-convexity control value
    ``gnc_control_val`` (an auxiliary variable the caller anneals between optimisations)."""

    def __init__(self, cost_function: CostFunction, loss_cls, log_loss_radius: Variable, gnc_control_val: Variable,
                 flatten_dims: bool = False, name: Optional[str] = None):
        if not issubclass(loss_cls, GNCRobustLoss):
            raise RuntimeError(f"{loss_cls} must be GNCRobustLoss type to initialize GNCRobustCostFunction.")
        super().__init__(cost_function, loss_cls, log_loss_radius, flatten_dims=flatten_dims, name=name)
        self.gnc_control_val = gnc_control_val

    def aux_vars(self):
        return super().aux_vars() + [self.gnc_control_val]

    def _evaluate_loss(self, squared_norm):
        return self.loss.evaluate(squared_norm, self.log_loss_radius.tensor, self.gnc_control_val.tensor)


# ------------------------------------------------------------------------------------------------
# objective
# ------------------------------------------------------------------------------------------------
class Objective:
    """Container of cost functions and their variables (theseus/core/objective.py:42-956).

    The linearization back end (``HipLinearization``) installs a packed device representation in
    ``self._packed`` -- the analogue of the reference's vectorisation hooks
    (core/objective.py:113-146,916-956) -- after which ``error``/``error_metric``/
    ``retract_vars_sequence``/``update`` run as fused HIP kernels on the packed buffers.
    """

    def __init__(self, dtype: Optional[torch.dtype] = None):
        self.dtype = dtype or torch.get_default_dtype()
        self.device = torch.device("cpu")
        self.cost_functions: "OrderedDict[str, CostFunction]" = OrderedDict()
        self.optim_vars: "OrderedDict[str, Variable]" = OrderedDict()
        self.aux_vars: "OrderedDict[str, Variable]" = OrderedDict()
        self.batch_size: Optional[int] = None
        self._packed = None
        self._num_updates_variables: Dict[str, int] = {}
        self.current_version = 0
        self._update_stamp = None   # (Variable._global_updates, current_version) of the last full update()

    # theseus/core/objective.py:148-300 (add / registration, name clash checks)
    def add(self, cost_function: CostFunction):
        if cost_function.name in self.cost_functions:
            raise ValueError(f"Two different cost function objects with the same name "
                             f"({cost_function.name}) are not allowed in the same objective.")
        for v in cost_function.optim_vars() + cost_function.aux_vars():
            if v.dtype != self.dtype:
                raise ValueError(f"Tried to add variable {v.name} with data type {v.dtype} but objective's "
                                 f"data type is {self.dtype}.")
        for v in cost_function.optim_vars():
            if v.name in self.aux_vars:
                raise ValueError(f"Variable {v.name} is already registered as an auxiliary variable.")
            if v.name in self.optim_vars and self.optim_vars[v.name] is not v:
                raise ValueError(f"Two different variable objects with the same name ({v.name}) are not allowed.")
            self.optim_vars[v.name] = v
        for v in cost_function.aux_vars():
            if v.name in self.optim_vars:
                raise ValueError(f"Variable {v.name} is already registered as an optimization variable.")
            if v.name in self.aux_vars and self.aux_vars[v.name] is not v:
                raise ValueError(f"Two different variable objects with the same name ({v.name}) are not allowed.")
            self.aux_vars[v.name] = v
        self.cost_functions[cost_function.name] = cost_function
        self.current_version += 1
        self._packed = None

    def dim(self) -> int:
        return sum(c.dim() for c in self.cost_functions.values())

    def size_cost_functions(self) -> int:
        return len(self.cost_functions)

    def size_variables(self) -> int:
        return len(self.optim_vars)

    def _all_variables(self):
        yield from self.optim_vars.values()
        yield from self.aux_vars.values()

    def get_variable(self, name: str) -> Variable:
        if name in self.optim_vars:
            return self.optim_vars[name]
        if name in self.aux_vars:
            return self.aux_vars[name]
        raise ValueError(f"Named variable {name} is not in the objective.")

    # theseus/core/objective.py:708-727
    def _resolve_batch_size(self):
        """Batch size of the objective (and, in the same pass over the variables, the set of devices their tensors live on: one
        attribute read per variable instead of three property calls -- 9 k variables of a 4096-pose graph are walked on every
        TheseusLayer.forward)."""
        bs, devs = 1, set()
        for v in self._all_variables():
            t = v._tensor
            b = t.shape[0]
            if b != 1:
                if bs != 1 and b != bs:
                    raise ValueError("Provided variable tensors must be broadcastable along batch dimension.")
                bs = b
            devs.add(t.device)
        self.batch_size = bs
        self._resolved_at = (Variable._global_updates, self.current_version)
        return devs

    def _batch_size_is_current(self) -> bool:
        return self.batch_size is not None and getattr(self, "_resolved_at", None) == (Variable._global_updates, self.current_version)

    # theseus/core/objective.py:729-811
    def update(self, input_tensors: Optional[Dict[str, torch.Tensor]] = None,
               batch_ignore_mask: Optional[torch.Tensor] = None):
        input_tensors = input_tensors or {}
        if not input_tensors and self.batch_size is not None and self._update_stamp == (Variable._global_updates, self.current_version):
            return   # nothing was updated anywhere since the last resolve: skip the pass over every variable (42 k of them in
                     # a bundle-adjustment objective: ~30 ms of host time per TheseusLayer.forward)
        optim, aux = self.optim_vars, self.aux_vars
        fast = 0
        # the batch size / device set survive this update when every new tensor has its variable's old batch size and device and
        # nothing else was edited since the last resolve: then the pass over ALL variables is skipped (9 k variables of a
        # 4096-pose graph: ~4 ms of the ~20 ms a TheseusLayer.forward spends on the host per call)
        unchanged = self._batch_size_is_current()
        nomask, _TENSOR, _VAR_UPDATE = batch_ignore_mask is None, torch.Tensor, Variable.update
        for name, t in input_tensors.items():
            v = optim.get(name)
            if v is None:
                v = aux.get(name)
            if v is None:
                import warnings
                warnings.warn(f"Attempted to update a tensor with name {name}, which is not associated to any "
                              f"variable in the objective.")
                continue
            cur = v._tensor
            # the commonest case first: a plain tensor of exactly the variable's shape, dtype and device (six attribute reads; the
            # general test below costs twice that, 4096 times per TheseusLayer.forward on a large graph)
            if nomask and type(t) is _TENSOR and t.shape == cur.shape and t.dtype is cur.dtype and t.device == cur.device \
                    and type(v).update is _VAR_UPDATE:
                v._tensor = t
                v._num_updates += 1
                fast += 1
                continue
            # the common case inline (a plain tensor of the variable's record shape and dtype, no mask): Variable.update's checks
            # and effects without 4096 method calls; everything else goes through it
            if batch_ignore_mask is None and type(t) is torch.Tensor and t.dtype == cur.dtype and t.shape[1:] == cur.shape[1:] \
                    and t.ndim == cur.ndim and type(v).update is Variable.update:
                if unchanged and (t.shape[0] != cur.shape[0] or t.device != cur.device):
                    unchanged = False
                v._tensor = t
                v._num_updates += 1
                fast += 1
            else:
                unchanged = False
                v.update(t, batch_ignore_mask=batch_ignore_mask)
        Variable._global_updates += fast
        if unchanged:
            self._resolved_at = (Variable._global_updates, self.current_version)
        else:
            devs = self._resolve_batch_size()
            if len(devs) == 1:
                self.device = devs.pop()
        self._update_stamp = (Variable._global_updates, self.current_version)

    def to(self, *args, **kwargs):
        for v in self._all_variables():
            v.to(*args, **kwargs)
        dev, dtype, *_ = torch._C._nn._parse_to(*args, **kwargs)
        self.device = dev or self.device
        self.dtype = dtype or self.dtype
        self._packed = None

    # theseus/core/objective.py:562-641
    def error(self, input_tensors: Optional[Dict[str, torch.Tensor]] = None, also_update: bool = False):
        saved = None
        if input_tensors is not None:
            if not also_update:
                saved = {n: self.get_variable(n).tensor for n in input_tensors}
            self.update(input_tensors)
        if self._packed is not None:
            err = self._packed.error_vector()
        else:
            err = torch.cat([c.weighted_error() for c in self.cost_functions.values()], dim=1)
        if saved is not None:
            self.update(saved)
        return err

    def error_metric(self, input_tensors: Optional[Dict[str, torch.Tensor]] = None, also_update: bool = False):
        """0.5 * squared norm of the weighted error vector, per batch item (objective.py:37-38,615-641)."""
        if self._packed is not None:
            saved = None
            if input_tensors is not None:
                if not also_update:
                    saved = {n: self.get_variable(n).tensor for n in input_tensors}
                self.update(input_tensors)
            out = self._packed.error_metric()
            if saved is not None:
                self.update(saved)
            return out
        return (self.error(input_tensors, also_update) ** 2).sum(dim=1) / 2

    # theseus/core/objective.py:873-914
    def retract_vars_sequence(self, delta: torch.Tensor, ordering, ignore_mask: Optional[torch.Tensor] = None,
                              force_update: bool = False):
        var_idx = 0
        mask = None if force_update else ignore_mask
        for var in ordering:
            new_var = var.retract(delta[:, var_idx:var_idx + var.dof()])
            var.update(new_var.tensor, batch_ignore_mask=mask)
            var_idx += var.dof()
