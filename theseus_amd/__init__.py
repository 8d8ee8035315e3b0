"""theseus_amd -- MI355X-native batched Gauss-Newton / Levenberg-Marquardt inner loop for Theseus.

Host-side mirror of the reference's modelling + optimizer API for the SE3 / SE2 pose-graph hot path, over
hand-written HIP kernels behind a C ABI (include/theseus_hip.h, theseus_amd/csrc).  No CPU fallback.
"""
from .core import (AutoDiffCostFunction, Between, CostFunction, CostWeight, DiagonalCostWeight, Difference, GemanMcClureLoss,  # noqa: F401
                   GNCRobustCostFunction, GNCRobustLoss, HingeLoss, HuberLoss, Local,
                   Objective, Point2, Point3, Reprojection, RobustCostFunction, RobustLoss, ScaleCostWeight, SE2, SE3, SO2, SO3,
                   Variable, Vector, WelschLoss)
from .kernels import (HipKernels, default_kernels, reset_global_params, set_global_params, set_lie_eps,  # noqa: F401
                      set_se2_eps)
from .layer import TheseusLayer  # noqa: F401
from .linear_solver import HipCholeskySolver, LinearSolver  # noqa: F401
from .sparse import HipSparseCholeskySolver, fill_reducing_ordering, level_ordering  # noqa: F401
from .linearization import HipLinearization, Linearization, VariableOrdering  # noqa: F401
from .nonlinear import (BackwardMode, Dogleg, GaussNewton, LevenbergMarquardt, NonlinearLeastSquares,  # noqa: F401
                        NonlinearOptimizerInfo, NonlinearOptimizerStatus, TrustRegion)
from .packed import PackedPoseGraph, UnsupportedObjective  # noqa: F401
from .ba import HipSchurLinearization, HipSchurSolver, PackedBA  # noqa: F401

# names a reference user would reach for on this path
CholeskyDenseSolver = HipCholeskySolver
DenseLinearization = HipLinearization

__version__ = "0.1.0"
