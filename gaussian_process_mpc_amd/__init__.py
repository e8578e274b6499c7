"""MI355X-native hot path of risk-sensitive GP-MPC.

The compute path is the hand-written HIP library ``csrc/libgpmpc_hip.so`` reached
through the C ABI declared in ``include/gpmpc.h``; this package is the host side that
mirrors the reference's Python surface for the path (same class / method names and
argument meaning):

* :class:`GaussianProcessRegression`  (reference ``src/gpr.py``)
* :class:`Dynamics`                   (reference ``src/dynamics.py``)
* :class:`RiskSensitiveMPC`           (reference ``src/mpc.py``)
* :func:`mean_prop_torch`, :func:`variance_prop_torch`, :func:`covariance_prop_torch`
  (reference ``src/tools/uncertainty_prop.py``)

There is no CPU fallback: every numerical entry point raises if the HIP library or a
GPU is missing.
"""
from ._lib import lib, LibraryMissing, GpmpcError, require_gpu          # noqa: F401
from .gpr import GaussianProcessRegression                              # noqa: F401
from .dynamics import Dynamics                                          # noqa: F401
from .mpc import RiskSensitiveMPC                                       # noqa: F401
from .uncertainty_prop import mean_prop_torch, variance_prop_torch, covariance_prop_torch  # noqa: F401
from .simulator import Simulator, PendulumPlant, CartPolePlant          # noqa: F401
from .rollout import GPPack, CostParams, rollout, rollout_fullcov, moment_match   # noqa: F401

__all__ = ["GaussianProcessRegression", "Dynamics", "RiskSensitiveMPC", "mean_prop_torch",
           "variance_prop_torch", "covariance_prop_torch", "GPPack", "CostParams", "rollout",
           "rollout_fullcov", "moment_match", "Simulator", "PendulumPlant", "CartPolePlant", "lib", "require_gpu"]
