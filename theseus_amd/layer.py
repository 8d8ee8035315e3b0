"""TheseusLayer mirror (theseus/theseus_layer.py:29-174): forward = Objective.update + optimize."""
from typing import Any, Dict, Optional, Tuple

import torch

from .nonlinear import NonlinearLeastSquares, NonlinearOptimizerInfo


class TheseusLayer(torch.nn.Module):
    def __init__(self, optimizer: NonlinearLeastSquares, vectorize: bool = True, **kwargs):
        super().__init__()
        self.objective = optimizer.objective
        self.optimizer = optimizer
        self._objectives_version = optimizer.objective.current_version

    def forward(self, input_tensors: Optional[Dict[str, torch.Tensor]] = None,
                optimizer_kwargs: Optional[Dict[str, Any]] = None) -> Tuple[Dict[str, torch.Tensor], NonlinearOptimizerInfo]:
        if self._objectives_version != self.objective.current_version:
            raise RuntimeError("The objective was modified after the layer's construction, which is "
                               "currently not supported.")
        optimizer_kwargs = optimizer_kwargs or {}
        self.objective.update(input_tensors)                      # theseus_layer.py:164-174
        info = self.optimizer.optimize(**optimizer_kwargs)
        vars_ = {name: var.tensor for name, var in self.objective.optim_vars.items()}
        return vars_, info

    def to(self, *args, **kwargs):
        super().to(*args, **kwargs)
        self.objective.to(*args, **kwargs)
        return self

    @property
    def device(self):
        return self.objective.device

    @property
    def dtype(self):
        return self.objective.dtype
