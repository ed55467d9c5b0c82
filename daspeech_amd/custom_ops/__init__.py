# Mirrors DASpeech/custom_ops/__init__.py:1 — the drop-in operator surface.
from .dag_loss import (dag_loss, dag_loss_with_alpha_beta, dag_best_alignment, dag_logsoftmax_gather_inplace,
                       torch_dag_loss, torch_dag_best_alignment, torch_dag_logsoftmax_gather_inplace,
                       logsumexp_keepdim)
from .dag_loss import set_lazy_softmax      # extension (not a reference name): backward state of the gather op, see dag_loss.py

__all__ = ["dag_loss", "dag_loss_with_alpha_beta", "dag_best_alignment", "dag_logsoftmax_gather_inplace",
           "torch_dag_loss", "torch_dag_best_alignment", "torch_dag_logsoftmax_gather_inplace", "logsumexp_keepdim"]
