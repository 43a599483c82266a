"""utils/loss.py counterpart: `sdf_bce_loss` (:44-62), HIP-backed (clid_loss_fwd_bwd)."""
from __future__ import annotations

import torch

from . import _lib


class _BceLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, label, sigma, weight, weighted):
        lib = _lib.load()
        pred_c = _lib.require_cuda(pred.detach().contiguous(), "pred", torch.float32)
        label_c = _lib.require_cuda(label.detach().contiguous(), "label", torch.float32)
        w_c = _lib.require_cuda(weight.detach().contiguous(), "weight", torch.float32) if weighted else None
        out = torch.zeros(4, device=pred.device, dtype=torch.float32)
        d_pred = torch.empty_like(pred_c)
        _lib.check(
            lib.clid_loss_fwd_bwd(_lib.ptr(pred_c), _lib.ptr(label_c), _lib.ptr(w_c), pred_c.numel(), float(sigma),
                                  int(bool(weighted)), None, 0, 0.0, _lib.ptr(out), _lib.ptr(d_pred), None,
                                  _lib.stream()),
            "clid_loss_fwd_bwd",
        )
        ctx.save_for_backward(d_pred)
        return out[1]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (d_pred,) = ctx.saved_tensors
        return g * d_pred, None, None, None, None


def sdf_bce_loss(pred, label, sigma, weight, weighted=False, bce_reduction="mean"):
    """BCE-with-logits between pred/sigma and sigmoid(label/sigma) (utils/loss.py:44-62)."""
    if bce_reduction != "mean":
        raise NotImplementedError("sdf_bce_loss: only bce_reduction='mean' is used by the mapping loop")
    return _BceLoss.apply(pred, label, sigma, weight, weighted)
