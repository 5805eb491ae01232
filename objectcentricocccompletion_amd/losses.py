"""The two mmdet losses the ococcnet config names (configs/ococc/ococcnet.py:86-91,
119-124,131-137): CrossEntropyLoss(use_sigmoid=True) and L1Loss, with mmdet's
weight / avg_factor / reduction semantics (mmdet.models.losses.utils.weight_reduce_loss,
external to the reference checkout).  Elementwise over a few thousand values."""
import torch
import torch.nn.functional as F
from torch import nn

from .registry import LOSSES


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == 'mean':
            return loss.mean()
        if reduction == 'sum':
            return loss.sum()
        return loss
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None,
                 loss_weight=1.0):
        super().__init__()
        assert use_sigmoid and not use_mask, 'only the sigmoid (binary) form is on the ococc path'
        self.use_sigmoid = use_sigmoid
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        if weight is not None:
            weight = weight.float()
        loss = F.binary_cross_entropy_with_logits(cls_score, label.float(), reduction='none')
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class L1Loss(nn.Module):

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        loss = torch.abs(pred - target)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):

    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        reduction = reduction_override if reduction_override else self.reduction
        diff = torch.abs(pred - target)
        loss = torch.where(diff < self.beta, 0.5 * diff * diff / self.beta, diff - 0.5 * self.beta)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


def build_loss(cfg):
    return LOSSES.build(cfg)


def reduce_mean(tensor):
    """mmdet.core.reduce_mean: all-reduce average when a process group is up (the two 4-byte
    collectives of ococc_bbox_head.py:480-496)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return tensor
