"""Implicit occupancy decoder -- host mirror of mmdet3d/models/occ/occ_base.py: PosEncode
(:26-57), OccDecoder (:59-153) and its dense-grid decode get_occ / get_roi_occ (:155-342).  Parameter names: ln.{weight,bias}, conv_occ.<i>.0.weight,
conv_occ.<i>.1.{weight,bias}, conv_occ.<n>.{weight,bias}.

MI355X formulation of the decoder's first layer.  The reference repeats each RoI feature K
times ([R+,K,1536], ococc_bbox_head.py:711), LayerNorms the copies and runs a 1596->512
Linear on every query point.  Here the first Linear is split by columns,
    W0 [512,1596] = [ W_roi [512,1536] | W_pe [512,60] ],
so   W0 . cat(LN(f_roi), pe)  =  W_roi . LN(f_roi)  (once per RoI)  +  W_pe . pe  (per point):
the 400 MB repeated tensor and 96 % of the first layer's FLOPs disappear; the sum is
mathematically identical (fp32 rounding differs in the last bits).
"""
import os

import numpy as np
import torch
from torch import nn

from .._lib import const_tensor
from ..norm import layer_norm_act
from ..sst.sst_ops import build_mlp
from ..linear import tall_addmm
from ..voxel.scatter_points import gather_rows
from . import occ_ops


FUSED_WHOLE_MLP = True   # ... and in ONE launch when the widths are the reference's 60 -> 512 -> 1024 -> 1024 -> 1
FUSED_TRAIN_MLP = os.environ.get('OCOCC_FUSED_TRAIN_MLP', '1') == '1'   # ... and the bf16 TRAINING forward on that launch too (fused_mlp.occ_mlp_train)
FUSED_MLP = True   # bf16 inference of OccDecoder on the per-layer kernels of occ/fused_mlp.py (False: library GEMMs + LN kernels)


class PosEncode(nn.Module):
    """x -> [sin(pi 2^l x^) , cos(pi 2^l x^)] for l = 0..L-1, x^ = x normalised to [-1,1] by
    `bound`; output layout [2L, 3] flattened (occ_base.py:33-57)."""

    def __init__(self, L=10, bound=[-8.0, -8.0, -4.0, 8.0, 8.0, 4.0], use_norm=True):
        super().__init__()
        self.L = L
        self.norm_bound = bound
        self.use_norm = use_norm

    def forward(self, x):
        assert x.size(-1) == 3
        if self.use_norm:
            lo = const_tensor(self.norm_bound[:3], x.device)
            hi = const_tensor(self.norm_bound[3:], x.device)
            x = (x - lo) / (hi - lo) * 2.0 - 1.0
        ori_shape = x.shape[:-1] + (-1,)
        x = x.reshape(-1, 1, 3)
        freq = torch.pow(2, torch.linspace(0.0, self.L - 1, self.L, device=x.device))
        x = x * freq.view(1, self.L, 1)
        x = torch.cat([torch.sin(np.pi * x), torch.cos(np.pi * x)], dim=1)
        return x.view(*ori_shape)


class OccDecoder(nn.Module):

    def __init__(self, roi_feature_channels, occ_mlp, use_positional_encoding=True, pos_encode_L=10,
                 norm_pos=True, norm_cfg=dict(type='LN', eps=1e-3), act='gelu', occ_dropout=0.0,
                 cls_dim=1, pos_thresh=0.5, use_ln=False):
        super().__init__()
        if use_positional_encoding:
            self.pos_encode = PosEncode(L=pos_encode_L, use_norm=norm_pos)
            pos_enc_size = 2 * pos_encode_L * 3
        else:
            self.pos_encode = nn.Identity()
            pos_enc_size = 3
        self.cls_dim = cls_dim
        self.pos_thresh = pos_thresh
        self.roi_feature_channels = roi_feature_channels
        assert cls_dim in (1, 2)
        if occ_mlp is not None:
            self.conv_occ = build_mlp(roi_feature_channels + pos_enc_size, list(occ_mlp) + [self.cls_dim],
                                      norm_cfg, True, act=act, dropout=occ_dropout)
        else:
            self.conv_occ = nn.Linear(roi_feature_channels, self.cls_dim)
        self.use_ln = use_ln
        if use_ln:
            self.ln = nn.LayerNorm(roi_feature_channels)
        # torch.bfloat16: the per-query MLP (4.8 MFLOP per query point, the FLOP hot spot of the model) runs
        # its GEMMs in bf16 with f32 accumulation and keeps bf16 activations; logits come back as f32.
        # None = f32 as the reference trains.
        self.compute_dtype = None

    def _ln(self, x):
        return layer_norm_act(x, self.ln.weight, self.ln.bias, self.ln.eps, 'none') if self.use_ln else x

    # ------------------------------------------------------------------ fused per-layer kernels (occ/fused_mlp.py)
    def _fused_layers(self):
        """[(Linear, LayerNorm)] of the hidden blocks + the head Linear when the MLP has the shape the fused kernels
        take (Linear(bias optional) -> LN with folded GELU [-> folded dropout], widths 512 / 1024, one logit), else None."""
        from ..norm import LayerNorm, FoldedDropout
        if not (isinstance(self.conv_occ, nn.Sequential) and isinstance(self.pos_encode, PosEncode) and self.cls_dim == 1):
            return None
        blocks, head = list(self.conv_occ)[:-1], self.conv_occ[-1]
        if not (isinstance(head, nn.Linear) and len(blocks) >= 2):
            return None
        out, k = [], None
        for b in blocks:
            if not (isinstance(b, nn.Sequential) and len(b) >= 2 and isinstance(b[0], nn.Linear) and isinstance(b[1], LayerNorm)
                    and b[1].fused_act == 'gelu' and all(isinstance(m, (nn.Identity, FoldedDropout)) for m in list(b)[2:])):
                return None
            n = b[0].out_features
            if n not in (512, 1024) or (k is not None and (b[0].in_features != k or k % 64 or k > 1024)):
                return None
            out.append((b[0], b[1]))
            k = n
        return out, head

    def _forward_fused(self, layers, head, roi_features, smp_xyzs, pts_roi_inds):
        """Inference form of forward() on the fused kernels: one launch per layer, the last one with the head folded in."""
        from . import fused_mlp as fm
        D = self.roi_feature_channels
        lin0 = layers[0][0]
        if not hasattr(self, '_fused_weights'):
            self._fused_weights = fm.DecoderWeights()
        pe_cols = lin0.in_features - D
        mats = [lin0.weight[:, D:]] + [l.weight for l, _ in layers[1:]]
        frags = self._fused_weights.get(mats, [fm.pad64(pe_cols)] + [l.in_features for l, _ in layers[1:]])
        roi_part = torch.mm(self._ln(roi_features).float(), lin0.weight[:, :D].t())           # [K, H] f32, once per RoI
        bound = self.pos_encode.norm_bound if self.pos_encode.use_norm else None
        h = fm.pos_encode_bf16(smp_xyzs, self.pos_encode.L, bound)
        idx = pts_roi_inds.to(torch.int32)
        widths = (h.shape[1],) + tuple(l.out_features for l, _ in layers)
        if (FUSED_WHOLE_MLP and widths == fm.OCC_MLP_WIDTHS and all(l.bias is None for l, _ in layers)
                and len({ln.eps for _, ln in layers}) == 1):
            out = fm.occ_mlp(h, roi_part, idx, frags, [ln.weight for _, ln in layers], [ln.bias for _, ln in layers],
                             layers[0][1].eps, head.weight.view(-1), head.bias)
            return out.view(-1, 1)
        out = None
        for i, ((lin, ln), wf) in enumerate(zip(layers, frags)):
            last = i == len(layers) - 1
            h, out = fm.mlp_layer(h, wf, lin.out_features, ln.weight, ln.bias, ln.eps, 'gelu', bias=lin.bias,
                                  add_rows=roi_part if i == 0 else None, add_index=idx if i == 0 else None,
                                  head_weight=head.weight.view(-1) if last else None,
                                  head_bias=head.bias if last else None, want_y=not last)
        return out.view(-1, 1)

    def _forward_fused_train(self, roi_features, smp_xyzs, pts_roi_inds):
        """Training form on the whole-MLP kernel (fused_mlp.occ_mlp_train): one launch forward, which also leaves what the
        backward chain reads.  None when the MLP is not the shape that kernel takes."""
        from . import fused_mlp as fm
        fused = self._fused_layers()
        if fused is None:
            return None
        layers, head = fused
        D = self.roi_feature_channels
        lin0 = layers[0][0]
        pe_cols = lin0.in_features - D
        widths = (fm.pad64(pe_cols),) + tuple(l.out_features for l, _ in layers)
        if not (len(layers) == 3 and widths == fm.OCC_MLP_WIDTHS and all(l.bias is None for l, _ in layers)
                and len({ln.eps for _, ln in layers}) == 1 and head.bias is not None
                and all(p.dtype == torch.float32 for l, ln in layers for p in (l.weight, ln.weight, ln.bias))):
            return None
        ps = {float(ln.fused_dropout) if self.training else 0.0 for _, ln in layers}
        if len(ps) != 1:
            return None
        p = ps.pop()
        thr = int(round(p * 65536)) if p > 0 else 0
        # (CPU generator: reproducible under manual_seed, no device synchronisation -- as norm.layer_norm_act draws its seed)
        seeds = [int(v) for v in torch.randint(0, 2 ** 62, (3,)).tolist()] if thr else (0, 0, 0)
        if not hasattr(self, '_train_weights'):
            self._train_weights = fm.DecoderWeights()
        roi_part = torch.mm(self._ln(roi_features).float(), lin0.weight[:, :D].t())           # [K, 512] f32, once per RoI
        bound = self.pos_encode.norm_bound if self.pos_encode.use_norm else None
        pe = fm.pos_encode_bf16(smp_xyzs, self.pos_encode.L, bound)
        idx = pts_roi_inds if pts_roi_inds.dtype == torch.int32 else pts_roi_inds.to(torch.int32)
        return fm.occ_mlp_train(pe, roi_part, idx.contiguous(), lin0.weight[:, D:], layers[1][0].weight, layers[2][0].weight,
                                [ln.weight for _, ln in layers], [ln.bias for _, ln in layers], layers[0][1].eps,
                                head.weight, head.bias, thr, seeds, self._train_weights)

    def forward(self, roi_features, smp_xyzs, pts_roi_inds):
        """roi_features [K,D], smp_xyzs [N,3], pts_roi_inds [N] in [0,K) -> logits [N, cls_dim]
        (occ_base.py:100-118), first layer factorised as described in the module docstring."""
        if not isinstance(self.conv_occ, nn.Sequential):
            return self.conv_occ(self._ln(roi_features)[pts_roi_inds.long()])
        if self.compute_dtype == torch.bfloat16 and FUSED_MLP and smp_xyzs.is_cuda and not (
                torch.is_grad_enabled() and (roi_features.requires_grad or any(p.requires_grad for p in self.parameters()))):
            fused = self._fused_layers()
            if fused is not None and not (self.training and any(ln.fused_dropout for _, ln in fused[0])):
                return self._forward_fused(fused[0], fused[1], roi_features, smp_xyzs, pts_roi_inds)
            if fused is None:
                from .. import _lib as L
                L.log_once(('occ-decoder', id(type(self)), tuple(m.__class__.__name__ for m in self.conv_occ)),
                           'OccDecoder MLP is outside the fused bf16 kernels (Linear -> LayerNorm with folded GELU [-> dropout] '
                           'blocks of width 512 / 1024 and a one-logit head): running operator by operator')
        if (self.compute_dtype == torch.bfloat16 and FUSED_MLP and FUSED_TRAIN_MLP and smp_xyzs.is_cuda
                and torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing()):
            out = self._forward_fused_train(roi_features, smp_xyzs, pts_roi_inds)
            if out is not None:
                return out
        first = self.conv_occ[0]
        lin = first[0] if isinstance(first, nn.Sequential) else first
        D = self.roi_feature_channels
        roi_part = torch.mm(self._ln(roi_features), lin.weight[:, :D].t())           # [K, H]
        # (rows of roi_part repeated per query point: backward is the HIP segment sum, not torch's sort-based index_add)
        h = tall_addmm(gather_rows(roi_part, pts_roi_inds), self.pos_encode(smp_xyzs), lin.weight[:, D:])
        if lin.bias is not None:
            h = h + lin.bias
        if self.compute_dtype is None:
            if isinstance(first, nn.Sequential):
                for m in list(first)[1:]:
                    h = m(h)
            for m in list(self.conv_occ)[1:]:
                h = m(h)
            return h
        h = h.to(self.compute_dtype)
        with torch.autocast('cuda', dtype=self.compute_dtype):
            if isinstance(first, nn.Sequential):
                for m in list(first)[1:]:
                    h = m(h)
            for m in list(self.conv_occ)[1:]:
                h = m(h)
        return h.float()

    def occ_forward(self, roi_feats_per_points, smp_xyzs):
        """Reference-shaped entry (occ_base.py:120-139): features already gathered per query
        point, any leading shape [..., D] / [..., 3]."""
        x = torch.cat([self._ln(roi_feats_per_points), self.pos_encode(smp_xyzs)], dim=-1)
        return self.conv_occ(x)

    # ------------------------------------------------------------------ dense grid decode (occ_base.py:155-342)
    def _dense_logits(self, roi_feats, sizes, voxel_size, scale_wlh, offset_wlh, chunk=1 << 20):
        """Logits of every cell of every box's dense grid in ONE batched pass: the reference loops over boxes
        and repeats the 1536-wide RoI feature K times per box (occ_base.py:190-194,300-308); here the cells of
        all boxes are one flat list and the decoder's first layer is factorised per RoI (forward())."""
        centers, box, k = occ_ops.dense_voxel_centers_batched(sizes, voxel_size, scale_wlh, offset_wlh)
        out = []
        with torch.no_grad():
            for lo in range(0, centers.size(0), chunk):
                out.append(self.forward(roi_feats, centers[lo:lo + chunk], box[lo:lo + chunk]))
        logits = torch.cat(out, 0) if out else centers.new_zeros((0, self.cls_dim))
        return centers, box, k, logits

    def _occupied(self, logits):
        if self.cls_dim == 1:
            return logits.sigmoid().view(-1) > self.pos_thresh
        return logits[..., 1] > logits[..., 0]

    @staticmethod
    def _to_lidar(pts, box, centers, sizes, yaw):
        """Box frame (origin at the gravity centre) -> LiDAR frame: rotate by the box yaw about z, add the
        bottom centre and half the height (occ_base.py:220-230,330-336; rotation_3d_in_axis axis=2)."""
        # the reference multiplies by the transposed matrix (einsum 'aij,jka->aik'): a clockwise turn
        c, s_ = torch.cos(yaw)[box], torch.sin(yaw)[box]
        x = pts[:, 0] * c + pts[:, 1] * s_
        y = -pts[:, 0] * s_ + pts[:, 1] * c
        out = torch.stack([x, y, pts[:, 2]], 1) + centers[box]
        out[:, 2] += sizes[box, 2] / 2
        return out

    def get_roi_occ(self, roi_feats, rois, voxel_size, scale_wlh, offset_wlh, transform=True, return_score=False,
                    random_sample_size=2048, occ_only=False):
        """occ_base.py:155-235.  occ_only: occupied cell centres of every RoI; otherwise (return_score) a
        random subset of ``random_sample_size`` cells per RoI (all cells when <= 0) with their scores."""
        if roi_feats.size(0) == 0:
            return []
        assert roi_feats.size(0) == rois.size(0), f'{roi_feats.size(0)}, {rois.size(0)}'
        assert rois.size(1) in (8, 10)
        centers, box, k, logits = self._dense_logits(roi_feats, rois[:, 4:7], voxel_size, scale_wlh, offset_wlh)
        if occ_only:
            sel = self._occupied(logits)
        else:
            assert return_score
            if random_sample_size > 0:
                # a random subset of <= random_sample_size cells per RoI (the reference draws torch.randperm(K)
                # per box on the host RNG: the same distribution, not the same stream)
                start = torch.cumsum(k, 0) - k
                key = torch.rand(centers.size(0), device=centers.device) + box.to(torch.float)
                order = torch.argsort(key)
                rank = torch.empty_like(order)
                rank[order] = torch.arange(order.numel(), device=order.device)
                sel = (rank - start[box]) < random_sample_size
            else:
                sel = torch.ones_like(box, dtype=torch.bool)
        pts, pbox = centers[sel], box[sel]
        if transform:
            pts = self._to_lidar(pts, pbox, rois[:, 1:4], rois[:, 4:7], rois[:, 7])
        if return_score:
            score = logits[sel].sigmoid().view(-1, 1) if self.cls_dim == 1 else logits[sel].softmax(-1)[..., 1].view(-1, 1)
            return pts, pbox, score
        return pts, pbox

    def get_occ(self, roi_feats, rois, voxel_size, scale_wlh, offset_wlh, concat_batch=False, local_xyz=None,
                local_pts_roi_inds=None, return_full=False, transform=True):
        """occ_base.py:238-342: per batch sample a list with the occupied cell centres of each of its RoIs
        (LiDAR frame when ``transform``); ``concat_batch`` joins them per sample; ``return_full`` returns every
        cell; ``local_xyz`` substitutes given box-frame points for the prediction."""
        if roi_feats.size(0) == 0:
            return []
        assert roi_feats.size(0) == rois.size(0), f'{roi_feats.size(0)}, {rois.size(0)}'
        assert rois.size(1) in (8, 10)
        # which RoIs belong to which sample: ONE read-back of the batch column, grouped on the host (the reference's
        # torch.nonzero(roi_batch_idx == i) per sample is a synchronisation per sample)
        batch_of = rois[:, 0].long().tolist()
        batch_size = max(batch_of) + 1
        rois_of = [[] for _ in range(batch_size)]
        for j, b in enumerate(batch_of):
            rois_of[b].append(j)
        sizes, yaw, ctr = rois[:, 4:7], rois[:, 7], rois[:, 1:4]
        if local_xyz is not None and not return_full:
            assert len(local_xyz) == len(local_pts_roi_inds), f'{len(local_xyz)}, {len(local_pts_roi_inds)}'
            order = torch.argsort(local_pts_roi_inds, stable=True)
            pts, pbox = local_xyz[order], local_pts_roi_inds[order].long()
        else:
            if return_full:
                pts, pbox, _ = occ_ops.dense_voxel_centers_batched(sizes, voxel_size, scale_wlh, offset_wlh)
            else:
                centers, box, _, logits = self._dense_logits(roi_feats, sizes, voxel_size, scale_wlh, offset_wlh)
                sel = self._occupied(logits)
                pts, pbox = centers[sel], box[sel]
        if transform:
            pts = self._to_lidar(pts, pbox, ctr, sizes, yaw)
        counts = torch.bincount(pbox, minlength=rois.size(0)).tolist()
        per_roi = list(torch.split(pts, counts))
        res = []
        for ids in rois_of:
            cur = [per_roi[j] for j in ids]
            res.append(torch.cat(cur, 0) if concat_batch else cur)
        return res

    def get_cls_from_pred(self, pred):
        if self.cls_dim == 1:
            return (pred.sigmoid() > self.pos_thresh).long().squeeze(-1)
        return pred.argmax(dim=-1)
