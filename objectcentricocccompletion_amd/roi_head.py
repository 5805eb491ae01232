"""TrackletRoIHeadOCC / TrackletDetectorOCC -- host mirror of
mmdet3d/models/roi_heads/tracklet_roi_head_occ.py (forward_train :85-114, simple_test
:492-610, test_occ :394-486, _bbox_forward(_train) :759-878, _assign_and_sample :880-991,
_select_one2one_candidates :993-1030, tracklets2rois :1045-1060, get_gt_rois :1065-1075) and
mmdet3d/models/detectors/tracklet_detector_occ.py (:96-198, :313-345).  Same type strings,
constructor arguments, loss-dict keys and result-dict keys; tracklets are the plain-tensor
``Tracklet`` of tracklet.py."""
import os

import torch
from torch import nn

from ._lib import const_tensor
from .bbox import points_box_to_box, rotation_3d_in_axis
from .registry import BBOX_ASSIGNERS, DETECTORS, HEADS, ROI_EXTRACTORS
from .tracklet import SamplingResult, Tracklet


BATCHED_ASSIGN = os.environ.get('OCOCC_BATCHED_ASSIGN', '1') == '1'   # assignment / sampling of all tracklets at once


def bbox3d2roi(bbox_list):
    """[N_i, 7] per sample -> [sum N_i, 8] with the sample index in column 0."""
    # (the sample index column for the whole list at once: the per-sample fill + concatenation was three launches per
    # tracklet, ~200 of a 64-tracklet step's)
    from .tracklet import host_index
    boxes = torch.cat(list(bbox_list), 0)
    col = [float(i) for i, b in enumerate(bbox_list) for _ in range(b.size(0))]
    idx = host_index(col, boxes.device, dtype=boxes.dtype) if boxes.is_floating_point() else host_index(
        [int(v) for v in col], boxes.device, dtype=boxes.dtype)
    rois = torch.cat([idx.view(-1, 1), boxes], dim=-1)
    # (largest sample index present + 1, known from the list's shapes: what the head would otherwise read back from
    # column 0 -- ococc_bbox_head.py:851 ``int(rois_batch_idx.max().item() + 1)``)
    rois._ococc_batch_size = max((i + 1 for i, b in enumerate(bbox_list) if b.size(0) > 0), default=0)
    return rois


@HEADS.register_module()
class TrackletRoIHeadOCC(nn.Module):

    def __init__(self, num_classes=3, roi_extractor=None, bbox_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None, general_cfg=dict(), history_only=False):
        super().__init__()
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.general_cfg, self.num_classes = general_cfg, num_classes
        self.with_roi_scores = general_cfg.get('with_roi_scores', False)
        self.with_roi_corners = general_cfg.get('with_roi_corners', False)
        self.roi_extractor = ROI_EXTRACTORS.build(roi_extractor)
        bh = dict(bbox_head)
        bh['train_cfg'], bh['test_cfg'] = train_cfg, test_cfg
        self.bbox_head = HEADS.build(bh)
        if train_cfg is not None:
            self.bbox_assigner = BBOX_ASSIGNERS.build(train_cfg['assigner'])
        self.history_only = history_only

    # ------------------------------------------------------------------ training
    def forward_train(self, pts_xyz, pts_feats, pts_batch_idx, pts_frame_inds, img_metas, tracklet_list,
                      gt_candidates_list, gt_occs_list, gt_occ_scores_list):
        samples = self._assign_and_sample(tracklet_list, gt_candidates_list, gt_occs_list, gt_occ_scores_list,
                                          pts_batch_idx, pts_frame_inds)
        return dict(self._bbox_forward_train(pts_xyz, pts_feats, pts_batch_idx, pts_frame_inds, samples)['loss_bbox'])

    def _select_one2one_candidates(self, tracklet_list, candidates_list, gt_occs_list, gt_occ_scores_list):
        """Per proposal tracklet: the GT candidate with most frames of IoU > candidate_thresh
        (tracklet_roi_head_occ.py:993-1030).  The reference walks tracklets and candidates and reads one count back per
        pair; here the timestamp matching of the whole batch is done on the host first, ALL (proposal box, candidate
        box) pairs go through one aligned-IoU launch, the counts come back in one read, and the per-frame IoUs of the
        chosen candidates are left with the proposals for the assigner (Tracklet.self_ious)."""
        from .tracklet import aligned_iou_3d, host_index
        cfg = self.train_cfg if self.train_cfg is not None else self.test_cfg
        thr = cfg.get('candidate_thresh', 0.5)
        dev = tracklet_list[0].device
        p_off, c_off, i1, i2, seg, spans = 0, 0, [], [], [], []   # spans[(t, c)] = (start, end, own frame indices)
        c_boxes = []
        for t, (trk, cands) in enumerate(zip(tracklet_list, candidates_list)):
            for c, cand in enumerate(cands):
                a, b = trk.common_frames(cand)
                spans.append((t, c, len(i1), len(i1) + len(a), a))
                i1 += [p_off + v for v in a]
                i2 += [c_off + v for v in b]
                seg += [len(spans) - 1] * len(a)
                c_boxes.append(cand.boxes[:, :7])
                c_off += len(cand)
            p_off += len(trk)
        counts, ious = [0] * len(spans), None
        if i1:
            P = torch.cat([t.boxes[:, :7] for t in tracklet_list], 0)
            C = torch.cat(c_boxes, 0)
            ious = aligned_iou_3d(P[host_index(i1, dev)], C[host_index(i2, dev)])
            hits = torch.zeros(len(spans), device=dev).index_add_(0, host_index(seg, dev), (ious > thr).float())
            counts = [int(v) for v in hits.tolist()]          # the one read-back of the selection
        best = {}
        for k, (t, c, lo, hi, own) in enumerate(spans):
            if t not in best or counts[k] > counts[best[t]]:   # first maximum, as torch.argmax
                best[t] = k
        out_trks, out_occs, out_scores = [], [], []
        for t, (trk, cands, occs, scores) in enumerate(zip(tracklet_list, candidates_list, gt_occs_list or [None] * len(tracklet_list),
                                                            gt_occ_scores_list or [None] * len(tracklet_list))):
            if len(cands) == 0:
                out_trks.append(trk.new_empty())
                out_occs.append(None)
                out_scores.append(None)
                continue
            _, c, lo, hi, own = spans[best[t]]
            out_trks.append(cands[c])
            out_occs.append(occs[c] if occs is not None else None)
            out_scores.append(scores[c] if scores is not None else None)
            if hi > lo and own == list(range(len(trk))):
                over = ious[lo:hi]
            else:
                over = trk.boxes.new_zeros(len(trk))
                if hi > lo:
                    over = over.index_copy(0, host_index(own, dev), ious[lo:hi])
            # valid for this candidate object and for the box tensors as they are now (in-place transforms bump _version)
            trk._self_iou_cache = (cands[c], trk.boxes, over, trk.boxes._version, cands[c].boxes._version)
        return out_trks, out_occs, out_scores

    def _assign_and_sample_batched(self, tracklet_list, gts, occs, occ_scores, pts_batch_idx, pts_frame_inds):
        """The loop below for TrackletAssigner without object_centric / keep_frame_inds, where everything but the box,
        score and IoU rows themselves follows from timestamps, i.e. is known on the host: the index lists of the whole
        batch go up in ONE copy, the rows are gathered from the concatenated tensors in one launch per field, and the
        per-tracklet SamplingResults are views of those (the reference's per-tracklet form costs ~25 launches and 3-4
        small uploads per tracklet: half of the launches of a 64-tracklet step).  Same fields, same values."""
        from .tracklet import host_index
        dev = tracklet_list[0].device
        shift_on = bool(self.train_cfg.get('random_shift_frame_inds', False))
        n_of, m_of, idx_of, pos_of, neg_of, shifts = [], [], [], [], [], []
        for trk, gt in zip(tracklet_list, gts):
            n, m = len(trk), len(gt)
            if m == 0 or n == 0:
                idx = [0 if m == 0 else -1] * n
            else:
                idx = [gt.get_index_from_ts(ts) + 1 for ts in trk.ts_list]
            n_of.append(n)
            m_of.append(m)
            idx_of.append(idx)
            pos_of.append([i for i, g in enumerate(idx) if g > 0])
            neg_of.append([i for i, g in enumerate(idx) if g == 0])
            shifts.append(int(torch.randint(0, 200 - n + 1, (1,)).item()) if shift_on else 0)   # (CPU generator)
        # global row numbers into the concatenated tensors
        big, cuts = [], []

        def part(values):
            cuts.append((len(big), len(big) + len(values)))
            big.extend(values)
            return len(cuts) - 1

        box_off, gt_off = 0, 0
        pos_g, neg_g, order_g, pos_gt_g, pos_assigned, frame_inds, labels_pos = [], [], [], [], [], [], []
        pos_loc = [i for p in pos_of for i in p]      # the indices the reference's fields hold: local to the tracklet
        neg_loc = [i for q in neg_of for i in q]
        for t, trk in enumerate(tracklet_list):
            order = pos_of[t] + neg_of[t]
            pos_g += [box_off + i for i in pos_of[t]]
            neg_g += [box_off + i for i in neg_of[t]]
            order_g += [box_off + i for i in order]
            pos_gt_g += [gt_off + idx_of[t][i] - 1 for i in pos_of[t]]
            pos_assigned += [idx_of[t][i] - 1 for i in pos_of[t]]
            frame_inds += [i + shifts[t] for i in order]
            labels_pos += [gts[t].type] * len(pos_of[t])
            box_off += n_of[t]
            gt_off += m_of[t]
        k_pos, k_neg, k_ord, k_pgt, k_pas, k_fr, k_lab, k_shift, k_ploc, k_nloc = (part(v) for v in (
            pos_g, neg_g, order_g, pos_gt_g, pos_assigned, frame_inds, labels_pos, shifts, pos_loc, neg_loc))
        up = host_index(big, dev)
        take = lambda k: up[cuts[k][0]:cuts[k][1]]
        cur_all = torch.cat([t.concated_boxes() for t in tracklet_list], 0)
        scores_all = torch.cat([t.concated_scores().detach() for t in tracklet_list], 0)
        over_all = torch.cat([trk.self_ious(gt) if (m and n) else trk.boxes.new_zeros(n)
                              for trk, gt, n, m in zip(tracklet_list, gts, n_of, m_of)], 0)
        have_gt = [gt.concated_boxes() for gt, m in zip(gts, m_of) if m > 0]
        n_pos = [len(p) for p in pos_of]
        n_neg = [len(q) for q in neg_of]
        n_ord = [a + b for a, b in zip(n_pos, n_neg)]
        pos_boxes = cur_all[take(k_pos)].split(n_pos)
        neg_boxes = cur_all[take(k_neg)].split(n_neg)
        if have_gt:
            gt_all = torch.cat(have_gt, 0)
            pos_gt_boxes = gt_all[take(k_pgt)].split(n_pos)
        iou = over_all[take(k_ord)].detach().split(n_ord)
        boxes_ord = cur_all[take(k_ord)].split(n_ord)     # (= cat(pos_bboxes, neg_bboxes) per tracklet, without the cats)
        scores = scores_all[take(k_ord)].split(n_ord)
        pos_local = take(k_ploc).split(n_pos)
        neg_local = take(k_nloc).split(n_neg)
        pas = take(k_pas).split(n_pos)
        frames = take(k_fr).split(n_ord)
        labels = take(k_lab).split(n_pos)
        if shift_on:   # the points of tracklet t move by the same shift as its boxes (one gather + one add for the batch)
            pts_frame_inds += take(k_shift)[pts_batch_idx.long()].to(pts_frame_inds.dtype)
        results = []
        for t, (trk, gt) in enumerate(zip(tracklet_list, gts)):
            s = SamplingResult.__new__(SamplingResult)
            s.pos_inds, s.neg_inds = pos_local[t], neg_local[t]
            s.pos_bboxes, s.neg_bboxes = pos_boxes[t], neg_boxes[t]
            s._bboxes = boxes_ord[t]
            s.num_gts = m_of[t]
            s.pos_assigned_gt_inds = pas[t]
            gtb = gt.concated_boxes()
            s.pos_gt_bboxes = gtb.new_zeros((0, 7)) if gtb.numel() == 0 else pos_gt_boxes[t]
            s.pos_gt_labels = labels[t]
            s.iou, s.scores, s.bboxes_frame_inds = iou[t], scores[t], frames[t]
            s.occ_labels, s.occ_scores = occs[t], occ_scores[t]
            results.append(s)
        return results

    def _assign_and_sample(self, tracklet_list, candidates_list, gt_occs_list, gt_occ_scores_list, pts_batch_idx,
                           pts_frame_inds):
        gts, occs, occ_scores = self._select_one2one_candidates(tracklet_list, candidates_list, gt_occs_list,
                                                                gt_occ_scores_list)
        from .tracklet import TrackletAssigner
        if (BATCHED_ASSIGN and type(self.bbox_assigner) is TrackletAssigner and not self.bbox_assigner.object_centric
                and not self.train_cfg.get('keep_frame_inds', True) and all(len(t) > 0 for t in tracklet_list)):
            return self._assign_and_sample_batched(tracklet_list, gts, occs, occ_scores, pts_batch_idx, pts_frame_inds)
        results = []
        for tid, (trk, gt) in enumerate(zip(tracklet_list, gts)):
            cur = trk.concated_boxes()
            assign = self.bbox_assigner.assign(trk, gt)
            s = SamplingResult(assign, cur, gt.concated_boxes())
            order = torch.cat([s.pos_inds, s.neg_inds])
            s.iou = assign.max_overlaps[order].detach()
            n = len(cur)
            base = torch.arange(n, device=trk.device, dtype=torch.long)
            if self.train_cfg.get('keep_frame_inds', True):
                fr = torch.unique(pts_frame_inds[pts_batch_idx == tid])
                if len(fr) < n:
                    fr = torch.cat([fr, fr[-1:].repeat(n - len(fr))])
                s.bboxes_frame_inds = fr[order]
            elif self.train_cfg.get('random_shift_frame_inds', False):
                shift = int(torch.randint(0, 200 - n + 1, (1,)).item())
                pts_frame_inds[pts_batch_idx == tid] += shift
                s.bboxes_frame_inds = base[order] + shift
            else:
                s.bboxes_frame_inds = base[order]
            s.scores = assign.scores[order]
            s.occ_labels, s.occ_scores = occs[tid], occ_scores[tid]
            results.append(s)
        return results

    def _bbox_forward_train(self, pts_xyz, pts_feats, pts_batch_idx, pts_frame_inds, sampling_results):
        rois = bbox3d2roi([r.bboxes for r in sampling_results])
        roi_frame_inds = torch.cat([r.bboxes_frame_inds for r in sampling_results])
        roi_scores = torch.cat([r.scores for r in sampling_results])
        res = self._bbox_forward(pts_xyz, pts_feats, pts_batch_idx, pts_frame_inds, rois, roi_scores, roi_frame_inds)
        pre = self.train_cfg.get('transform_occ_pre', False)
        targets = self.bbox_head.get_targets(sampling_results, self.train_cfg, transform_occ=pre,
                                             num_occ_per_tracklet=self.train_cfg.get('num_occ_per_tracklet', -1))
        loss = self.bbox_head.loss(res, rois, *targets, transform_occ=not pre, roi_frame_inds=roi_frame_inds)
        labels = targets[0].view(-1) > 0.5
        preds = res['cls_score'].view(-1).sigmoid().detach() > 0.5
        # four of the reference's five logging figures (tracklet_roi_head_occ.py:803-824) from ONE reduction: the counts are sums of
        # 0 / 1 in f32 (exact), so every figure has the value its own float().sum() chain gives -- in a dozen launches
        # instead of three dozen
        hit, miss = preds & labels, ~(preds | labels)
        c = torch.stack([hit, preds, labels, miss, ~preds, ~labels]).float().sum(1)
        # (index tensors from the constant cache: a Python list as an index is a pageable host-to-device copy = a host sync)
        num, den = const_tensor((0, 0, 3, 3), c.device, torch.long), const_tensor((1, 2, 4, 5), c.device, torch.long)
        ratios = c[num] / (c[den] + 1e-6)
        loss['acc'] = (preds == labels).float().mean().detach()
        (loss['precision_posbox'], loss['recall_posbox'], loss['precision_negbox'], loss['recall_negbox']) = ratios.unbind(0)
        res.update(loss_bbox=loss)
        return res

    def _bbox_forward(self, pts_xyz, pts_feats, pts_batch_idx, pts_frame_inds, rois, roi_scores, roi_frame_inds):
        """Pool the points of every RoI, append the RoI score, run the head (:828-878)."""
        assert pts_xyz.size(0) == pts_feats.size(0) == pts_batch_idx.size(0) == pts_frame_inds.size(0)
        inds, roi_inds, info = self.roi_extractor(pts_xyz[:, :3], pts_batch_idx, pts_frame_inds, rois[:, :8],
                                                  roi_frame_inds)
        new_feats, new_xyz = pts_feats[inds], pts_xyz[inds]
        if self.with_roi_scores:
            new_feats = torch.cat([new_feats, roi_scores[roi_inds].unsqueeze(1)], 1)
        if self.with_roi_corners:
            new_feats = torch.cat([new_feats, self.roi_corner_offsets(rois, roi_inds, new_xyz).to(new_feats.dtype)], 1)
        return self.bbox_head(new_xyz, new_feats, info, roi_inds, rois, roi_frame_inds)

    @staticmethod
    def roi_corner_offsets(rois, roi_inds, xyz):
        """27 extra point features of general_cfg.with_roi_corners (tracklet_roi_head_occ.py:861-868): the offsets from
        the point to the 8 corners of its RoI and to a ninth "centre" row, over 10.  The reference takes that ninth row
        from the first three RoI COLUMNS, (batch index, x, y) -- reproduced as is (tests/golden/roi_corners.npz)."""
        from .bbox import box_corners
        c = torch.cat([box_corners(rois[:, 1:]).to(xyz.dtype), rois[:, :3].to(xyz.dtype)[:, None, :]], 1)   # [R, 9, 3]
        return (c[roi_inds.long()] - xyz[:, None, :]).reshape(xyz.size(0), 27) / 10

    # ------------------------------------------------------------------ inference
    def tracklets2rois(self, tracklets):
        rois = bbox3d2roi([t.concated_boxes()[:, :7] for t in tracklets])
        frames = torch.cat([torch.arange(len(t), device=rois.device, dtype=torch.long) for t in tracklets])
        return rois, frames, torch.cat([t.concated_scores() for t in tracklets]), \
            torch.cat([t.concated_labels() for t in tracklets])

    def get_gt_rois(self, tracklets, gt_tracklets):
        boxes, masks = zip(*[gt.concated_boxes_from_ts(trk.ts_list) for trk, gt in zip(tracklets, gt_tracklets)])
        return torch.cat([torch.cat(masks, 0)[:, None].float(), torch.cat(boxes, 0)], 1)

    def simple_test(self, pts_xyz, pts_feats, pts_batch_idx, pts_frame_inds, img_metas, tracklet_list,
                    gt_candidates_list=None, gt_occs_list=None, gt_occ_scores_list=None, **kwargs):
        """One tracklet at a time (batch size 1, as the reference asserts): ``[dict(out_tracklets=[refined
        tracklet], inters, unions, gt_boxes)]`` (tracklet_roi_head_occ.py:492-610).  The refined tracklet is a
        copy of the proposal updated through Tracklet.update_from_prediction (RoIs without points keep their
        proposal box and score); with test_occ_iou the per-RoI occupancy intersection / union counts follow."""
        assert len(tracklet_list) == 1, 'only support batch size 1'
        gt_rois = gt_occ_list = gt_occ_score_list = None
        if gt_candidates_list is not None:
            gts, gt_occ_list, gt_occ_score_list = self._select_one2one_candidates(
                tracklet_list, gt_candidates_list, gt_occs_list, gt_occ_scores_list)
            gt_rois = self.get_gt_rois(tracklet_list, gts)
        rois, roi_frame_inds, cls_preds, labels_3d = self.tracklets2rois(tracklet_list)
        res = self._bbox_forward(pts_xyz, pts_feats, pts_batch_idx, pts_frame_inds, rois, cls_preds, roi_frame_inds)
        decoded = self.bbox_head.get_bboxes_from_tracklet(rois, res['cls_score'], res['bbox_pred'],
                                                          res['nonempty_roi_mask'], labels_3d, cls_preds, img_metas,
                                                          gt_rois=gt_rois, cfg=self.test_cfg)
        out_tracklets = []
        for i, trk in enumerate(tracklet_list):
            new = trk.clone()
            boxes, scores, labels, valid = decoded[i]
            if self.test_cfg.get('tta', None) is not None:
                boxes = self.inverse_aug(new, boxes, img_metas[i])
            new.update_from_prediction(boxes, scores, labels, valid, to_ego=True)
            out_tracklets.append(new)
        out = dict(out_tracklets=out_tracklets)
        if self.test_cfg.get('test_occ_iou', False):
            out.update(self.test_occ(rois, res['fused_roi_feats'], gt_rois, gt_occ_list, gt_occ_score_list, pts_xyz,
                                     pts_batch_idx, pts_frame_inds, roi_frame_inds))
        return [out]

    @staticmethod
    def inverse_aug(trk, boxes, meta):
        """Undo the test-time augmentation recorded in the sample's meta on the refined boxes [L,7] and on the
        tracklet they refine (tracklet_roi_head_occ.py:746-757): flips first, then the rotation."""
        holder = Tracklet(boxes, list(range(boxes.size(0))))
        if meta.get('pcd_horizontal_flip', False):
            holder.flip('horizontal')
            trk.flip('horizontal')
        if meta.get('pcd_vertical_flip', False):
            holder.flip('vertical')
            trk.flip('vertical')
        if 'pcd_rot_angle' in meta:
            assert getattr(trk, 'rot_angle', meta['pcd_rot_angle']) == meta['pcd_rot_angle']
            holder.rotate(-meta['pcd_rot_angle'])
            trk.rotate(-meta['pcd_rot_angle'])
        return holder.boxes

    @torch.no_grad()
    def test_occ(self, rois, fused_roi_feats, gt_rois, gt_occ_list, gt_occ_score_list, pts_xyz=None,
                 pts_batch_idx=None, pts_frame_inds=None, roi_frame_inds=None):
        """Occupancy IoU counts of one tracklet (tracklet_roi_head_occ.py:268-486): every known GT voxel is
        decoded in every RoI that has a GT box at its timestamp, in chunks of test_cfg.iou_chunk_size RoIs;
        integer inter / union per RoI.  Nothing is counted without occupancy labels, without a matched frame,
        or when the label confidence is under occ_label_thresh.  The decoder is called with (features, points,
        RoI index) -- no [chunk,K,1536] copies.  test_cfg.test_baseline (:289-393) rasterises the points
        accumulated up to each frame instead of decoding."""
        empty = dict(inters=[], unions=[], gt_boxes=[])
        if gt_rois is None or gt_occ_list is None or gt_occ_list[0] is None:
            return empty
        match = gt_rois[:, 0] == 1
        if not bool(match.any()) or float(gt_occ_score_list[0]) < self.bbox_head.occ_label_thresh:
            return empty
        occ_xyz, occ_label = gt_occ_list[0][..., :3], (gt_occ_list[0][..., 3] == 1).long()
        K = occ_xyz.size(0)

        def to_roi_frame(xyz, gb, pb):
            if not self.test_cfg.get('transform_to_gt', True):
                return xyz
            return points_box_to_box(xyz, gb, pb)   # GT box frame -> ego frame -> RoI frame (labels sit at voxel gravity centres)

        if self.test_cfg.get('test_baseline', False):
            return self._test_occ_accumulated_points(rois, gt_rois, match, occ_xyz, occ_label, to_roi_frame, pts_xyz,
                                                     pts_batch_idx, pts_frame_inds, roi_frame_inds)
        chunk = self.test_cfg.get('iou_chunk_size', -1)
        chunk = int(match.numel()) if chunk == -1 else chunk
        pred_boxes, gt_boxes_all, feats = rois[match][:, 1:], gt_rois[match][:, 1:], fused_roi_feats[match]
        decoder = self.bbox_head.occ_ae_head.occ_decoder
        inters, unions, gt_boxes = [], [], []
        for f, pb, gb in zip(torch.split(feats, chunk), torch.split(pred_boxes, chunk), torch.split(gt_boxes_all, chunk)):
            n = gb.size(0)
            xyz = to_roi_frame(occ_xyz[None].repeat(n, 1, 1), gb, pb)
            lab = occ_label[None].repeat(n, 1)
            if self.test_cfg.get('ignore_outside_occ', False):
                half = pb[:, None, 3:6] / 2
                inside = (xyz >= -half).all(-1) & (xyz <= half).all(-1)
            else:
                inside = torch.ones((n, K), dtype=torch.bool, device=xyz.device)
            idx = torch.arange(n, device=xyz.device).repeat_interleave(K)
            cls = decoder.get_cls_from_pred(decoder(f, xyz.reshape(n * K, 3), idx)).view(n, K) * inside
            inters.append(((cls == 1) & (lab == 1)).sum(1).cpu())
            unions.append(((cls == 1) | (lab == 1)).sum(1).cpu())
            gt_boxes.append(gb.cpu())
        return dict(inters=inters, unions=unions, gt_boxes=gt_boxes)

    def _test_occ_accumulated_points(self, rois, gt_rois, match, occ_xyz, occ_label, to_roi_frame, pts_xyz,
                                     pts_batch_idx, pts_frame_inds, roi_frame_inds):
        """The point-accumulation baseline (:289-393): the occupancy of RoI i is the set of 0.2 m cells of its own
        box hit by the pooled points of RoIs 0..i, each in its own RoI frame."""
        inds, roi_inds, info = self.roi_extractor(pts_xyz[:, :3], pts_batch_idx, pts_frame_inds, rois[:, :8],
                                                  roi_frame_inds)
        local = rotation_3d_in_axis(info['local_xyz'][None], info['local_xyz'].new_tensor([torch.pi / 2]), axis=2)[0]
        vs = self.bbox_head.occ_ae_head.voxel_size
        inters, unions, gt_boxes = [], [], []
        for i in torch.nonzero(match).view(-1).tolist():
            gb, pb = gt_rois[i, 1:], rois[i, 1:]
            q = to_roi_frame(occ_xyz[None].clone(), gb[None], pb[None])[0]
            size = pb[3:6]
            dims = torch.ceil(size / vs).to(torch.int32)
            cells = lambda p: torch.floor((p + size[None] / 2) / vs).to(torch.long)
            inb = lambda c: (c >= 0).all(1) & (c < dims[None]).all(1)
            pc = cells(local[roi_inds <= i])
            pc = pc[inb(pc)]
            qc = cells(q)
            qin = inb(qc)
            pred = torch.zeros_like(occ_label)
            if bool(qin.any()):
                grid = torch.zeros(tuple(int(d) for d in dims), dtype=torch.bool, device=q.device)
                grid[pc[:, 0], pc[:, 1], pc[:, 2]] = True
                pred[qin] = grid[qc[qin, 0], qc[qin, 1], qc[qin, 2]].long()
            inters.append(((pred == 1) & (occ_label == 1)).sum(-1, keepdim=True).cpu())
            unions.append(((pred == 1) | (occ_label == 1)).sum(-1, keepdim=True).cpu())
            gt_boxes.append(gb[None])
        return dict(inters=inters, unions=unions, gt_boxes=gt_boxes)


@DETECTORS.register_module()
class TrackletDetectorOCC(nn.Module):
    """Concatenate the per-sample point lists, build batch / frame indices, call the RoI head
    (tracklet_detector_occ.py:96-198).  Points: [n_i, 3 + C] with xyz first; the decorated
    features (intensity, elongation, yaw/pi, size/10 x3, score) follow."""

    def __init__(self, roi_head, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None, **kwargs):
        super().__init__()
        rh = dict(roi_head)
        rh['train_cfg'], rh['test_cfg'] = train_cfg, test_cfg
        self.roi_head = HEADS.build(rh)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    @staticmethod
    def _cat_points(points, pts_frame_inds):
        xyz = torch.cat([p[:, :3] for p in points], 0)
        feats = torch.cat([p[:, 3:] for p in points], 0)
        # (sample index of every point: one repeat with the counts known on the host -- no fill per sample, no read-back)
        from .tracklet import host_index
        dev = points[0].device
        counts = [int(p.size(0)) for p in points]
        batch = torch.repeat_interleave(torch.arange(len(points), device=dev), host_index(counts, dev), output_size=sum(counts))
        return xyz.contiguous(), feats.contiguous(), batch, torch.cat(pts_frame_inds, 0).long()

    @staticmethod
    def fake_points_for_empty_input(tensors):
        """After PointsRangeFilter a sample can arrive without points: give it one zero row
        (tracklet_detector_occ.py:299-311)."""
        return [t.new_zeros((1,) + tuple(t.shape[1:])) if len(t) == 0 else t for t in tensors]

    def forward_train(self, points, pts_frame_inds=None, img_metas=None, tracklet=None, gt_tracklet_candidates=None,
                      occ_labels=None, occ_labels_scores=None):
        """tracklet_detector_occ.py:96-147 (argument names = the keys Collect3D emits, ococcnet.py:247-254)."""
        points = self.fake_points_for_empty_input(points)
        pts_frame_inds = self.fake_points_for_empty_input(pts_frame_inds)
        xyz, feats, batch, frames = self._cat_points(points, pts_frame_inds)
        return self.roi_head.forward_train(pts_xyz=xyz, pts_feats=feats, pts_batch_idx=batch, pts_frame_inds=frames,
                                           img_metas=img_metas, tracklet_list=tracklet,
                                           gt_candidates_list=gt_tracklet_candidates, gt_occs_list=occ_labels,
                                           gt_occ_scores_list=occ_labels_scores)

    def simple_test(self, points, img_metas, pts_frame_inds, tracklet, gt_tracklet_candidates=None, occ_labels=None,
                    occ_labels_scores=None, rescale=False, **kwargs):
        """tracklet_detector_occ.py:149-198."""
        points = self.fake_points_for_empty_input(points)
        pts_frame_inds = self.fake_points_for_empty_input(pts_frame_inds)
        xyz, feats, batch, frames = self._cat_points(points, pts_frame_inds)
        return self.roi_head.simple_test(pts_xyz=xyz, pts_feats=feats, pts_batch_idx=batch, pts_frame_inds=frames,
                                         img_metas=img_metas, tracklet_list=tracklet,
                                         gt_candidates_list=gt_tracklet_candidates, gt_occs_list=occ_labels,
                                         gt_occ_scores_list=occ_labels_scores)

    def forward_test(self, points, img_metas, img=None, **kwargs):
        """tracklet_detector_occ.py:313-345: with test_cfg.tta the arguments are lists over augmentations."""
        if self.test_cfg is not None and self.test_cfg.get('tta', None) is not None:
            return self.aug_test(points, img_metas, **kwargs)
        return self.simple_test(points, img_metas, **kwargs)

    def aug_test(self, points, img_metas, pts_frame_inds, tracklet, rescale=False):
        """Test-time augmentation (tracklet_detector_occ.py:200-221): points / metas / frame indices / tracklets
        are lists over the augmentations, each a batch of ONE sample; every augmentation is refined on its own,
        its boxes are mapped back through the augmentation (TrackletRoIHeadOCC.inverse_aug) and the per-frame
        boxes are merged with Tracklet.merge_augs under test_cfg['tta'].  Returns one merged Tracklet per sample."""
        assert len(points) == len(img_metas) == len(pts_frame_inds) == len(tracklet)
        tta = self.roi_head.test_cfg['tta']
        per_aug = []
        for p, meta, inds, trks in zip(points, img_metas, pts_frame_inds, tracklet):
            outs = []
            for i, trk in enumerate(trks):  # simple_test refines one tracklet per call and undoes the augmentation
                r = self.simple_test([p[i]], [meta[i]], [inds[i]], [trk])[0]
                outs.append(r['out_tracklets'][0])
            per_aug.append(outs)
        bsz = len(points[0])
        return [Tracklet.merge_augs([per_aug[k][i] for k in range(len(points))], tta, points[0][0].device)
                for i in range(bsz)]

    def forward(self, return_loss=True, **kwargs):
        return self.forward_train(**kwargs) if return_loss else self.forward_test(**kwargs)


def occupancy_iou_metrics(results):
    """Aggregate per-RoI (inter, union, gt box) lists exactly as WaymoTrackletDatasetWithOcc.evaluate
    does for metric 'iou' (mmdet3d/datasets/waymo_tracklet_dataset.py:629-672): overall IoU =
    sum inter / sum union, mIoU over tracklets, mIoU over boxes, and the mean box IoU by GT
    volume (<30, [30,150), >=150 m^3)."""
    total_inter = total_union = 0.0
    track, box, small, medium, large = [], [], [], [], []
    for r in results:
        if 'inters' not in r or (len(r['inters']) == 0 and len(r['unions']) == 0):
            continue
        inters = torch.cat(r['inters'], 0)
        unions = torch.cat(r['unions'], 0)
        box_ious = inters / unions
        box.extend(box_ious.tolist())
        if 'gt_boxes' in r:
            vol = torch.cat(r['gt_boxes'], 0)[:, 3:6].prod(1)
            small.extend(box_ious[vol < 30].tolist())
            medium.extend(box_ious[(vol >= 30) & (vol < 150)].tolist())
            large.extend(box_ious[vol >= 150].tolist())
        total_inter += float(inters.sum())
        total_union += float(unions.sum())
        track.append(float(inters.sum() / unions.sum()))
    if not track:
        return {}
    out = dict(iou=total_inter / total_union, miou_track=sum(track) / len(track), miou_box=sum(box) / len(box))
    for name, lst in (('small', small), ('medium', medium), ('large', large)):
        if lst:
            out['iou_' + name] = sum(lst) / len(lst)
    return out
