"""SEAM video model, mirror of reference ``models/video_matchrcnn.py``.

``videomatchrcnn_resnet50_fpn(pretrained=False, progress=True, num_classes=91,
pretrained_backbone=True, n_frames=3, **kwargs)``                      ref :331-343
``VideoMatchRCNN.forward(images, targets=None)`` -> eval: list of dicts with keys
``boxes, labels, scores, masks, match_features, w, b, roi_features``     ref :239-253,293,310-314
``VideoMatchRCNN.load_saved_matchrcnn(sd)``                              ref :325-328
``model.roi_heads.{match_predictor, temporal_aggregator}``               ref :33-36

The reference subclasses torchvision's ``MaskRCNN``; here the same module tree (same
state-dict keys) is assembled from ``models/detection.py`` and every stage launches the
gfx950 kernels.  Inference only: ``forward`` in training mode raises (the reference's own
training loop runs this model under ``eval()`` + ``no_grad``, stuffs/engine.py:113-116).
"""
from __future__ import annotations

from copy import deepcopy
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F   # softmax of 14 class logits in postprocess_detections only
from torch import nn

from .. import ops
from . import detection as det
from .match_head import MatchPredictor, TemporalAggregationNLB as TemporalAggregation

model_urls = {
    'maskrcnn_resnet50_fpn_coco':
        'https://download.pytorch.org/models/maskrcnn_resnet50_fpn_coco-bf2d0c1e.pth',
}


class TemporalRoIHeads(nn.Module):
    """Eval branch of the reference's ``TemporalRoIHeads`` (ref :19-316).  ``features`` are the
    library's NHWC FPN maps; everything else keeps the reference's meaning."""

    video = True                    # emit 'roi_features' (ref :314); NewRoIHeads does not
    roi_features_contiguous = True   # 'roi_features' is a fresh contiguous NCHW tensor, as the reference returns (ref :314);
    #                                  False: an NCHW-shaped channels_last VIEW of the RoIAlign tile (no transpose pass; fp16 storage on
    #                                  the fp16 path; opt-in for pipelines whose only consumer is the aggregator, e.g. bench.py)
    nms_prefix = 4096                # candidates per image that enter the detection NMS (exactness is checked, see postprocess_detections)
    fallback_score = 0.1            # ref :252 (1.0 in models/matchrcnn.py:377)

    def __init__(self, num_classes=91, n_frames=3, box_score_thresh=0.05, box_nms_thresh=0.5,
                 box_detections_per_img=100):
        super().__init__()
        self.n_frames = n_frames
        self.box_roi_pool = det.MultiScaleRoIAlign(("0", "1", "2", "3"), 7, 2)
        self.box_head = det.TwoMLPHead(256 * 7 * 7, 1024)
        self.box_predictor = det.FastRCNNPredictor(1024, num_classes)
        self.mask_roi_pool = det.MultiScaleRoIAlign(("0", "1", "2", "3"), 14, 2)
        self.mask_head = det.MaskRCNNHeads(256, (256, 256, 256, 256))
        self.mask_predictor = det.MaskRCNNPredictor(256, 256, num_classes)
        self.match_predictor = MatchPredictor()
        self.temporal_aggregator = TemporalAggregation()
        self.keypoint_roi_pool = self.keypoint_head = self.keypoint_predictor = None
        self.score_thresh = box_score_thresh
        self.nms_thresh = box_nms_thresh
        self.detections_per_img = box_detections_per_img
        self.num_classes = num_classes
        self.with_masks = True      # set False to skip the (caller-unused) mask branch
        if not 0 < int(self.nms_prefix) <= det.NMS_MAX_BOXES:
            raise ValueError(f"nms_prefix must be in 1..{det.NMS_MAX_BOXES}, got {self.nms_prefix}")

    has_mask = property(lambda self: self.mask_roi_pool is not None and self.mask_head is not None
                        and self.mask_predictor is not None)
    has_keypoint = property(lambda self: False)
    has_match = property(lambda self: self.match_predictor is not None)

    # ref :154-205 -- softmax, decode (weights 10,10,5,5), clip, drop background, score > thr,
    # remove small, per-class NMS, top-k.  Decode/clip and the NMS bit-matrix + scan are HIP kernels.
    def postprocess_detections(self, class_logits, box_regression, proposals, image_shapes, valid_counts=None):
        """``valid_counts`` ([N] int64 on the device, padded-proposal form): only the first valid_counts[i] proposals of image i
        exist; the rest of its row is padding and can never be detected."""
        num_classes = class_logits.shape[-1]
        counts = [len(p) for p in proposals]
        n_img = len(counts)
        if n_img == 0 or max(counts) == 0:
            e = class_logits.new_zeros((0,))
            return ([e.view(0, 4)] * n_img, [e] * n_img, [e.to(torch.int64)] * n_img)
        dev = class_logits.device
        pred_scores = F.softmax(class_logits, -1)
        # decode + clip (HIP); images of one batch normally share a size, else decode per image
        if all(tuple(s) == tuple(image_shapes[0]) for s in image_shapes):
            boxes = ops.decode_boxes(box_regression.contiguous(), torch.cat(proposals).contiguous(),
                                     (10.0, 10.0, 5.0, 5.0), image_shapes[0])
        else:
            boxes = torch.cat([ops.decode_boxes(r.contiguous(), p.contiguous(), (10.0, 10.0, 5.0, 5.0), s)
                               for r, p, s in zip(box_regression.split(counts, 0), proposals, image_shapes)])
        # pad every image to the same number of proposals so the whole batch is filtered at once
        pmax = max(counts)
        c = (num_classes - 1) * pmax
        pb = boxes.new_zeros((n_img, pmax, num_classes, 4))
        ps = pred_scores.new_full((n_img, pmax, num_classes), -1.0)
        row = torch.cat([torch.arange(k, device=dev) for k in counts])
        img = torch.repeat_interleave(torch.arange(n_img, device=dev), torch.tensor(counts, device=dev))
        pb[img, row] = boxes.view(-1, num_classes, 4)
        ps[img, row] = pred_scores
        if valid_counts is not None:
            real = torch.arange(pmax, device=dev)[None] < valid_counts.to(dev)[:, None]
            ps = torch.where(real[..., None], ps, ps.new_full((), -1.0))
        pb, ps = pb[:, :, 1:].reshape(n_img, c, 4), ps[:, :, 1:].reshape(n_img, c)           # drop background
        labels = torch.arange(1, num_classes, device=dev).repeat(pmax)[None].expand(n_img, -1)
        valid = (ps > self.score_thresh) & ((pb[..., 2] - pb[..., 0]) >= 1e-2) & ((pb[..., 3] - pb[..., 1]) >= 1e-2)
        # Only the NMS_PREFIX best-scored candidates of an image enter the IoU matrix (13 classes x 1000 proposals = 13 000 padded
        # candidates would cost a 21 MB bit matrix per image for 100 detections).  Exact whenever the prefix holds
        # detections_per_img survivors or all valid candidates -- checked on the device and read back with the counts (the one
        # sync); otherwise the call is repeated on everything, or per image when that exceeds the kernel's capacity.
        # (a prefix outside 1..NMS_MAX_BOXES -- the attribute was changed after construction -- is clamped: whenever c exceeds
        # the kernel's per-image capacity SOME prefix must be used, and the exactness check below catches the rest)
        big = c > det.NMS_MAX_BOXES
        prefix = min(max(int(self.nms_prefix), 1), det.NMS_MAX_BOXES) if (c > self.nms_prefix or big) else 0
        order, sel, exact = det.batched_nms_images(pb, ps, labels, valid, self.nms_thresh, self.detections_per_img, prefix)
        stats = torch.cat([sel.sum(1), exact.to(torch.int64)]).tolist()                      # one sync
        if not all(stats[n_img:]):
            if big:                         # beyond the batched NMS kernel's per-image capacity (e.g. 91 classes): per-image path
                if valid_counts is not None:    # padded proposals (every row has pmax entries): padding never passes the score filter
                    pred_scores = torch.where(real.reshape(-1, 1), pred_scores, pred_scores.new_full((), -1.0))
                return self._postprocess_per_image(boxes.view(-1, num_classes, 4), pred_scores, counts)
            order, sel, _ = det.batched_nms_images(pb, ps, labels, valid, self.nms_thresh, self.detections_per_img)
            stats = sel.sum(1).tolist()
        kept = stats[:n_img]
        kb = torch.gather(pb, 1, order[..., None].expand(-1, -1, 4))
        ks, kl = torch.gather(ps, 1, order), torch.gather(labels, 1, order)
        # survivors to the front of each row on the device; the per-image results are views of length kept[i] (a boolean-mask
        # index here would be a second, hidden synchronisation)
        kb, ks, kl = det.compact_rows(sel, [kb, ks, kl], self.detections_per_img)
        return ([kb[i, :k] for i, k in enumerate(kept)], [ks[i, :k] for i, k in enumerate(kept)],
                [kl[i, :k] for i, k in enumerate(kept)])

    def _postprocess_per_image(self, boxes, scores, counts):
        """Same rule as above for candidate sets too large for one batched NMS launch: filter first (one host
        sync per image), and if more than 16384 candidates still pass the score threshold keep the 16384 best
        (NMS keeps at most ``detections_per_img`` of them anyway)."""
        num_classes = scores.shape[-1]
        out_b, out_s, out_l = [], [], []
        for b, s in zip(boxes.split(counts, 0), scores.split(counts, 0)):
            labels = torch.arange(num_classes, device=s.device).view(1, -1).expand_as(s)
            b, s, labels = b[:, 1:].reshape(-1, 4), s[:, 1:].flatten(), labels[:, 1:].flatten()
            keep = (s > self.score_thresh) & ((b[:, 2] - b[:, 0]) >= 1e-2) & ((b[:, 3] - b[:, 1]) >= 1e-2)
            b, s, labels = b[keep], s[keep], labels[keep]
            if s.numel() > 16384:
                top = torch.argsort(s, descending=True, stable=True)[:16384]
                b, s, labels = b[top], s[top], labels[top]
            k = det.batched_nms(b, s, labels, self.nms_thresh, self.detections_per_img)
            out_b.append(b[k]); out_s.append(s[k]); out_l.append(labels[k])
        return out_b, out_s, out_l

    def detect(self, features, proposals, image_shapes):
        """box branch (ref :225-253).  ``proposals``: the reference's list of [k_i,4] tensors, or the padded form
        ``(boxes [N,P,4], counts [N])`` of ``RegionProposalNetwork.forward(padded_out=True)`` (no host sync before the detections)."""
        valid_counts = None
        if isinstance(proposals, tuple):
            padded, valid_counts = proposals
            proposals = list(padded.unbind(0))
        box_features = self.box_roi_pool(features, proposals, image_shapes)
        box_features = self.box_head(box_features)
        class_logits, box_regression = self.box_predictor(box_features)
        boxes, scores, labels = self.postprocess_detections(class_logits, box_regression, proposals, image_shapes, valid_counts)
        result = []
        for i in range(len(boxes)):
            if boxes[i].numel() > 0:
                result.append(dict(boxes=boxes[i], labels=labels[i], scores=scores[i]))
            else:   # empty image -> full-image fallback detection (ref :246-253)
                dev = boxes[i].device
                result.append(dict(
                    boxes=torch.tensor([[0.0, 0.0, float(image_shapes[i][1]), float(image_shapes[i][0])]], device=dev),
                    labels=torch.tensor([0], device=dev),
                    scores=torch.tensor([self.fallback_score], device=dev)))
        return result

    def match_branch(self, features, result, image_shapes, targets=None):
        """mask + match branches on the detections (ref :255-314)."""
        if targets is not None:     # eval with ground truth: prepend GT boxes, score 1 (ref :256-262)
            assert len(targets) == len(result)
            for t, r in zip(targets, result):
                dev = r["boxes"].device
                r["boxes"] = torch.cat([t["boxes"].to(dev), r["boxes"]])
                r["labels"] = torch.cat([t["labels"].to(dev), r["labels"]])
                r["scores"] = torch.cat([torch.ones((t["labels"].numel(),), device=dev), r["scores"]])
        mask_proposals = [r["boxes"] for r in result]
        roi_nhwc = self.mask_roi_pool(features, mask_proposals, image_shapes)      # [K,14,14,256]
        counts = [len(p) for p in mask_proposals]
        side = None
        if self.has_mask and self.with_masks:
            # the mask branch (caller-unused, but part of the output contract) is independent of the match branch: with
            # det.LEVEL_STREAMS it runs on the side stream next to the match trunk, whose launches fill the tails of its own
            if det.LEVEL_STREAMS and roi_nhwc.is_cuda and roi_nhwc.shape[0] > 0:
                cur, side = torch.cuda.current_stream(), det._side_stream(roi_nhwc.device)
                side.wait_stream(cur)
                labels = [r["labels"] for r in result]       # (`roi_nhwc` and the labels outlive the join at the end of this function:
                with torch.cuda.stream(side):                #  fork / join lifetimes, no record_stream -- see det.FeaturePyramidNetwork.forward)
                    logits = self.mask_predictor(self.mask_head(roi_nhwc))
                    probs = det.maskrcnn_inference(logits, labels, self.num_classes)
            else:
                logits = self.mask_predictor(self.mask_head(roi_nhwc))
                probs = det.maskrcnn_inference(logits, [r["labels"] for r in result], self.num_classes)
            for pr, r in zip(probs, result):
                r["masks"] = pr
        if self.has_match and roi_nhwc.shape[0] > 0:
            # types = 0 for image 0's ROIs, 1 for all others (ref :299-307); the pairwise logits the
            # reference computes here are dropped by every caller (ref :309), only x3 is kept
            final_features = self.match_predictor.trunk_nhwc(roi_nhwc)
            # 'roi_features' [K,256,14,14] (ref :314): a contiguous NCHW copy by default (one transpose kernel).  With
            # ``roi_features_contiguous = False`` the exact-fp32 path hands out an NCHW-shaped VIEW of the NHWC tile RoIAlign wrote
            # (torch's channels_last memory format: same shape and values, no transpose pass) -- the aggregator reads that layout
            # back without a copy (on the fp16 path the view is fp16; the default contiguous form is always fp32).
            roi_nchw = None
            if self.video:
                if not self.roi_features_contiguous:
                    # (fp16 path, round 6: the view keeps the tile's fp16 storage -- the aggregator's trunk takes it as it is; the two
                    #  conversion passes f16 NHWC -> f32 NCHW -> f16 NHWC were 3.2 ms of a config-5 step)
                    roi_nchw = roi_nhwc.permute(0, 3, 1, 2)
                else:
                    roi_nchw = ops.nhwc_to_nchw(roi_nhwc)
            off = 0
            for r, c in zip(result, counts):
                r['match_features'] = final_features[off:off + c]
                r['w'] = self.match_predictor.last.weight
                r['b'] = self.match_predictor.last.bias
                if self.video:
                    r['roi_features'] = roi_nchw[off:off + c]
                off += c
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        return result

    def forward(self, features, proposals, image_shapes, targets=None):
        if targets is not None:
            for t in targets:
                assert t["boxes"].dtype.is_floating_point, 'target boxes must of float type'
                assert t["labels"].dtype == torch.int64, 'target labels must of int64 type'
        if self.training:
            raise NotImplementedError("training branch (losses, proposal sampling) is outside the forward hot path")
        result = self.detect(features, proposals, image_shapes)
        return self.match_branch(features, result, image_shapes, targets), {}


class VideoMatchRCNN(nn.Module):
    roi_heads_cls = TemporalRoIHeads
    # True: padded RPN proposals, ONE host synchronisation per forward (the box head then runs on N x post_nms_top_n padded rows);
    # False: the reference's list form (one more sync, the box head on the real proposal count); None (default): True when
    # rpn_post_nms_top_n_test <= 1000 -- the video model -- and False beyond (MatchRCNN's 4000: the padded rows would multiply the
    # box head's work when NMS keeps far fewer).  An instance may set its own `model.ONE_SYNC`.
    ONE_SYNC = None

    def __init__(self, backbone, num_classes, n_frames=3, min_size=800, max_size=1333,
                 rpn_pre_nms_top_n_test=1000, rpn_post_nms_top_n_test=1000, rpn_nms_thresh=0.7,
                 box_score_thresh=0.05, box_nms_thresh=0.5, box_detections_per_img=100, **kwargs):
        super().__init__()
        self.transform = det.GeneralizedRCNNTransform(min_size, max_size)
        self.backbone = backbone
        self.rpn = det.RegionProposalNetwork(rpn_pre_nms_top_n_test, rpn_post_nms_top_n_test, rpn_nms_thresh)
        self.roi_heads = self.roi_heads_cls(num_classes, n_frames, box_score_thresh, box_nms_thresh,
                                            box_detections_per_img)
        self._ignored_kwargs = kwargs       # training-only knobs of torchvision's MaskRCNN ctor

    def set_compute_dtype(self, dtype: torch.dtype):
        """torch.float32 (default: exact fp32 MFMA) or torch.float16 (fp16 MFMA with fp32 accumulation for
        the extractor and the trunks -- BASELINE config 5; descriptors, NLB, match logits stay fp32)."""
        det.set_compute_dtype(self, dtype)
        return self

    def load_saved_matchrcnn(self, sd):
        """phase-1 -> phase-2 hand-off (ref :325-328)."""
        self.load_state_dict(sd, strict=False)
        self.roi_heads.temporal_aggregator.load_state_dict(
            deepcopy(self.roi_heads.match_predictor.state_dict()), strict=False)

    # ---- stages ------------------------------------------------------------------------------------
    def extract_features(self, images: Sequence[torch.Tensor]):
        x, sizes, orig, padded = self.transform(images)
        s2d_padded = x.shape[-1] == 16 and x.shape[1] == padded[0] // 2 + 3        # fp16 path: zero cells around the space-to-depth frame
        return self.backbone(x, s2d_padded), sizes, orig, padded

    def postprocess(self, result, sizes, orig):
        for r, sz, o in zip(result, sizes, orig):
            boxes = det.GeneralizedRCNNTransform.rescale_boxes(r["boxes"], sz, o)
            r["boxes"] = boxes
            if "masks" in r:
                r["masks"] = det.paste_masks_in_image(r["masks"], boxes, o)
        return result

    def forward(self, images, targets=None):
        if self.training:
            raise NotImplementedError(
                "VideoMatchRCNN on HIP is the inference forward; the reference's training loop calls it under "
                "model.eval() + torch.no_grad() (stuffs/engine.py:113-116)")
        images = list(images)
        if not images:
            return []
        feats, sizes, orig, padded = self.extract_features([i.detach() for i in images])
        # padded proposals: the forward synchronises with the device ONCE, at the detection counts (ONE_SYNC = False: the
        # reference's list form, one more sync at the RPN's variable-length split; identical results, tested)
        one_sync = self.ONE_SYNC if self.ONE_SYNC is not None else self.rpn.post_nms_top_n <= 1000
        proposals = self.rpn(feats, sizes, padded, padded_out=bool(one_sync))
        if targets is not None:        # GT boxes arrive in original-image pixels -> resized frame
            targets = [dict(t, boxes=det.GeneralizedRCNNTransform.rescale_boxes(
                t["boxes"].to(feats["0"].device).to(torch.float32), o, s)) for t, s, o in zip(targets, sizes, orig)]
        result, _ = self.roi_heads(feats, proposals, sizes, targets)
        return self.postprocess(result, sizes, orig)

    @torch.no_grad()
    def forward_fixed_rois(self, images, rois: Sequence[torch.Tensor], run_rpn_head: bool = True):
        """Extension used by the BASELINE.json "fixed ROI" configs: the given boxes (resized-image
        pixels, one [k_i,4] tensor per image) replace RPN proposals + box head + NMS; everything
        downstream (RoIAlign 14x14, mask head, match trunk) is the regular forward."""
        feats, sizes, orig, padded = self.extract_features(list(images))
        rpn_out = self.rpn.head(list(feats.values())) if run_rpn_head else None
        dev = feats["0"].device
        counts = [len(b) for b in rois]
        labels = torch.ones(sum(counts), dtype=torch.int64, device=dev).split(counts)      # one fill, per-image views
        scores = torch.ones(sum(counts), device=dev).split(counts)
        result = [dict(boxes=b.to(dev).to(torch.float32), labels=l, scores=s) for b, l, s in zip(rois, labels, scores)]
        result = self.roi_heads.match_branch(feats, result, sizes)
        return result, feats, rpn_out


def videomatchrcnn_resnet50_fpn(pretrained=False, progress=True, num_classes=91, pretrained_backbone=True,
                                n_frames=3, **kwargs):
    if pretrained:
        pretrained_backbone = False
    backbone = det.resnet_fpn_backbone('resnet50', pretrained_backbone)
    model = VideoMatchRCNN(backbone, num_classes, n_frames, **kwargs)
    if pretrained:
        raise RuntimeError("pretrained=True needs a download (" + model_urls['maskrcnn_resnet50_fpn_coco'] +
                           "); fetch it yourself and call model.load_state_dict(state_dict)")
    return model
