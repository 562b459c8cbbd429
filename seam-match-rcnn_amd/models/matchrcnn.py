"""Image Match R-CNN (phase 1), mirror of reference ``models/matchrcnn.py``.

``matchrcnn_resnet50_fpn(pretrained=False, progress=True, num_classes=91,
pretrained_backbone=True, **kwargs)``                                ref :481-492
eval forward -> per image ``boxes, labels, scores, masks, match_features, w, b``
(no ``roi_features`` key; fallback detection score 1.0)              ref :373-379,451-468

Same pipeline as ``video_matchrcnn`` without the temporal aggregator.  The reference runs
RoIAlign a second time for the match branch on the same boxes (ref :463); the result is
identical, so the 14x14 ROI features are computed once here.
"""
from __future__ import annotations

from . import detection as det
from .match_head import MatchPredictor
from .video_matchrcnn import TemporalRoIHeads, VideoMatchRCNN, model_urls  # noqa: F401

# non-default RPN / RoI-pool kwargs of the reference (ref :14-29), expressed for this build's ctor
params = {
    'rpn_pre_nms_top_n_train': 2000,
    'rpn_pre_nms_top_n_test': 1000,
    'rpn_post_nms_top_n_test': 4000,
    'rpn_post_nms_top_n_train': 8000,
}


class NewRoIHeads(TemporalRoIHeads):
    video = False
    fallback_score = 1.0

    def __init__(self, num_classes=91, n_frames=None, *a, **k):
        super().__init__(num_classes, n_frames, *a, **k)
        self.temporal_aggregator = None         # phase-1 model has no aggregator (ref :333-472)


class MatchRCNN(VideoMatchRCNN):
    roi_heads_cls = NewRoIHeads

    def __init__(self, backbone, num_classes, **kwargs):
        super().__init__(backbone, num_classes, None, **kwargs)

    def load_saved_matchrcnn(self, sd):
        self.load_state_dict(sd, strict=False)


def matchrcnn_resnet50_fpn(pretrained=False, progress=True, num_classes=91, pretrained_backbone=True, **kwargs):
    if pretrained:
        pretrained_backbone = False
    backbone = det.resnet_fpn_backbone('resnet50', pretrained_backbone)
    model = MatchRCNN(backbone, num_classes, **kwargs)
    if pretrained:
        raise RuntimeError("pretrained=True needs a download (" + model_urls['maskrcnn_resnet50_fpn_coco'] +
                           "); fetch it yourself and call model.load_state_dict(state_dict)")
    return model
