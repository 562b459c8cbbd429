import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import seam_match_rcnn_amd.synth as synth
from conftest import to_torch
from oracle import detection as OD, model as OM
from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
torch.set_grad_enabled(False)
sd = to_torch(synth.video_matchrcnn_state(5))
m = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14); m.load_state_dict(sd); m = m.to("cuda:0").eval()
img = torch.from_numpy(synth.frames(300, 1, 800, 800)[0])
t0=time.time()
ofe, osz, opad = OM.extract_features([img], sd)
oprops, oobj, odlt = OM.rpn_proposals(ofe, osz, opad, sd)
print("oracle features+rpn", time.time()-t0, len(oprops[0]))
feats, sizes, orig, padded = m.extract_features([img.cuda()])
props = m.rpn(feats, sizes, padded)
def unmatched(a, b, tol):
    d = (a[:, None, :] - b[None, :, :]).abs().amax(-1)
    mn, arg = d.min(1)
    return (mn > tol).nonzero().view(-1), mn, arg
p, o = props[0].cpu(), oprops[0]
print("proposals", len(p), len(o))
for tol in (1e-3, 1e-2, 5e-2, 0.5):
    ua, mn, _ = unmatched(o, p, tol); ub, mn2, _ = unmatched(p, o, tol)
    print(" tol", tol, "oracle-unmatched", len(ua), "gpu-unmatched", len(ub), "median dist", float(mn.median()), "p99", float(mn.kthvalue(int(0.99*len(mn)))[0]))
ua, mn, _ = unmatched(o, p, 5e-2); print(" oracle-unmatched idx", ua.tolist()[:20]); ub, mn2, _ = unmatched(p, o, 5e-2); print(" gpu-unmatched idx", ub.tolist()[:20])
# position-by-position agreement
n = min(len(p), len(o)); same = ((p[:n]-o[:n]).abs().amax(-1) < 5e-2); print(" same position:", int(same.sum()), "of", n)
# detection with oracle proposals on GPU (stage isolation)
t0=time.time()
ref = OM.detect(ofe, oprops, osz, sd, 0.1)
print("oracle detect", time.time()-t0, len(ref[0]["scores"]))
res = m.roi_heads.detect(feats, [oprops[0].cuda()], sizes)
def cmp(r, g, name):
    rb, gb = r["boxes"], g["boxes"].cpu()
    print(name, len(rb), len(gb))
    d = (rb[:, None] - gb[None]).abs().amax(-1)
    d = torch.where(r["labels"][:, None] == g["labels"].cpu()[None], d, torch.full_like(d, 1e9))
    mn, arg = d.min(1)
    for tol in (1e-3, 1e-2, 5e-2):
        print("  tol", tol, "unmatched oracle", int((mn > tol).sum()), "unmatched gpu", int((d.min(0)[0] > tol).sum()))
    ok = mn < 5e-2
    print("  score max rel err of matched", float(((g["scores"].cpu()[arg[ok]] - r["scores"][ok]).abs() / r["scores"][ok]).max()))
    print("  oracle scores head", r["scores"][:8].tolist(), "tail", r["scores"][-4:].tolist())
    print("  same position", int(((rb[:min(len(rb),len(gb))] - gb[:min(len(rb),len(gb))]).abs().amax(-1) < 5e-2).sum()))
cmp(ref[0], res[0], "detect(oracle proposals)")
res2 = m.roi_heads.detect(feats, props, sizes)
cmp(ref[0], res2[0], "detect(gpu proposals)")
# score gaps among oracle's final detections
s = ref[0]["scores"]; print("min adjacent score gap rel", float(((s[:-1]-s[1:])/s[:-1]).min()))
