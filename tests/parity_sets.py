"""Set comparison for the DISCRETE stages (top-k, NMS): two fp32 implementations whose continuous outputs agree to ~1e-6 can
still select different members when two candidates are nearly tied at a cut (k-th score, IoU threshold).  The comparison here is
exact up to a printed, bounded list of such flips: every reference box needs a partner with the same label whose four
coordinates agree within `tol_px`; what stays unpaired on either side is listed and must not exceed `max_flips`."""
import torch


def pair_boxes(ref_boxes, got_boxes, ref_labels=None, got_labels=None, tol_px=5e-2):
    """-> (partner index in `got` of every ref box or -1, indices of unpaired got boxes, per-ref coordinate distance)."""
    rb, gb = ref_boxes.detach().float().cpu(), got_boxes.detach().float().cpu()
    if len(rb) == 0 or len(gb) == 0:
        return torch.full((len(rb),), -1, dtype=torch.int64), torch.arange(len(gb)), torch.full((len(rb),), float("inf"))
    d = (rb[:, None, :] - gb[None, :, :]).abs().amax(-1)
    if ref_labels is not None:
        d = torch.where(ref_labels.cpu()[:, None] == got_labels.cpu()[None, :], d, torch.full_like(d, float("inf")))
    dist, arg = d.min(1)
    partner = torch.where(dist <= tol_px, arg, torch.full_like(arg, -1))
    taken = torch.zeros(len(gb), dtype=torch.bool)
    taken[partner[partner >= 0]] = True
    return partner, (~taken).nonzero().view(-1), dist


def assert_same_set(ref_boxes, got_boxes, ref_labels=None, got_labels=None, ref_scores=None, got_scores=None, tol_px=5e-2,
                    max_flips=1, what="boxes"):
    """Exact set equality up to <= max_flips unpaired members per side (printed with their scores).  Returns `partner`.
    The default gate is the observed flip count (0 on every full-size case of the suite, printed below when not) + 1."""
    partner, extra, dist = pair_boxes(ref_boxes, got_boxes, ref_labels, got_labels, tol_px)
    missing = (partner < 0).nonzero().view(-1)
    print(f"[{what}] {len(ref_boxes)} reference / {len(got_boxes)} device members, flips: {len(missing)} missing, {len(extra)} extra")
    if len(missing) or len(extra):
        print(f"[{what}] near-tie flips: {len(missing)} reference member(s) without a partner, {len(extra)} extra member(s)")
        for i in missing.tolist():
            s = f" score {float(ref_scores[i]):.7f}" if ref_scores is not None else ""
            print(f"   missing  ref[{i}] = {[round(v, 3) for v in ref_boxes[i].tolist()]}{s}  (nearest candidate {float(dist[i]):.3g} px away)")
        for j in extra.tolist():
            s = f" score {float(got_scores[j]):.7f}" if got_scores is not None else ""
            print(f"   extra    got[{j}] = {[round(v, 3) for v in got_boxes[j].tolist()]}{s}")
    assert len(missing) <= max_flips and len(extra) <= max_flips, (what, len(missing), len(extra))
    assert len(set(partner[partner >= 0].tolist())) == int((partner >= 0).sum()), f"{what}: two reference members share a partner"
    return partner
