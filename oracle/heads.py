"""ORACLE (test infrastructure, not product code) -- SEAM match heads on the CPU.

A plain torch-CPU fp32 restatement of the reference's in-tree heads.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product path (``seam-match-rcnn_amd``) never does.

PINNED: every function here is checked in ``tests/test_oracle_golden.py``
against fixtures captured by importing the reference's own ``models/nlb.py`` and
``models/match_head.py`` (generator: ``tests/golden/make_golden.py``).

Parameters are passed as a flat dict keyed by the reference's state-dict names
(``conv_seq.0.weight`` ... ``newnlb.W.bias``), values = torch fp32 CPU tensors.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5   # nn.BatchNorm1d default (ref models/match_head.py:62)


def match_trunk(x: torch.Tensor, p: dict, bn_train: bool = False) -> torch.Tensor:
    """``conv_seq -> pool -> linear`` : [K,256,14,14] -> x3[K,256].

    ref models/match_head.py:50-62 (ctor), :67-69 / :93-95 (forward).
    Four *valid* 3x3 convs + ReLU (14->12->10->8->6), AvgPool 6x6, (no-op) ReLU,
    Linear 1024->256, BatchNorm1d: eval mode = running statistics; ``bn_train`` = what the module does in
    ``.train()`` (batch statistics, running buffers in ``p`` updated in place with momentum 0.1)."""
    for i in (0, 2, 4, 6):
        x = F.relu(F.conv2d(x, p[f"conv_seq.{i}.weight"], p[f"conv_seq.{i}.bias"]))
    x = F.relu(F.avg_pool2d(x, (6, 6)))
    x = F.linear(x.flatten(1), p["linear.0.weight"], p["linear.0.bias"])
    x = F.batch_norm(x, p["linear.1.running_mean"], p["linear.1.running_var"],
                     p["linear.1.weight"], p["linear.1.bias"], bn_train, 0.1 if bn_train else 0.0, BN_EPS)
    return x


def pair_logits(a: torch.Tensor, b: torch.Tensor, w: torch.Tensor, bias: torch.Tensor,
                chunk: int = 64) -> torch.Tensor:
    """``x5[i,j,:] = last((a_i - b_j)^2)`` : a[Q,256], b[G,256] -> [Q,G,2].

    ref models/match_head.py:73-74 and :161-162.  Chunked over Q so the
    [Q,G,256] temporary of the reference never exceeds ``chunk*G*256`` floats."""
    q, g = a.shape[0], b.shape[0]
    out = torch.empty((q, g, w.shape[0]), dtype=a.dtype)
    for s in range(0, q, chunk):
        d = (a[s:s + chunk, None, :] - b[None, :, :]) ** 2
        out[s:s + chunk] = F.linear(d, w, bias)
    return out


def match_predictor_forward(x, types, p, bn_train: bool = False):
    """``MatchPredictor.forward(x, types) -> (x3, x5)``; ref models/match_head.py:66-76."""
    x3 = match_trunk(x, p, bn_train)
    t = torch.as_tensor(types)
    x5 = pair_logits(x3[t == 0], x3[t == 1], p["last.weight"], p["last.bias"])
    return x3, x5


def nlb_closed_form(x: torch.Tensor, p: dict, prefix: str = "newnlb.") -> torch.Tensor:
    """Concatenation-form non-local block on one sequence X[T,256] -> Z[T,256].

    ref models/nlb.py:66-101 with ``NONLocalBlock1D(256, sub_sample=False,
    bn_layer=False)`` (ctor models/match_head.py:87).  Closed form:
      G=X Wg^T+bg, TH=X Wth^T+bth, PH=X Wph^T+bph            (:74-80)
      a=TH.w[:128], b=PH.w[128:], f=ReLU(a_i+b_j)/T           (:82-93)
      Y=f G ; Z=Y Ww^T + bw + X                               (:95-99)
    The reference materialises [1,256,T,T]; only the TxT matrix is needed."""
    t = x.shape[0]
    g = F.linear(x, p[prefix + "g.weight"][:, :, 0], p[prefix + "g.bias"])
    th = F.linear(x, p[prefix + "theta.weight"][:, :, 0], p[prefix + "theta.bias"])
    ph = F.linear(x, p[prefix + "phi.weight"][:, :, 0], p[prefix + "phi.bias"])
    wc = p[prefix + "concat_project.0.weight"].reshape(-1)
    a = th @ wc[:128]
    b = ph @ wc[128:]
    f = F.relu(a[:, None] + b[None, :]) / t
    y = f @ g
    return F.linear(y, p[prefix + "W.weight"][:, :, 0], p[prefix + "W.bias"]) + x


def attention_pool(z: torch.Tensor, p: dict):
    """``sum_t softmax_t(scorer(z)) * z`` : Z[T,256] -> ([256], scores[T,1]).

    ref models/match_head.py:119-121 / :149-151 (scorer :86)."""
    s = F.softmax(F.linear(z, p["attention_scorer.weight"], p["attention_scorer.bias"]), 0)
    return (s * z).sum(0), s


def aggregate_sequences(seqs, p, use_nlb: bool = True):
    """NLB (skipped for length-1 sequences, ref :115-117) + attention pooling."""
    outs, atts = [], []
    for x in seqs:
        z = nlb_closed_form(x, p) if (use_nlb and x.shape[0] > 1) else x
        o, s = attention_pool(z, p)
        outs.append(o[None])
        atts.append(s)
    return torch.cat(outs, 0), atts


def pack_sequences(x3_1: torch.Tensor, ids: torch.Tensor):
    """Mode-A packing rules, ref models/match_head.py:98-111.

    sequences ordered by *sorted unique id*; ``maxlen`` = count of the modal id;
    ``x3_1_seq[1+maxlen, S, 256]`` with a dummy zero row 0; ``x3_1_mask[S,1+maxlen]``
    True on padding."""
    uniq = torch.unique(ids)            # sorted
    counts = [(ids == u).sum().item() for u in uniq]
    maxlen = max(counts)
    s = len(counts)
    seq = torch.zeros((1 + maxlen, s, x3_1.shape[1]), dtype=x3_1.dtype)
    mask = torch.zeros((s, 1 + maxlen), dtype=torch.bool)
    lst = []
    for i, u in enumerate(uniq):
        rows = x3_1[ids == u]
        n = rows.shape[0]
        seq[1:n + 1, i] = rows
        mask[i, n + 1:] = True
        lst.append(rows)
    return seq, mask, lst


def unpack_sequences(x3_1_seq: torch.Tensor, x3_1_mask: torch.Tensor):
    """Mode-B slicing ``1:first_masked`` ; ref models/match_head.py:136-139."""
    lst = []
    for i in range(x3_1_seq.shape[1]):
        m = x3_1_mask[i]
        end = int(m.nonzero()[0].item()) if bool(m.any()) else m.numel()
        lst.append(x3_1_seq[1:end, i])
    return lst


def temporal_aggregation_forward(x, types, ids, p, x3_1_seq=None, x3_1_mask=None, x3_2=None,
                                 getatt=False, bn_train: bool = False):
    """``TemporalAggregationNLB.forward``; ref models/match_head.py:90-169 (both modes)."""
    atts = None
    if x3_1_seq is None:
        x3 = match_trunk(x, p, bn_train)
        t = torch.as_tensor(types)
        idt = torch.as_tensor(ids)
        x3_1 = x3[t == 0]
        x3_1_ids = idt[t == 0]
        if x3_1_ids.numel() > 0:
            x3_1_seq, x3_1_mask, lst = pack_sequences(x3_1, x3_1_ids)
            x3_1b, atts = aggregate_sequences(lst, p)
        else:
            x3_1b = None
        x3_2 = x3[t == 1]
    else:
        lst = unpack_sequences(x3_1_seq, x3_1_mask)
        x3_1b, atts = aggregate_sequences(lst, p)
        x3_1_ids = torch.zeros((1, 2))
    x5 = pair_logits(x3_1b, x3_2, p["last.weight"], p["last.bias"]) if x3_1b is not None else None
    if getatt:
        return x3_1b, x3_2, x5, x3_1_seq, x3_1_mask, x3_1_ids, atts
    return x3_1b, x3_2, x5, x3_1_seq, x3_1_mask, x3_1_ids


def match_scores(x5: torch.Tensor) -> torch.Tensor:
    """``softmax(x5,-1)[...,1]`` ; ref evaluate_movingfashion.py:97-98,265-267."""
    return F.softmax(x5, -1)[..., 1]


def rank_topk(x5: torch.Tensor, k: int):
    """Descending ranking of the match score, first k; ref evaluate_movingfashion.py:99,268.
    Returns (idx[Q,k] int64, score[Q,k]).  Ties: lower index first (the
    reference's ``np.argsort`` is unstable, so tie order is unspecified there)."""
    sc = match_scores(x5)
    # rank on the logit difference (monotone in the score, avoids softmax saturation ties)
    d = x5[..., 1] - x5[..., 0]
    k = min(k, d.shape[1])
    order = torch.argsort(d, dim=1, descending=True, stable=True)[:, :k]
    return order, torch.gather(sc, 1, order)
