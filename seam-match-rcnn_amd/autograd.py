"""torch.autograd bridges of the match heads.

The training caller (ref stuffs/engine.py:120-121,158-168,183-185) puts ``match_predictor`` and
``temporal_aggregator`` in ``.train()`` and back-propagates the reference's losses through them.  Each
``Function`` below is one fused stage of the heads: forward = the same HIP kernels the inference path uses
(+ BatchNorm1d batch statistics), backward = the gradient kernels of csrc/seam_backward.hip and the
implicit-GEMM conv kernel on rotated weights.  torch is the tape (plus the scalar chain-rule factor of the loss).
"""
from __future__ import annotations

import torch

from . import ops

F32 = torch.float32


class TrunkFunction(torch.autograd.Function):
    """conv_seq (4 valid 3x3 convs + ReLU) -> AvgPool2d(6,6) + ReLU -> Linear -> BatchNorm1d
    (ref models/match_head.py:50-62,67-69).  x NCHW [K,256,14,14] -> x3 [K,256]."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, w2, b2, w3, b3, lw, lb, gamma, beta, running_mean, running_var, bn_train,
                momentum, eps):
        ws, bs = (w0, w1, w2, w3), (b0, b1, b2, b3)
        acts = [ops.nchw_to_nhwc(x.detach().to(F32))]
        for w, b in zip(ws, bs):                                 # 14 -> 12 -> 10 -> 8 -> 6
            acts.append(ops.conv2d(acts[-1], ops.pack_conv(w, b, wino=False), relu=True))
        pool = ops.avgpool(acts[-1])                             # its ReLU is the identity on a mean of ReLU outputs
        lin = ops.linear(pool, ops.pack_conv(lw, lb))
        if bn_train:
            out, mean, inv = ops.bn1d_train_fwd(lin, gamma, beta, running_mean, running_var, momentum, eps)
        else:
            inv = torch.rsqrt(running_var.detach() + eps)        # frozen statistics: plain affine map
            mean = running_mean.detach().clone()
            out = ops.linear(pool, ops.pack_conv(lw, lb, (gamma, beta, running_mean, running_var), bn_eps=eps))
        ctx.save_for_backward(*acts, pool, lin, mean, inv, w1, w2, w3, w0, lw, gamma)
        ctx.bn_train = bn_train
        ctx.x_shape = x.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        x0, y1, y2, y3, y4, pool, lin, mean, inv, w1, w2, w3, w0, lw, gamma = ctx.saved_tensors
        dout = dout.contiguous().to(F32)
        k = dout.shape[0]
        dlin, dgamma, dbeta = ops.bn1d_bwd(dout, lin, mean, inv, gamma, frozen=not ctx.bn_train)
        dlw = ops.conv_wgrad(pool.view(k, 1, 1, -1), dlin.view(k, 1, 1, -1), 1, 1).view(lw.shape)
        dlb = ops.colsum(dlin)
        dpool = ops.linear(dlin, ops.pack_conv_dgrad(lw))
        dy = ops.avgpool_relu_bwd(dpool, y4)
        acts = (x0, y1, y2, y3)
        weights = (w0, w1, w2, w3)
        dws, dbs = [None] * 4, [None] * 4
        dx = None
        for l in (3, 2, 1, 0):
            dws[l] = ops.conv_wgrad(acts[l], dy, 3, 3)
            dbs[l] = ops.colsum(dy)
            if l > 0:                                             # input gradient, masked by the ReLU of the producer
                dy = ops.conv2d(dy, ops.pack_conv_dgrad(weights[l], wino=False), relu=2, residual=acts[l])
            elif ctx.needs_input_grad[0]:
                dx = ops.nhwc_to_nchw(ops.conv2d(dy, ops.pack_conv_dgrad(weights[0], wino=False)))
        return (dx, dws[0], dbs[0], dws[1], dbs[1], dws[2], dbs[2], dws[3], dbs[3], dlw, dlb, dgamma, dbeta,
                None, None, None, None, None)


class PairLogitsFunction(torch.autograd.Function):
    """x5 = last((a_i - b_j)^2)   (ref models/match_head.py:73-74,161-162)."""

    @staticmethod
    def forward(ctx, a, b, w, bias):
        a, b = a.detach().contiguous(), b.detach().contiguous()
        ctx.save_for_backward(a, b, w)
        return ops.pair_logits(a, b, w, bias)

    @staticmethod
    def backward(ctx, g):
        a, b, w = ctx.saved_tensors
        da, db, dw, dbias = ops.pair_logits_bwd(a, b, w, g.contiguous())
        return da, db, dw, dbias


class NlbAttnPoolFunction(torch.autograd.Function):
    """Batched non-local block (sequences longer than one row) + softmax attention pooling
    (ref models/nlb.py:66-101, models/match_head.py:114-121).  seq time-major [T,S,256], lens int32 [S]."""

    @staticmethod
    def forward(ctx, seq, lens, use_nlb, theta_w, theta_b, phi_w, phi_b, g_w, g_b, cat_w, W_w, W_b, att_w, att_b):
        seq = seq.detach().contiguous()
        pk = ops.PackedNLB(
            w_proj_t=torch.cat([theta_w[:, :, 0], phi_w[:, :, 0], g_w[:, :, 0]], 0).detach().t().contiguous(),
            b_proj=torch.cat([theta_b, phi_b, g_b]).detach().contiguous(),
            w_cat=cat_w.detach().reshape(256).contiguous(),
            w_out_t=W_w.detach()[:, :, 0].t().contiguous(), b_out=W_b.detach().contiguous(),
            w_att=att_w.detach().reshape(256).contiguous(), b_att=att_b.detach().reshape(1).contiguous())
        t, s = seq.shape[0], seq.shape[1]
        out, _ = ops.nlb_attnpool(seq, s * 256, 256, lens, s, t, pk, use_nlb=use_nlb)
        ctx.save_for_backward(seq, lens)
        ctx.pk, ctx.use_nlb = pk, use_nlb
        return out

    @staticmethod
    def backward(ctx, dout):
        seq, lens = ctx.saved_tensors
        t, s = seq.shape[0], seq.shape[1]
        dseq, grads = ops.nlb_attnpool_bwd(seq, s * 256, 256, lens, s, t, ctx.pk, dout.contiguous(), ctx.use_nlb)
        return (dseq, None, None, *grads)


class NlbBlockFunction(torch.autograd.Function):
    """The non-local block alone, ``NONLocalBlock1D.forward`` called directly (ref models/nlb.py:66-101):
    x [b,256,t] -> z [b,256,t].  The block is applied whatever t is (the length-1 bypass is the aggregator's rule)."""

    @staticmethod
    def forward(ctx, x, theta_w, theta_b, phi_w, phi_b, g_w, g_b, cat_w, W_w, W_b):
        b, c, t = x.shape
        xt = ops.nchw_to_nhwc(x.detach().to(F32).contiguous().view(b, c, t))             # rows [b,t,256]
        dev = x.device
        pk = ops.PackedNLB(
            w_proj_t=torch.cat([theta_w[:, :, 0], phi_w[:, :, 0], g_w[:, :, 0]], 0).detach().t().contiguous(),
            b_proj=torch.cat([theta_b, phi_b, g_b]).detach().contiguous(),
            w_cat=cat_w.detach().reshape(256).contiguous(),
            w_out_t=W_w.detach()[:, :, 0].t().contiguous(), b_out=W_b.detach().contiguous(),
            w_att=torch.zeros(256, device=dev), b_att=torch.zeros(1, device=dev))
        lens = torch.full((b,), t, dtype=torch.int32, device=dev)
        _, _, z = ops.nlb_attnpool(xt, 256, t * 256, lens, b, t, pk, use_nlb=2, want_z=True)
        ctx.save_for_backward(xt, lens)
        ctx.pk = pk
        return ops.nhwc_to_nchw(z)

    @staticmethod
    def backward(ctx, dz):
        xt, lens = ctx.saved_tensors
        b, t, _ = xt.shape
        dzt = ops.nchw_to_nhwc(dz.contiguous().to(F32))                                    # [b,t,256]
        dxt, grads = ops.nlb_block_bwd(xt, 256, t * 256, lens, b, t, ctx.pk, dzt, 256, t * 256, use_nlb=2)
        return (ops.nhwc_to_nchw(dxt), *grads)


class WeightedCE2Function(torch.autograd.Function):
    """nn.CrossEntropyLoss(weight=[w0,w1]) over [n,2] logits (the criterion of every loss in the reference's
    models/match_head.py:213,257,367,386): weighted mean of -log softmax(x)[y]."""

    @staticmethod
    def forward(ctx, logits, target, weight):
        loss, dlogits = ops.ce2_fwd_bwd(logits.detach().contiguous(), target, weight)
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None, None
