"""Non-local block, mirror of reference ``models/nlb.py``.

Only the configuration the reference ever instantiates on the hot path is implemented:
``NONLocalBlock1D(in_channels=256, sub_sample=False, bn_layer=False)``
(ctor call: ref models/match_head.py:87).  The 2D/3D, sub-sampled and BatchNorm variants
are never constructed outside the reference's ``__main__`` smoke test (models/nlb.py:128-145)
and raise ``NotImplementedError`` here.

The concatenation-form block (ref models/nlb.py:66-101) runs as ONE kernel launch for a
whole batch of sequences (``seam_nlb_attnpool_f32``), never materialising [b,256,T,T].
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops


class _ForwardOnly(torch.autograd.Function):
    """Identity on a forward-kernel result that makes it a differentiable-looking output: backward() raises."""

    @staticmethod
    def forward(ctx, z, t, *deps):
        ctx.t = int(t)
        return z.clone()

    @staticmethod
    def backward(ctx, grad):
        raise NotImplementedError(f"NONLocalBlock1D: the grad-enabled pass supports sequences of <= 64 frames, got {ctx.t} "
                                  "(seam_nlb_block_bwd_f32 keeps a sequence in LDS); the forward result is valid")


class _NonLocalBlockND(nn.Module):
    def __init__(self, in_channels, inter_channels=None, dimension=3, sub_sample=True, bn_layer=True):
        super().__init__()
        assert dimension in [1, 2, 3]
        if dimension != 1 or sub_sample or bn_layer:
            raise NotImplementedError(
                "only NONLocalBlock1D(sub_sample=False, bn_layer=False) is on the SEAM hot path "
                "(reference models/match_head.py:87)")
        self.dimension = dimension
        self.sub_sample = sub_sample
        self.in_channels = in_channels
        self.inter_channels = inter_channels
        if self.inter_channels is None:
            self.inter_channels = max(in_channels // 2, 1)
        if in_channels != 256 or self.inter_channels != 128:
            raise NotImplementedError("the HIP kernel is specialised for 256 -> 128 channels")
        ic = self.inter_channels
        # parameter containers keep the reference's state-dict keys
        self.g = nn.Conv1d(in_channels, ic, 1)
        self.W = nn.Conv1d(ic, in_channels, 1)
        nn.init.constant_(self.W.weight, 0)         # ref models/nlb.py:48-49
        nn.init.constant_(self.W.bias, 0)
        self.theta = nn.Conv1d(in_channels, ic, 1)
        self.phi = nn.Conv1d(in_channels, ic, 1)
        self.concat_project = nn.Sequential(nn.Conv2d(ic * 2, 1, 1, 1, 0, bias=False), nn.ReLU())
        self._pk = None
        self._pk_key = None

    # ---- packed weights (rebuilt when a parameter is replaced or modified in place) -------------
    def packed(self, scorer: "nn.Linear | None" = None) -> ops.PackedNLB:
        ps = [self.theta.weight, self.theta.bias, self.phi.weight, self.phi.bias, self.g.weight, self.g.bias,
              self.W.weight, self.W.bias, self.concat_project[0].weight]
        if scorer is not None:
            ps += [scorer.weight, scorer.bias]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._pk is None or key != self._pk_key:
            with torch.no_grad():
                dev = self.W.weight.device
                if scorer is not None:
                    wa, ba = scorer.weight.reshape(256).contiguous(), scorer.bias.reshape(1).contiguous()
                else:
                    wa, ba = torch.zeros(256, device=dev), torch.zeros(1, device=dev)
                self._pk = ops.PackedNLB(
                    w_proj_t=torch.cat([self.theta.weight[:, :, 0], self.phi.weight[:, :, 0], self.g.weight[:, :, 0]], 0)
                    .t().contiguous(),
                    b_proj=torch.cat([self.theta.bias, self.phi.bias, self.g.bias]).contiguous(),
                    w_cat=self.concat_project[0].weight.reshape(256).contiguous(),
                    w_out_t=self.W.weight[:, :, 0].t().contiguous(),
                    b_out=self.W.bias.contiguous(), w_att=wa, b_att=ba)
            self._pk_key = key
        return self._pk

    def forward(self, x):
        """x: (b, 256, t) -> z: (b, 256, t)      (ref models/nlb.py:66-101)"""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            # grad-enabled direct call: the same forward kernel with the block's own backward behind it (t <= 64)
            if x.shape[-1] > 64:
                # The backward kernel keeps a sequence in LDS (<= 64 frames).  The reference module accepts such a call, and an
                # eval-mode call without torch.no_grad() is one (parameters require grad by default): run the forward kernel and
                # hand out a result whose BACKWARD raises, with a clear message, if anyone asks for it.
                return _ForwardOnly.apply(self._forward_nograd(x), x.shape[-1], x, *self.parameters())
            from ..autograd import NlbBlockFunction
            return NlbBlockFunction.apply(x, self.theta.weight, self.theta.bias, self.phi.weight, self.phi.bias, self.g.weight,
                                          self.g.bias, self.concat_project[0].weight, self.W.weight, self.W.bias)
        return self._forward_nograd(x)

    def _forward_nograd(self, x):
        b, c, t = x.shape
        xt = ops.nchw_to_nhwc(x.detach().contiguous().view(b, c, t))            # [b,t,256]
        lens = torch.full((b,), t, dtype=torch.int32, device=x.device)
        # use_nlb=2: apply the block even for t == 1 (the length-1 bypass is the CALLER's rule,
        # ref models/match_head.py:115-117, not this module's)
        _, _, z = ops.nlb_attnpool(xt, 256, t * 256, lens, b, t, self.packed(), use_nlb=2, want_z=True)
        return ops.nhwc_to_nchw(z)


class NONLocalBlock1D(_NonLocalBlockND):
    def __init__(self, in_channels, inter_channels=None, sub_sample=True, bn_layer=True):
        super().__init__(in_channels, inter_channels=inter_channels, dimension=1,
                         sub_sample=sub_sample, bn_layer=bn_layer)


class NONLocalBlock2D(_NonLocalBlockND):
    def __init__(self, in_channels, inter_channels=None, sub_sample=True, bn_layer=True):
        super().__init__(in_channels, inter_channels=inter_channels, dimension=2,
                         sub_sample=sub_sample, bn_layer=bn_layer)


class NONLocalBlock3D(_NonLocalBlockND):
    def __init__(self, in_channels, inter_channels=None, sub_sample=True, bn_layer=True):
        super().__init__(in_channels, inter_channels=inter_channels, dimension=3,
                         sub_sample=sub_sample, bn_layer=bn_layer)
