"""Repeated A/B of the weights-stationary pointwise kernel against the implicit GEMM on ragged / small shapes, all three residual
modes (none, plain, FPN top-down), 30 runs each: any element that differs by more than 1e-4 is counted (a race shows up as a count
that changes from run to run).  usage: python tools/pw_stress.py"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seam_match_rcnn_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
bad_total = 0
for (n,h,w,c,k) in [(2,48,72,256,128),(2,48,72,256,256),(3,40,56,64,256),(8,100,100,256,256)]:
    x = torch.randn(n,h,w,c,device=dev)
    wt = torch.randn(k,c,1,1,device=dev)*0.05
    bias = torch.randn(k,device=dev)
    pc = ops.pack_conv(wt, bias)
    top = torch.randn(n,(h+1)//2,(w+1)//2,k,device=dev)
    res = torch.randn(n,h,w,k,device=dev)
    for it in range(30):
        for mode in (0,1,2):
            ops.SW=True
            y = ops.conv2d(x,pc) if mode==0 else ops.conv2d(x,pc,True,res) if mode==1 else ops.conv2d_topdown(x,pc,top)
            ops.SW=False
            y0 = ops.conv2d(x,pc) if mode==0 else ops.conv2d(x,pc,True,res) if mode==1 else ops.conv2d_topdown(x,pc,top)
            nb = int(((y-y0).abs() > 1e-4).sum())
            bad_total += nb
            if nb: print((n,h,w,c,k), "mode", mode, "iter", it, "bad", nb)
print("total bad", bad_total)
