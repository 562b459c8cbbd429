#!/usr/bin/env python3
"""Latency of the full drop-in forward (RPN proposals + box head + NMS + mask/match branches) on 10 frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import seam_match_rcnn_amd.synth as synth

dev = torch.device("cuda:0")
model, sd = bench.build_model(dev)
frames = list(torch.from_numpy(synth.frames(0, 10, 800, 800)).to(dev).unbind(0))

def stage_times():
    t = {}
    def tick(name, t0):
        torch.cuda.synchronize(); t[name] = t.get(name, 0) + time.perf_counter() - t0
    with torch.no_grad():
        t0 = time.perf_counter(); feats, sizes, orig, padded = model.extract_features(frames); tick("backbone+fpn", t0)
        t0 = time.perf_counter(); props = model.rpn(feats, sizes, padded); tick("rpn (head+filter)", t0)
        t0 = time.perf_counter(); res = model.roi_heads.detect(feats, props, sizes); tick("box branch + postprocess", t0)
        t0 = time.perf_counter(); res = model.roi_heads.match_branch(feats, res, sizes); tick("mask+match branch", t0)
        t0 = time.perf_counter(); res = model.postprocess(res, sizes, orig); tick("postprocess (mask paste)", t0)
    return t, res

stage_times()
t, res = stage_times()
print({k: round(v * 1e3, 2) for k, v in t.items()}, "total ms", round(sum(t.values()) * 1e3, 2))
print("detections per image:", [len(r["scores"]) for r in res])
