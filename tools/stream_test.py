#!/usr/bin/env python3
"""Does splitting the 10-frame batch over 2 (or more) HIP streams hide tile-quantisation tails? (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import seam_match_rcnn_amd.synth as synth

dev = torch.device("cuda:0")
model, sd = bench.build_model(dev)
frames = list(torch.from_numpy(synth.frames(0, 10, 800, 800)).to(dev).unbind(0))

def run(parts, streams):
    outs = []
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for fr, s in zip(parts, streams):
        with torch.cuda.stream(s):
            feats, sizes, orig, padded = model.extract_features(fr)
            rpn = model.rpn.head(list(feats.values()))
            outs.append((feats, rpn))
    for s in streams: cur.wait_stream(s)
    return outs

with torch.no_grad():
    for nsplit in (1, 2, 5):
        streams = [torch.cuda.Stream() for _ in range(nsplit)]
        parts = [frames[i::nsplit] for i in range(nsplit)]
        for _ in range(2): run(parts, streams)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): run(parts, streams)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        print(f"{nsplit} stream(s): backbone+FPN+RPN head for 10 frames = {dt*1e3:.2f} ms")
