"""Time of the weight re-pack after an optimiser step (E6D2: six layers, H = 1024): python tools/pack_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgedict_amd import config, encoder_stack as es
H = 1024
ws = []
for l in range(6):
    I = 240 if l == 0 else H
    ws.append([torch.randn(4 * H, I).cuda(), torch.randn(4 * H, H).cuda(), torch.randn(4 * H).cuda(), torch.randn(4 * H).cuda()])
for it in range(5):
    config.bump_param_epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for w in ws:
        es.packed_weights(*w)
    e1.record()
    torch.cuda.synchronize()
    print("re-pack of 6 layers: %.1f us GPU, %.1f us host+GPU" % (1e3 * e0.elapsed_time(e1), 1e6 * (time.perf_counter() - t0)))
