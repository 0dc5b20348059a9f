import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import vcrnet_amd  # noqa
from vcrnet_amd import synth
from test_hip_forward import build_net
net, _ = build_net()
src, tgt, *_ = synth.make_batch(1234, 16, 1024)
s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
P = lambda *a: print(*a, flush=True)
with torch.no_grad():
    ref = net(s, t); torch.cuda.synchronize()
    g, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(st):
        net(s, t); st.synchronize()
        with torch.cuda.graph(g, stream=st):
            out = net(s, t)
        torch.cuda.synchronize(); P("captured")
        for name, fn in (("graph replay", g.replay), ("eager", lambda: net(s, t)), ("graph replay", g.replay), ("eager", lambda: net(s, t))):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200): fn()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
            P(f"{name:14s} {dt*1e3:.3f} ms  {16/dt:.1f} pairs/s")
        P(torch.equal(out[2], ref[2]))
