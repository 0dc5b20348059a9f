"""Experiment: one step of B = 16 pairs as two concurrent halves on two streams (own module instance / workspace each)
against the single-stream step."""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vcrnet_amd  # noqa
from vcrnet_amd import synth


def main():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
    from test_hip_forward import build_net
    net, _ = build_net()
    src, tgt, *_ = synth.make_batch(1234, 16, 1024)
    s, t = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    nets = [net, copy.deepcopy(net)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    halves = [(s[:8].contiguous(), t[:8].contiguous()), (s[8:].contiguous(), t[8:].contiguous())]

    def step_single():
        with torch.no_grad():
            return net(s, t)

    def step_two():
        cur = torch.cuda.current_stream()
        outs = []
        for n, st, (a, b) in zip(nets, streams, halves):
            st.wait_stream(cur)
            with torch.cuda.stream(st), torch.no_grad():
                outs.append(n(a, b))
        for st in streams:
            cur.wait_stream(st)
        return outs

    def step_half():
        with torch.no_grad():
            return net(*halves[0])

    for name, fn, pairs in (("single B=16", step_single, 16), ("two streams 2 x B=8", step_two, 16), ("single B=8", step_half, 8),
                            ("single B=16", step_single, 16), ("two streams 2 x B=8", step_two, 16)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        print(f"{name:24s} {dt * 1e3:7.3f} ms/step  {pairs / dt:8.1f} pairs/s")
    o1 = step_single(); o2 = step_two()
    print("max|dR| two-stream vs single:", (torch.cat((o2[0][2], o2[1][2])) - o1[2]).abs().max().item())


main()
