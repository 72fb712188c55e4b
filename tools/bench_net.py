import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hivemind_amd as hm
from hivemind_amd import net as N
hm.init(0)
torch.manual_seed(0)
out = {}
for name, mk in (("small", N.rise_v3_small), ("full", N.rise_v33)):
    model = mk()
    fl = N.flops_per_position(model)
    fused = N.FusedNet(model)
    for n in (64, 512, 2048, 8192):
        x = torch.rand((n, 74, 8, 8), device="cuda").half()
        inf = N.InferenceNet(model).capture(n)
        res = {}
        for label, f in (("fused", fused), ("torch_graph", inf)):
            for _ in range(3): f(x)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 20
            s.record()
            for _ in range(iters): f(x)
            e.record(); torch.cuda.synchronize()
            ms = s.elapsed_time(e) / iters
            res[label] = dict(ms=ms, tflops=n * fl / (ms * 1e-3) / 1e12)
        out[f"{name}_{n}"] = res
        print(name, n, json.dumps(res), flush=True)
