import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hivemind_amd as hm
from hivemind_amd import net as N
hm.init(0)
torch.manual_seed(0)
for name, mk in (("small", N.rise_v3_small), ("full", N.rise_v33)):
    model = mk(); fl = N.flops_per_position(model); fused = N.FusedNet(model)
    for n in [int(a) for a in sys.argv[1:]] or (64, 160, 512, 4096):
        x = torch.rand((n, 74, 8, 8), device="cuda").half()
        for _ in range(3): fused(x)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 20; s.record()
        for _ in range(it): fused(x)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / it
        print(name, n, round(ms, 4), "ms", round(n * fl / (ms * 1e-3) / 1e12, 1), "TFLOP/s", flush=True)
