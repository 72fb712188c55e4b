"""4-wave (narrow_position4) against 8-wave (narrow_position) fused forward: outputs must be bit-identical; time per launch of both.
HM_NET_WAVES is read when the network handle is created.  python tools/net_waves_compare.py [rows ...]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hivemind_amd as hm
from hivemind_amd import net as N
hm.init(0)
rows = [int(a) for a in sys.argv[1:]] or [1, 64, 256, 512]
out = {}
for name, mk in (("small", N.rise_v3_small), ("full", N.rise_v33)):
    torch.manual_seed(0)
    model = mk().eval()
    fl = N.flops_per_position(model)
    nets = {}
    for w in (8, 4):
        os.environ["HM_NET_WAVES"] = str(w)
        nets[w] = N.FusedNet(model)
    os.environ.pop("HM_NET_WAVES")
    torch.manual_seed(1)
    x = (torch.rand((max(rows), 74, 8, 8), device="cuda") < 0.2).half()
    ref = [t.clone() for t in nets[8](x)]
    got = [t.clone() for t in nets[4](x)]
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(ref, got))
    print(name, "bit-identical:", same, flush=True)
    out[name] = dict(bit_identical=same)
    for n in rows:
        xs = x[:n].contiguous()
        for w in (8, 4):
            f = nets[w]
            for _ in range(3): f(xs)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = 30; s.record()
            for _ in range(it): f(xs)
            e.record(); torch.cuda.synchronize()
            ms = s.elapsed_time(e) / it
            out[name][f"{n}rows_{w}waves_ms"] = round(ms, 4)
            print(name, n, "rows", w, "waves", round(ms, 4), "ms", round(n * fl / (ms * 1e-3) / 1e12, 1), "TFLOP/s", flush=True)
    assert same
json.dump(out, open("gpurun_out/net_waves_compare.json", "w"), indent=1)
