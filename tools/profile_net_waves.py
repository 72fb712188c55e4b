"""phase stamps (s_memtime of workgroup 0) of the fused forward, 8-wave against 4-wave form: python tools/profile_net_waves.py [small|full]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, hivemind_amd as hm
from hivemind_amd import net as N
hm.init(0)
name = sys.argv[1] if len(sys.argv) > 1 else "small"
torch.manual_seed(0)
model = (N.rise_v3_small if name == "small" else N.rise_v33)().eval()
x = torch.rand((1, 74, 8, 8), device="cuda").half()
res = {}
for w in (8, 4):
    os.environ["HM_NET_WAVES"] = str(w)
    f = N.FusedNet(model)
    for _ in range(3): f(x)
    v, a, b, ww, m = f._buffers(1)
    best = None
    for rep in range(5):
        st = torch.zeros(256, dtype=torch.int64, device="cuda")
        hm.check(hm.lib.hm_net_profile(f.handle, x.data_ptr(), 1, v.data_ptr(), a.data_ptr(), b.data_ptr(), ww.data_ptr(), m.data_ptr(), None, st.data_ptr()))
        torch.cuda.synchronize()
        t = st.cpu().numpy(); t = t[t > 0]
        d = np.diff(t)
        if best is None or d.sum() < best.sum(): best = d
    res[w] = best
n = min(len(res[8]), len(res[4]))
print(name, "total cycles: 8 waves", int(res[8].sum()), " 4 waves", int(res[4].sum()))
for i in range(n):
    print(f"{i:3d} {int(res[8][i]):8d} {int(res[4][i]):8d}  {res[4][i] / max(1, res[8][i]):.2f}")
