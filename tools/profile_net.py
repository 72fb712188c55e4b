import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, hivemind_amd as hm
from hivemind_amd import net as N
hm.init(0)
torch.manual_seed(0)
model = N.rise_v3_small() if (len(sys.argv) < 2 or sys.argv[1] == "small") else N.rise_v33()
f = N.FusedNet(model)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
x = torch.rand((n, 74, 8, 8), device="cuda").half()
for _ in range(3): f(x)
v, a, b, w, m = f._buffers(n)
st = torch.zeros(256, dtype=torch.int64, device="cuda")
hm.check(hm.lib.hm_net_profile(f.handle, x.data_ptr(), n, v.data_ptr(), a.data_ptr(), b.data_ptr(), w.data_ptr(), m.data_ptr(), None, st.data_ptr()))
torch.cuda.synchronize()
t = st.cpu().numpy()
t = t[t > 0]
d = np.diff(t)
print("stamps", len(t), "total cycles", int(t[-1] - t[0]), "(100 MHz ticks? see s_memtime: shader clock)")
print("deltas:", d.tolist())
