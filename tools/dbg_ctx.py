"""Compare the collect_batch event logs (oracle Search::ctxTrace vs GPU -DHM_SEARCH_TRACE build) of one root."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRACE_LIB = os.path.join(ROOT, "hivemind_amd", "csrc", "libhivemind_amd_trace.so")
if "--build" in sys.argv:      # python tools/dbg_ctx.py --build   (here), then on the GPU: python tools/dbg_ctx.py [nodes] [game]
    src = [os.path.join(ROOT, "hivemind_amd", "csrc", f) for f in ("hm_kernels.hip", "hm_search.hip", "hm_selfplay.hip", "hm_net.hip")]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-strict-aliasing", "-fPIC", "-shared",
                           "-DHM_SEARCH_TRACE", "-I", os.path.join(ROOT, "include"), *src, "-o", TRACE_LIB])
    print("built", TRACE_LIB); sys.exit(0)
os.environ.setdefault("HIVEMIND_AMD_LIB", TRACE_LIB)
import numpy as np, torch
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hivemind_amd as hm, oracle_py as O
from hivemind_amd import _lib
from test_gpu_search import _hash_eval_gpu, _roots
hm.init(0)
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 400
gsel = int(sys.argv[2]) if len(sys.argv) > 2 else 23
G = 24
roots = _roots(G, 77 + nodes); roots[0] = O.Board().compact(0, False)[0]
_lib.lib.hm_sp_trace_select.argtypes = [C.c_int]; _lib.lib.hm_sp_trace.argtypes = [C.c_void_p, C.c_int]
_lib.lib.hm_sp_trace_select(gsel)
eng = hm.SearchEngine(G, 1700); eng.set_games(roots); eng.begin_search(nodes); eng.run(_hash_eval_gpu)
buf = np.zeros(65536, np.uint64)
n = _lib.lib.hm_sp_trace(buf.ctypes.data, 65536)
v = buf[:n]
gpu = np.stack([(v >> np.uint64(32)), (v >> np.uint64(24)) & np.uint64(0xff), (v >> np.uint64(8)) & np.uint64(0xffff), v & np.uint64(0xff)], axis=1).astype(np.int64)
b = O.Board(); b.from_compact(roots[gsel:gsel + 1]); s = O.Search(1, 1)
s.run(b, int(roots['team'][gsel]), bool(roots['time_adv'][gsel]), nodes)
ora = s.ctx_trace()
full = gpu
gpu = gpu[gpu[:, 1] < 10]
print('events gpu', len(gpu), 'oracle', len(ora))
m = min(len(gpu), len(ora))
d = np.nonzero((gpu[:m] != ora[:m]).any(axis=1))[0]
k = int(d[0]) if len(d) else int(sys.argv[3]) if len(sys.argv) > 3 else 0
if len(d) == 0: print('identical prefix; showing event', k)
print('first differing event', k)
for i in range(max(0, k - 10), min(m, k + 6)):
    print(i, 'gpu', gpu[i].tolist(), 'ora', ora[i].tolist(), '<--' if i == k else '')

seq = gpu[k][0]
print('internal events of collect', seq)
for r in full[full[:, 0] == seq]:
    if r[1] >= 10: print('   ', r.tolist())
