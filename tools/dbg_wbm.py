import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import hivemind_amd as hm, oracle_py as O
from hivemind_amd import _lib
from test_gpu_search import _hash_eval_gpu, _roots
hm.init(0)
G = 24; nodes = 400; gsel = 23
roots = _roots(G, 77 + nodes); roots[0] = O.Board().compact(0, False)[0]
_lib.lib.hm_sp_trace_select.argtypes = [C.c_int]; _lib.lib.hm_sp_trace.argtypes = [C.c_void_p, C.c_int]
_lib.lib.hm_sp_trace_select(gsel)
eng = hm.SearchEngine(G, 1700); eng.set_games(roots); eng.begin_search(nodes); eng.run(_hash_eval_gpu)
st = eng.root_stats()
print('info', st['info'][gsel].tolist())
buf = np.zeros(4096, np.uint64)
n = _lib.lib.hm_sp_trace(buf.ctypes.data, 4096)
print('wbm false returns with mates available:', n)
for v in buf[:min(n, 40)]:
    v = int(v); print('  r', v >> 48, 'nr', (v >> 40) & 255, 'nm', (v >> 32) & 255, 'drawAfter', (v >> 31) & 1, 'reply', hex(v & 0x7fffffff))
