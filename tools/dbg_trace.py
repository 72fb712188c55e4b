"""First evaluated leaf at which the GPU search of one root departs from the oracle (plane hashes per batch)."""
import sys, zlib, numpy as np, torch
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import hivemind_amd as hm, oracle_py as O
from test_gpu_search import _hash_eval_gpu, _roots
hm.init(0)
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 400
gsel = int(sys.argv[2]) if len(sys.argv) > 2 else 23
G = 24
roots = _roots(G, 77 + nodes); roots[0] = O.Board().compact(0, False)[0]
eng = hm.SearchEngine(G, 1700); eng.set_games(roots); eng.begin_search(nodes)
rows = torch.zeros(G, dtype=torch.int32, device='cuda')
gpu = []          # list of batches: list of crc per evaluated row of game gsel
prev_rows = 0
for it in range(3000):
    planes = eng.collect(rows_next=rows)          # planes = CURRENT buffer (collected one iteration ago)
    h = planes.cpu().numpy().view(np.uint16).reshape(G, 8, 4736)
    if prev_rows:
        gpu.append([zlib.crc32(h[gsel, k].tobytes()) for k in range(prev_rows)])
    prev_rows = int(rows[gsel].item())
    if eng.process(*_hash_eval_gpu(planes)) == 0:
        break
ora = []
def cb(a):
    n = a.shape[0]
    ora.append([zlib.crc32(np.ascontiguousarray(a[k]).tobytes()) for k in range(n)])
    return O.hash_evaluator(np.ascontiguousarray(a))
b = O.Board(); b.from_compact(roots[gsel:gsel + 1]); s = O.Search(1, 1)
s.set_evaluator(cb)
s.run(b, int(roots['team'][gsel]), bool(roots['time_adv'][gsel]), nodes)
print('batches gpu', len(gpu), 'oracle', len(ora))
for i, (x, y) in enumerate(zip(gpu, ora)):
    if x != y:
        print('first differing batch', i, 'sizes', len(x), len(y))
        print(' gpu', x); print(' ora', y)
        for j in range(max(0, i - 2), i): print(' prev batch', j, 'size', len(gpu[j]), gpu[j] == ora[j])
        break
else:
    print('no difference in common prefix')
