import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import hivemind_amd as hm
import oracle_py as O
from hivemind_amd import net as N
hm.init(0)
torch.manual_seed(0)
net = N.FusedNet(N.rise_v3_small())
G = 8
roots = O.random_positions(900 + 24, 24 * 13, 140)[::13][:G].copy()
roots[0] = O.Board().compact(0, False)[0]
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def run(persist, cap=421):
    eng = hm.SearchEngine(G, cap)
    eng.set_games(roots)
    eng.begin_search(nodes, None, 0.0, 0.0)
    if persist: eng.search_persistent(net)
    else: eng.run(net)
    st = eng.root_stats()
    eng.close()
    return st
want = run(False)
for label, env, cap in (("mode0 lds", {}, 421), ("mode1 inplace", {"HM_SEARCH_NO_LDS_NODES": "1"}, 421), ("mode2 mg3", {"HM_SEARCH_GAMES_PER_WG": "3"}, 421)):
    for k, v in env.items(): os.environ[k] = v
    got = run(True, cap)
    for k in env: os.environ.pop(k)
    for g in range(G):
        n = want["counts"][g]
        same_n = n == got["counts"][g]
        m = min(n, got["counts"][g])
        pri = np.array_equal(want["prior"][g, :m].view(np.uint32), got["prior"][g, :m].view(np.uint32))
        vis = np.array_equal(want["visits"][g, :m], got["visits"][g, :m])
        q = np.array_equal(want["q"][g, :m].view(np.uint32), got["q"][g, :m].view(np.uint32))
        mv = np.array_equal(want["move_a"][g, :m], got["move_a"][g, :m]) and np.array_equal(want["move_b"][g, :m], got["move_b"][g, :m])
        print(label, "game", g, "edges", n, got["counts"][g], "moves", mv, "priors", pri, "visits", vis, "q", q, "info", want["info"][g][:6].tolist(), got["info"][g][:6].tolist(), flush=True)
        if not pri and g == 0:
            print("  want prior", want["prior"][g, :6], "got", got["prior"][g, :6])
