import sys, numpy as np, torch
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import hivemind_amd as hm, oracle_py as O
from test_gpu_search import _hash_eval_gpu, _roots
hm.init(0)
nodes=int(sys.argv[1]) if len(sys.argv)>1 else 400
G=24
roots=_roots(G,77+nodes); roots[0]=O.Board().compact(0,False)[0]
eng=hm.SearchEngine(G,1700); eng.set_games(roots); eng.begin_search(nodes); eng.run(_hash_eval_gpu)
st=eng.root_stats()
bad=0
for g in range(G):
    b=O.Board(); b.from_compact(roots[g:g+1]); s=O.Search(1,1)
    ok=s.run(b,int(roots['team'][g]),bool(roots['time_adv'][g]),nodes)
    if not ok: print(g,'noaction',st['info'][g][0]); continue
    e=s.edges(); n=st['counts'][g]; oi=s.info(); info=st['info'][g]
    keys=('move_a','move_b','visits','prior','q')
    diffs=[k for k in keys if n!=len(e['visits']) or not np.array_equal(st[k][g,:n],e[k])]
    if diffs or info[1]!=oi['nodes'] or info[5]!=oi['node_count']:
        bad+=1
        print('GAME',g,'diffs',diffs,'n',n,len(e['visits']),'info',info.tolist(),oi)
        for k in diffs[:3]:
            m=min(n,len(e[k]))
            idx=[i for i in range(m) if st[k][g,i]!=e[k][i]][:6]
            print('  ',k,[(i,st[k][g,i],e[k][i]) for i in idx])
print('bad',bad,'of',G)
