"""Size of documented deviation (1): how often does a WHOLE search differ between the oracle's two modes?

  mode (tie 0, exp 0) = the reference's own std::sort / std::priority_queue tie order and glibc std::exp
                        (joint_action.h:234-241, utils.h:143-166)
  mode (tie 1, exp 1) = what the GPU implements: strict total order (prior desc, index asc), portable_expf

Both pass the 48 transcribed reference gtest cases; this reports the exact-match rate of root visit vectors over N roots at a
node budget under the shared hash evaluator (noise off), attributes mismatches to the tie rule or the exp flavour by running the
two mixed modes too, and names the first divergence of a few mismatching roots.  CPU only (oracle/liboracle.so).

  python tools/deviation_modes.py [--roots 128] [--nodes 400] [--evaluator hash|net] [--out profiles/r04_deviation_modes.json]

The hash stand-in quantises its logits, so equal priors — the only place the tie rule can act — are far more frequent under it
than under a network; --evaluator net repeats the count with the bench's random-init RISEv3-small on the host cores.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O  # noqa: E402


def roots_for(n, seed=4242):
    """the root set of tests/test_gpu_search.py's 128-root sweep (late-game positions included) + the start position"""
    r = O.random_positions(seed, n * 11, 160)[::11][:n].copy()
    r[0] = O.Board().compact(0, False)[0]
    return r


_EVAL = None        # None: the oracle's built-in hash evaluator; else fn(planes u16) -> 5 u16 arrays


def net_evaluator():
    """random-init RISEv3-small (torch.manual_seed(0), the bench's network) in fp32 on the host cores, heads rounded to fp16"""
    import torch
    sys.path.insert(0, ROOT)
    from hivemind_amd import net as N
    torch.manual_seed(0)
    model = N.rise_v3_small().float().eval()
    torch.set_num_threads(min(8, os.cpu_count() or 1))

    def ev(planes_u16):
        x = torch.from_numpy(planes_u16.view(np.float16).astype(np.float32)).reshape(-1, 74, 8, 8)
        with torch.no_grad():
            v, a, b, w, m = model(x)
        h = lambda t: t.to(torch.float16).contiguous().numpy().view(np.uint16)
        return h(v.reshape(-1)), h(a), h(b), h(w), h(m.reshape(-1))
    return ev


def search(root, tie, exp, nodes):
    b = O.Board()
    b.from_compact(root[None] if root.ndim == 0 else root)
    s = O.Search(tie, exp)
    if _EVAL is not None:
        s.set_evaluator(_EVAL)
    ok = s.run(b, int(root["team"]), bool(root["time_adv"]), nodes)
    if not ok:
        return None
    e = s.edges()
    return dict(move_a=e["move_a"].copy(), move_b=e["move_b"].copy(), visits=e["visits"].copy(), prior=e["prior"].copy(),
                q=e["q"].copy(), best=s.best_move(), nodes=s.info()["nodes"])


def same_visits(a, b):
    if a is None or b is None:
        return a is None and b is None
    return (len(a["visits"]) == len(b["visits"]) and np.array_equal(a["move_a"], b["move_a"]) and np.array_equal(a["move_b"], b["move_b"])
            and np.array_equal(a["visits"], b["visits"]))


def first_divergence(a, b):
    """why two searches of one root differ: edge order (tie rule), a prior's bits (exp flavour), or only visit counts downstream"""
    n = min(len(a["visits"]), len(b["visits"]))
    for i in range(n):
        if a["move_a"][i] != b["move_a"][i] or a["move_b"][i] != b["move_b"][i]:
            tied = bool(i > 0 and a["prior"][i] == a["prior"][i - 1]) or bool(i + 1 < n and a["prior"][i] == a["prior"][i + 1])
            return dict(kind="root edge order", edge=i, equal_priors_adjacent=tied)
    for i in range(n):
        if a["prior"][i].tobytes() != b["prior"][i].tobytes():
            return dict(kind="root prior bits", edge=i, ulp=int(abs(int(a["prior"][i:i + 1].view(np.int32)[0]) - int(b["prior"][i:i + 1].view(np.int32)[0]))))
    for i in range(n):
        if a["visits"][i] != b["visits"][i]:
            return dict(kind="visits only (divergence below the root)", edge=i, visits=[int(a["visits"][i]), int(b["visits"][i])])
    return dict(kind="edge count", counts=[len(a["visits"]), len(b["visits"])])


def run(n_roots, nodes):
    roots = roots_for(n_roots)
    modes = {"ref(0,0)": (0, 0), "gpu(1,1)": (1, 1), "tie_only(1,0)": (1, 0), "exp_only(0,1)": (0, 1)}
    res = {k: [search(roots[g], t, e, nodes) for g in range(n_roots)] for k, (t, e) in modes.items()}
    base, gpu = res["ref(0,0)"], res["gpu(1,1)"]
    match = [same_visits(base[g], gpu[g]) for g in range(n_roots)]
    best_same = [(base[g] is None and gpu[g] is None) or (base[g] is not None and gpu[g] is not None and base[g]["best"] == gpu[g]["best"])
                 for g in range(n_roots)]
    mism = [g for g in range(n_roots) if not match[g]]
    out = dict(roots=n_roots, nodes=nodes,
               evaluator=("hash evaluator (oracle/search.hpp)" if _EVAL is None else "random-init RISEv3-small, torch CPU fp32 -> fp16 heads") + ", noise off",
               exact_visit_vector_matches=int(sum(match)), exact_match_rate=float(sum(match)) / n_roots,
               best_move_matches=int(sum(best_same)),
               mismatches_caused_by=dict(
                   tie_rule_alone=int(sum(1 for g in mism if not same_visits(base[g], res["tie_only(1,0)"][g]))),
                   exp_flavour_alone=int(sum(1 for g in mism if not same_visits(base[g], res["exp_only(0,1)"][g])))),
               mismatching_roots=mism,
               first_divergence={str(g): first_divergence(base[g], gpu[g]) for g in mism[:8] if base[g] is not None and gpu[g] is not None})
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--roots", type=int, default=128)
    ap.add_argument("--nodes", type=int, default=400)
    ap.add_argument("--out", default="")
    ap.add_argument("--evaluator", choices=("hash", "net"), default="hash")
    a = ap.parse_args()
    if a.evaluator == "net":
        _EVAL = net_evaluator()
    r = run(a.roots, a.nodes)
    s = json.dumps(r, indent=1)
    print(s)
    if a.out:
        with open(a.out, "w") as f:
            f.write(s + "\n")
