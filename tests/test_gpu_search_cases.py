"""GPU replay of the reference's gtest known answers that the product exposes directly
(tests/golden/search_cases.json, cases tagged "gpu"): classify_terminal_position incl. the waiting-board mate
rule with expected endInPly (engine/tests/test_move_gen.cc:510-596) on positions reached through real pushes
(game history on the device), the masked softmax's robustness rules (:643-671) through the raw-policy kernel, and the
last-move planes after a joint make (:306-317).  Positions travel as compact boards; pushes go through hm_sp_apply."""
import numpy as np
import pytest
import torch

import oracle_py as O
import search_lab as SL

pytestmark = pytest.mark.gpu
CASES = [c for c in SL.load_cases() if c.get("gpu") and c["gpu"] != "tree_reuse"]      # tree reuse: its own test below (trees come from searches)


class GpuRunner(SL.Runner):
    """Runs the scripted case on the oracle Board for bookkeeping (move lookup), and every classify / softmax /
    planes query on the GPU engine, whose game slot 0 mirrors the scripted board push by push."""

    def __init__(self, hm):
        super().__init__(1, 1)
        self.hm = hm
        self.eng = hm.SearchEngine(1, 64)
        self.synced = None

    def close(self):
        self.eng.close()
        super().close()

    def step(self, s):
        op = s["op"]
        if op in ("set_fen", "set", "board"):
            super().step(s)
            name = s.get("board", s.get("as"))
            self.eng.set_games(self.boards[name].compact(0, False))          # Board::set / set_fen restart the history
            self.synced = name
        elif op == "push_uci":
            bd = self.boards[s["board"]]
            m = SL.L.ora_board_uci_to_move(bd.h, s["which"], s["uci"].encode())
            assert m, s
            super().step(s)
            self.eng.apply([m if s["which"] == 0 else 0], [m if s["which"] == 1 else 0])
        elif op == "make_moves":
            a, b = self.move(s["a"], s["board"]), self.move(s["b"], s["board"])
            super().step(s)
            self.eng.apply([a], [b])
        elif op in ("unmake_moves",):
            super().step(s)
            self.synced = None                                                 # copy-make engine: nothing to unmake
        elif op == "planes":
            if self.synced is None:
                return
            boards, _ = self.eng.game_state()
            boards["team"][0], boards["time_adv"][0] = s["team"], int(s["adv"])
            p = self.hm.board_to_planes(self.hm.to_device(boards), "f32").cpu().numpy().reshape(74, 64)
            for ch, sq in s.get("ones", []):
                assert p[ch, sq] == 1.0, (s, ch, sq)
            # and the whole tensor equals the oracle's for the mirrored board
            want = O.planes(self.boards[s["board"]].compact(s["team"], s["adv"]), "f32")[0].reshape(74, 64)
            assert np.array_equal(p, want)
        elif op == "expect_board" and s["what"] in ("is_draw",):
            super().step(s)
            got = self.eng.classify(0, 0, 0, s["ply"])[0]
            assert bool(got[1]) == s["eq"], (s, got)
        else:
            super().step(s)

    def classify(self, bd, team, root_team, root_adv, ply):
        got = int(self.eng.classify(team, root_team, int(root_adv), ply)[0, 0])
        assert got == super().classify(bd, team, root_team, root_adv, ply), "GPU and oracle disagree"
        return got

    def normalized_probability(self, policy_f32, half, actions, stm):
        # hm_sp_raw_policy = get_normalized_probability over legal moves + pass of the game's position (both boards)
        pol = torch.from_numpy(policy_f32.astype(np.float16)).cuda().reshape(1, 4672).contiguous()
        moves, probs, caps, counts, on_turn = self.eng.raw_policy(pol, pol)
        which = 0                                                   # the tagged cases query board A of a white-to-move team
        n = int(counts[0, which])
        assert moves[0, which, :n].tolist() == [int(a) for a in actions], "action list differs"
        want = super().normalized_probability(policy_f32.astype(np.float16).astype(np.float32), True, actions, stm)
        assert np.array_equal(probs[0, which, :n], want), "GPU and oracle (portable exp) disagree"
        return probs[0, which, :n].copy()


@pytest.mark.parametrize("case", CASES, ids=[c["test"] for c in CASES])
def test_reference_known_answer_on_gpu(hm, case):
    if case["gpu"] == "softmax" and any(s["op"] == "normalize_logits" for s in case["steps"]):
        pytest.skip("bare-vector case: covered by test_softmax_robustness_rules_on_gpu")
    r = GpuRunner(hm)
    try:
        r.run(case["steps"])
    finally:
        r.close()


def test_softmax_robustness_rules_on_gpu(hm):
    """normalize_logits' rules (utils.h:127-167; test_move_gen.cc:643-657) through the device softmax: huge logits stay finite
    and sum to 1 with the larger one ahead; non-finite logits are skipped; all-non-finite -> uniform."""
    eng = hm.SearchEngine(1, 64)
    eng.set_games(O.Board().compact(0, False))
    b = O.Board()
    acts = list(b.legal_moves(0)) + [0]
    idx = [int(O.lib.ora_policy_index(int(m), 0)) for m in acts]

    def run(vals):
        pol = np.zeros(4672, np.float16)
        pol[idx] = np.asarray(vals, np.float16)
        t = torch.from_numpy(pol).cuda().reshape(1, 4672).contiguous()
        moves, probs, _, counts, _ = eng.raw_policy(t, t)
        n = int(counts[0, 0])
        assert n == len(acts) and moves[0, 0, :n].tolist() == [int(a) for a in acts]
        return probs[0, 0, :n]
    n = len(acts)
    p = run([1000.0, 999.0] + [-1000.0] * (n - 2))
    assert np.isfinite(p).all() and abs(float(p.sum(dtype=np.float32)) - 1.0) < 1e-6 and p[0] > p[1] > p[2]
    p = run([np.nan, -np.inf] * (n // 2) + [np.inf] * (n % 2))
    assert np.array_equal(p, np.full(n, np.float32(1.0) / np.float32(n)))           # fallback: uniform over all actions
    p = run([np.nan, 0.0, np.inf, -np.inf] + [0.0] * (n - 4))
    assert p[0] == 0 and p[2] == 0 and p[3] == 0 and abs(float(p.sum(dtype=np.float32)) - 1.0) < 1e-6
    assert np.array_equal(p[[1] + list(range(4, n))], np.full(n - 3, p[1]))
    eng.close()


def _hash_eval_gpu(planes):
    h = planes.cpu().numpy().view(np.uint16).reshape(-1, 4736)
    return tuple(torch.from_numpy(x.view(np.float16)).cuda() for x in O.hash_evaluator(h))


@pytest.mark.parametrize("pick", ["largest non-principal reply", "smallest reached reply"])
def test_tree_reuse_adopts_a_non_principal_reply_on_gpu(hm, pick):
    """EngineTest.TreeReuseRetainsNonPrincipalOpponentReplies (test_move_gen.cc:1332-1391): after a search the selected child AND every
    reply generated below it stay candidates for the next root; a reply that is NOT the child's principal move is adopted with
    everything searched beneath it.  The reference builds the tree by hand; the scripted case (search_cases.json) replays that on
    oracle/search.hpp.  Here the tree comes from a real search: the oracle lists its retained candidates, a non-principal reply is
    played, and the device (hm_sp_set_tree_reuse + k_begin's find_reusable_root) must adopt that very node — recovered visits equal
    the candidate's, and the follow-up search equals the oracle's follow-up search from the adopted node, edge for edge."""
    nodes = 400
    b = O.Board()
    s = O.Search(1, 1)
    s.set_tree_reuse(True)
    assert s.run(b, 0, False, nodes)
    first = s.edges()
    best = s.best_move()
    ret = s.retained()
    assert len(ret) >= 3                                                       # the selected child + at least two replies (reference: 3)
    replies = [r for r in ret[1:] if not r[5] and r[2] >= 1]                   # non-principal, reached (has a position on the device)
    assert replies, ret
    rep = max(replies, key=lambda r: r[2]) if pick.startswith("largest") else min(replies, key=lambda r: r[2])
    eng = hm.SearchEngine(1, 1000)
    eng.set_games(b.compact(0, False))
    mode = np.array([1], np.uint8)
    hm.check(hm.lib.hm_sp_set_tree_reuse(eng.h, mode.ctypes.data, 1))
    eng.begin_search(nodes)
    eng.run(_hash_eval_gpu)
    st = eng.root_stats()
    n = int(st["counts"][0])
    assert np.array_equal(st["visits"][0, :n], first["visits"]) and int(st["info"][0, 12]) == best and int(st["info"][0, 16]) == -1
    own = (int(first["move_a"][best]), int(first["move_b"][best]))
    eng.apply([own[0]], [own[1]])
    eng.apply([int(rep[0]) & 0xffffffff], [int(rep[1]) & 0xffffffff])
    assert b.make_moves(*own) == 0 and b.make_moves(int(rep[0]) & 0xffffffff, int(rep[1]) & 0xffffffff) == 0
    eng.begin_search(nodes)
    recovered = int(eng.root_stats()["info"][0, 16])
    assert recovered == int(rep[2]) >= 1, (recovered, rep)                     # the adopted root IS the retained reply node
    eng.run(_hash_eval_gpu)
    assert s.run(b, 0, False, nodes) and s.reused_visits() == recovered
    st2, e2 = eng.root_stats(), s.edges()
    n2 = int(st2["counts"][0])
    assert n2 == len(e2["visits"]) and np.array_equal(st2["visits"][0, :n2], e2["visits"])
    assert np.array_equal(st2["move_a"][0, :n2], e2["move_a"]) and np.array_equal(st2["q"][0, :n2], e2["q"])
    assert int(st2["info"][0, 1]) == s.info()["nodes"]
    eng.close()
