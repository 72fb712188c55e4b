"""UCI front end (hm_uci_*, interface/uci.cc) on the GPU search engine: the command dialect, `position` replay (FEN parsing,
single-board move lists, history) against the CPU restatement's Board, and `go nodes N` against the CPU restatement's search
under the shared stand-in network — same solver-aware best move (agent.cc:1031-1049), same node count."""
import re

import numpy as np
import pytest
import torch

import oracle_py as O

pytestmark = pytest.mark.gpu

START = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"


@pytest.fixture(scope="module")
def hm():
    import hivemind_amd as hm
    hm.init(0)
    return hm


class DeviceHashNet:
    native = False

    def __init__(self, hm):
        self.hm = hm

    def __call__(self, planes):
        n = planes.shape[0]
        f16 = dict(dtype=torch.float16, device=planes.device)
        out = (torch.empty(n, **f16), torch.empty((n, 4672), **f16), torch.empty((n, 4672), **f16), torch.empty((n, 3), **f16), torch.empty(n, **f16))
        self.hm.check(self.hm.lib.hm_hash_evaluator(planes.data_ptr(), n, 0, *[t.data_ptr() for t in out], None))
        return out


def _oracle_board(fen=None, moves=()):
    b = O.Board()
    if fen:
        b.set(fen)
    for tok in moves:
        bd = int(tok[0]) - 1
        m = b.find_move(bd, tok[1:])
        assert m != 0, tok
        b.push(bd, m)
    return b


def _joint(a, b):
    return "(" + O.move_uci(a) + "," + O.move_uci(b) + ")"


def test_uci_handshake_and_options(hm):
    u = hm.Uci(DeviceHashNet(hm), max_nodes=2000)
    text, quit_ = u.command("uci")
    assert text.startswith("id name hivemind\nid author aminwoo\n") and text.rstrip().endswith("uciok") and not quit_
    for opt in ("Hash", "MultiPV", "Ponder", "DrawContemptPermille", "PWCoefficientPermille", "RootPWCoefficientPermille", "PWExponentPermille",
                "Transpositions", "Team", "Mode"):
        assert f"option name {opt} " in text
    assert u.command("isready") == ("readyok\n", False)
    assert u.command("setoption name MultiPV value 3")[0] == "info string MultiPV set to 3\n"
    assert u.command("setoption name DrawContemptPermille value 2500")[0] == "info string DrawContemptPermille set to 1000\n"   # clamped, uci.cc:265
    assert u.command("setoption name Team value black") == ("", False)
    assert u.board()["team"][0] == 1
    assert u.command("stop") == ("", False)
    assert u.command("quit")[1] is True
    u.close()


@pytest.mark.parametrize("fen,moves", [
    (None, ["1e2e4", "2d2d4", "1e7e5", "2g8f6", "1g1f3", "1b8c6", "1f1c4", "1f8c5", "1e1g1"]),                     # castling text e1g1
    ("r1bqkb1r/pppp1ppp/2n2n2/4p3/2B1P3/5N2/PPPP1PPP/RNBQK2R[Pn] w KQkq - 4 4|" + START, ["1P@d3", "2e2e4", "1N@g4"]),   # pockets, drops
    ("rnbqkbnr/ppp1p1pp/8/3pPp2/8/8/PPPP1PPP/RNBQKBNR w KQkq f6 0 3|rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR/Qq b KQkq - 0 1",
     ["1e5f6", "2Q@e4"]),                                                                                            # en passant; pocket after the 8th slash
    ("8/P6k/8/8/8/8/8/K7[] w - - 0 1|" + START, ["1a7a8q", "2e2e4"]),                                                  # promotion text
])
def test_position_replay_matches_oracle_board(hm, fen, moves):
    u = hm.Uci(DeviceHashNet(hm), max_nodes=500)
    cmd = ("position fen " + fen if fen else "position startpos") + " moves " + " ".join(moves)
    text, _ = u.command(cmd)
    assert text == "", text
    want = _oracle_board(fen, moves).compact(0, False)
    got = u.board()
    assert got["pos"].tobytes() == want["pos"].tobytes()                      # both positions, Zobrist keys included
    assert np.array_equal(got["last_move"], want["last_move"]) and np.array_equal(got["rep_count"], want["rep_count"])
    # an illegal move is reported and the replay stops there (uci.cc:118-136)
    text, _ = u.command("position startpos moves 1e2e4 1e2e4")
    assert "Invalid move 'e2e4' on board 1 at move 2" in text
    u.close()


@pytest.mark.parametrize("team,mode,moves,nodes", [("white", "go", [], 200), ("black", "sit", ["1e2e4", "2d2d4"], 320),
                                                   ("white", "sit", ["1e2e4", "1e7e5", "2d2d4"], 160)])
def test_go_nodes_matches_oracle_search(hm, team, mode, moves, nodes):
    u = hm.Uci(DeviceHashNet(hm), max_nodes=2000)
    u.command(f"setoption name Team value {team}")
    u.command(f"setoption name Mode value {mode}")
    u.command("position startpos" + (" moves " + " ".join(moves) if moves else ""))
    text, _ = u.command(f"go nodes {nodes}")
    lines = text.strip().split("\n")
    assert lines[-1].startswith("bestmove (") and lines[0].startswith("info depth ")
    b = _oracle_board(None, moves)
    s = O.Search(1, 1)
    t, adv = (0 if team == "white" else 1), mode == "sit"
    assert s.run(b, t, adv, nodes)
    e = s.edges()
    best = s.best_move()
    want = "(" + O.move_uci(e["move_a"][best]) + "," + O.move_uci(e["move_b"][best]) + ")"
    assert lines[-1].split(" ponder ")[0] == "bestmove " + want, (lines, want)
    m = re.match(r"info depth (\d+) score (cp|mate) (-?\d+) nodes (\d+) nps (\d+) hashfull 0 tbhits 0 time (\d+) pv (\(.*\))$", lines[0])
    assert m and int(m.group(4)) == s.info()["nodes"] and m.group(7).split(" ")[0] == want, lines[0]
    # the principal variation (extract_pv_from_child, agent.cc:1218-1290) and the ponder move (agent.cc:1054-1113)
    pv = s.pv_lines(1, 20)[0]
    assert pv["child"] == best and m.group(7) == " ".join(_joint(a, bb) for a, bb in pv["moves"]), (lines[0], pv)
    if len(pv["moves"]) >= 2:
        assert lines[-1] == "bestmove " + want + " ponder " + _joint(*pv["moves"][1])
    if m.group(2) == "cp":
        import math
        assert int(m.group(3)) == int(180.0 * math.tan(1.56 * float(e["q"][best])))          # format_uci_score, agent.cc:75
    u.close()


@pytest.mark.parametrize("moves,nodes,multipv", [([], 400, 4), (["1e2e4", "1e7e5", "2d2d4", "2d7d5"], 800, 3), (["1e2e4", "2e2e4"], 96, 500)])
def test_go_multipv_lines_match_oracle(hm, moves, nodes, multipv):
    """`setoption MultiPV n`: n final info lines (fewer when the root has fewer children), children by visit count with the solver-aware
    best move first (agent.cc:917-965), each with score, PV and `multipv k`; Ponder false drops the ponder move (agent.cc:993-999)"""
    import math
    u = hm.Uci(DeviceHashNet(hm), max_nodes=2000)
    assert u.command(f"setoption name MultiPV value {multipv}")[0] == f"info string MultiPV set to {multipv}\n"
    assert u.command("setoption name Ponder value false")[0] == "info string Ponder set to false\n"
    u.command("position startpos" + (" moves " + " ".join(moves) if moves else ""))
    text, _ = u.command(f"go nodes {nodes}")
    lines = text.strip().split("\n")
    s = O.Search(1, 1)
    assert s.run(_oracle_board(None, moves), 0, False, nodes)
    want = s.pv_lines(multipv, 20)
    info = [l for l in lines if l.startswith("info depth")]
    assert len(info) == len(want) == min(multipv, len(s.edges()["visits"])) and len(want) >= 3
    assert want[0]["child"] == s.best_move()
    for k, (line, w) in enumerate(zip(info, want)):
        m = re.match(r"info depth (\d+) multipv (\d+) score (cp|mate) (-?\d+) nodes (\d+) nps \d+ hashfull 0 tbhits 0 time \d+ pv (\(.*\))$", line)
        assert m and int(m.group(2)) == k + 1 and int(m.group(5)) == s.info()["nodes"], line
        assert m.group(6) == " ".join(_joint(a, b) for a, b in w["moves"]), (k, line, w)
        if m.group(3) == "cp":
            assert int(m.group(4)) == int(180.0 * math.tan(1.56 * w["q"])), (line, w)
    assert any(len(w["moves"]) >= 3 for w in want)                    # the walk goes below the root's children
    assert lines[-1] == "bestmove " + _joint(*want[0]["moves"][0])   # Ponder false: no ponder move
    u.close()


def test_policy_command_matches_oracle(hm):
    """UCI::policy (uci.cc:306-393): value / WDL / plies of one forward of the position as the team sees it, and per board the legal
    moves + pass with normalised policy, most probable first; the FEN strings are Board::fen's"""
    moves = ["1e2e4", "2d2d4", "2d7d5", "1d7d5", "1e4d5", "1d8d5", "2e2e4", "2d5e4"]          # captures: pockets are not empty
    normal, drop = O.policy_tables("ora")
    promo_fen = "8/P6k/8/8/8/8/8/K7[] w - - 0 1|" + START                                     # a7a8 in all four flavours
    for team, mode, fen in (("white", "go", None), ("black", "sit", None), ("white", "go", promo_fen)):
        b = _oracle_board(fen, [] if fen else moves)
        u = hm.Uci(DeviceHashNet(hm), max_nodes=500)
        u.command(f"setoption name Team value {team}")
        u.command(f"setoption name Mode value {mode}")
        u.command("position fen " + fen if fen else "position startpos moves " + " ".join(moves))
        text, _ = u.command("policy")
        t = 0 if team == "white" else 1
        planes = O.planes(b.compact(t, mode == "sit"))
        v, pa, pb, w, ml = O.hash_evaluator(planes)
        f = lambda x: x.view(np.float16).astype(np.float32)
        lines = text.split("\n")
        assert lines[0] == "Value: %g" % f(v)[0]
        e = np.exp(f(w)[0] - f(w)[0].max()).astype(np.float32)
        got = [float(x) for x in lines[1].split()[1:]]
        assert lines[1].startswith("WDL: ") and np.allclose(got, [e[2] / e.sum(), e[1] / e.sum(), e[0] / e.sum()], rtol=2e-5)
        assert lines[2] == "Predicted plies to end: %g" % (f(ml)[0] * np.float32(100.0)) and lines[3] == ""
        at = 4
        for bd, pol in ((0, pa), (1, pb)):
            assert lines[at] == f"Board {'AB'[bd]} ({b.fen(bd)}):", lines[at]
            at += 1
            stm = int(b.compact(0, False)["pos"][0, bd]["stm"])
            if stm != (t if bd == 0 else t ^ 1):
                assert lines[at] == "  (not our turn)"
                at += 1
            else:
                legal = [int(m) for m in b.legal_moves(bd)] + [0]
                logit = []
                for m in legal:
                    kind, promo = (m >> 12) & 15, (m >> 16) & 63
                    if m == 0:
                        idx = 0
                    elif kind == 4:
                        idx = drop[stm, m & 63, promo]
                    elif kind == 3 and promo in (3, 4):
                        idx = -1                      # rook / bishop promotions have no policy plane (utils.h:183-216): listed with probability 0
                    else:
                        idx = normal[stm, (m >> 6) & 63, m & 63, 1 if (kind == 3 and promo == 2) else 0]
                    logit.append(f(pol)[0][idx] if idx >= 0 else -np.inf)
                logit = np.array(logit, np.float32)
                pr = np.exp(logit - logit.max()).astype(np.float32)
                pr = pr / pr.sum()
                rows = lines[at:at + len(legal)]
                at += len(legal)
                got = {r.strip().split(": ")[0]: float(r.split(": ")[1]) for r in rows}
                assert set(got) == {O.move_uci(m) for m in legal} and "pass" in got
                if fen and bd == 0:
                    assert got["a7a8r"] == got["a7a8b"] == 0.0 and got["a7a8q"] > 0.0 and got["a7a8n"] > 0.0
                for m, p_ in zip(legal, pr):
                    assert abs(got[O.move_uci(m)] - p_) <= 2e-5 * max(p_, 1e-3), (O.move_uci(m), got[O.move_uci(m)], p_)
                vals = [float(r.split(": ")[1]) for r in rows]
                assert vals == sorted(vals, reverse=True)
            if bd == 0:
                assert lines[at] == ""
                at += 1
        assert lines[at:] == [""]
        u.close()


def _tokens(joint):
    """'(e2e4,pass)' -> ['1e2e4']: the `position ... moves` tokens of a joint action"""
    a, b = joint.strip("()").split(",")
    return [f"{bd + 1}{m}" for bd, m in ((0, a), (1, b)) if m != "pass"]


@pytest.mark.parametrize("switch_side,nodes,max_nodes", [(False, 400, 4000), (True, 400, 4000), (False, 100, 560), (True, 120, 560)])
def test_tree_reuse_between_searches_matches_oracle(hm, switch_side, nodes, max_nodes):
    """ENABLE_TREE_REUSE (search_params.h:194; Agent::try_reuse_tree / store_next_root_candidates, agent.cc:1345-1451): the next `go`
    starts from the retained subtree when the new position is the selected child (engine switched to the other team) or one of the
    replies generated below it (same team, after the predicted reply); `ucinewgame` drops it.  Every search ≡ the CPU restatement
    running the same sequence with tree reuse on: recovered visits, node count, best move, the whole PV."""
    # max_nodes 560: the node pool is small enough for k_collect's LDS mirror (hm_search.hip), now with a root that is not node 0
    u = hm.Uci(DeviceHashNet(hm), max_nodes=max_nodes)
    s = O.Search(1, 1)
    s.set_tree_reuse(True)
    moves = ["1e2e4", "2d2d4"]
    team, adv = 0, False
    recovered = []
    for step in range(4):
        u.command(f"setoption name Team value {'white' if team == 0 else 'black'}")
        u.command(f"setoption name Mode value {'sit' if adv else 'go'}")
        u.command("position startpos moves " + " ".join(moves))
        text, _ = u.command(f"go nodes {nodes}")
        lines = text.strip().split("\n")
        assert s.run(_oracle_board(None, moves), team, adv, nodes)
        rv = s.reused_visits()
        recovered.append(rv)
        reuse_lines = [l for l in lines if l.startswith("info string Tree reuse")]
        assert reuse_lines == ([f"info string Tree reuse: {rv} visits recovered"] if rv >= 0 else []), (step, lines, rv)
        pv = s.pv_lines(1, 20)[0]
        info = [l for l in lines if l.startswith("info depth")][0]
        m = re.match(r"info depth \d+ score (cp|mate) -?\d+ nodes (\d+) nps \d+ hashfull 0 tbhits 0 time \d+ pv (\(.*\))$", info)
        assert m and int(m.group(2)) == s.info()["nodes"], (step, info, s.info())
        assert m.group(3) == " ".join(_joint(a, b) for a, b in pv["moves"]), (step, info, pv)
        assert lines[-1].startswith("bestmove " + _joint(*pv["moves"][0])), (step, lines[-1])
        assert len(pv["moves"]) >= 2
        moves += _tokens(_joint(*pv["moves"][0]))
        if switch_side:                       # the engine now answers for the other team: the selected child is the new root
            team, adv = team ^ 1, not adv
        else:                                 # the predicted reply is played: one of the retained replies is the new root
            moves += _tokens(_joint(*pv["moves"][1]))
    assert recovered[0] == -1 and all(r > 0 for r in recovered[1:]), recovered
    # ucinewgame: reset_search_state (uci.cc:77-86) -- the same position is searched from a fresh root again
    u.command("ucinewgame")
    s.reset()
    u.command(f"setoption name Team value {'white' if team == 0 else 'black'}")
    u.command(f"setoption name Mode value {'sit' if adv else 'go'}")
    u.command("position startpos moves " + " ".join(moves))
    text, _ = u.command(f"go nodes {nodes}")
    assert "Tree reuse" not in text
    assert s.run(_oracle_board(None, moves), team, adv, nodes) and s.reused_visits() == -1
    pv = s.pv_lines(1, 20)[0]
    assert text.strip().split("\n")[-1].startswith("bestmove " + _joint(*pv["moves"][0]))
    u.close()


def test_tree_reuse_falls_back_when_the_pool_is_full(hm):
    """the retained tree stays in the node pool; when the next node budget no longer fits behind it the search starts from an
    empty pool (a bounded-memory deviation from the reference, which keeps reusing) and still answers"""
    u = hm.Uci(DeviceHashNet(hm), max_nodes=300)
    moves = ["1e2e4", "2d2d4"]
    seen = []
    for step in range(4):
        u.command("position startpos moves " + " ".join(moves))
        text, _ = u.command("go nodes 300")
        lines = text.strip().split("\n")
        seen.append(any(l.startswith("info string Tree reuse") for l in lines))
        best = re.match(r"bestmove (\(\S+\)) ponder (\(\S+\))", lines[-1])
        assert best, lines
        moves += _tokens(best.group(1)) + _tokens(best.group(2))
    assert seen == [False] * 4, seen                       # 300 more nodes never fit behind a 300-node tree in a 300-node pool
    u.close()


def test_timed_searches_reuse_the_tree_with_a_shrunk_budget(hm):
    """a time-limited search asks for the whole pool; with a retained tree in it the budget shrinks to what still fits behind the
    retained nodes (hm_sp_set_tree_reuse mode 2) or, when less than half would fit, the search starts from an empty pool — either way
    it answers and never overflows the pool"""
    u = hm.Uci(DeviceHashNet(hm), max_nodes=1200)
    moves = ["1e2e4", "2d2d4"]
    reused = 0
    for step in range(5):
        u.command("position startpos moves " + " ".join(moves))
        text, _ = u.command("go movetime 40")
        assert "search failed" not in text and "overflow" not in text, text
        lines = text.strip().split("\n")
        reused += any(l.startswith("info string Tree reuse") for l in lines)
        best = re.match(r"bestmove (\(\S+\))(?: ponder (\(\S+\)))?", lines[-1])
        assert best, lines
        info = [l for l in lines if l.startswith("info depth")][-1]
        assert int(re.search(r" nodes (\d+) ", info).group(1)) <= 1200 * 1.05 + 16
        moves += _tokens(best.group(1)) + (_tokens(best.group(2)) if best.group(2) else [])
        if not best.group(2):
            break
    assert reused >= 1, "no search of the sequence started from the retained tree"
    u.close()


def _wait_for_bestmove(u, limit_s=20.0):
    import time
    text, t0 = "", time.time()
    while "bestmove" not in text and time.time() - t0 < limit_s:
        text += u.command("")[0]
        time.sleep(0.002)
    return text


def test_go_ponder_runs_until_ponderhit_or_stop(hm):
    """`go ponder` (uci.cc:150-151, agent.h:31): the search ignores its budget and prints nothing until `ponderhit` (then the budget
    applies: nodes -> at least that many, movetime -> the clock starts at ponderhit) or `stop`"""
    import time
    u = hm.Uci(DeviceHashNet(hm), max_nodes=60000)
    u.command("position startpos moves 1e2e4 2d2d4")
    # node budget: far more than 64 nodes are searched while pondering; after ponderhit the budget is already met
    assert u.command("go ponder nodes 64") == ("", False) and u.busy()
    time.sleep(0.25)
    assert u.command("")[0] == "" and u.busy()                              # still pondering: nothing printed
    assert u.command("ponderhit") [1] is False
    text = _wait_for_bestmove(u)
    info = [l for l in text.split("\n") if l.startswith("info depth")][0]
    nodes = int(re.search(r" nodes (\d+) ", info).group(1))
    assert nodes > 64 and text.strip().split("\n")[-1].startswith("bestmove (") and not u.busy()
    # stop: ends the ponder search, bestmove follows at once; the answer is a legal joint action
    u.command("go ponder movetime 100000")
    time.sleep(0.1)
    text, _ = u.command("stop")
    last = text.strip().split("\n")[-1]
    m = re.match(r"bestmove \((\S+),(\S+)\)", last)
    assert m and not u.busy(), text
    b = _oracle_board(None, ["1e2e4", "2d2d4"])
    for bd, mv in ((0, m.group(1)), (1, m.group(2))):
        assert mv == "pass" or b.find_move(bd, mv) != 0, (bd, mv)
    # movetime after ponderhit: the clock starts at ponderhit, not at `go`
    u.command("go ponder movetime 150")
    time.sleep(0.3)
    assert u.busy() and u.command("")[0] == ""
    t0 = time.time()
    u.command("ponderhit")
    text = _wait_for_bestmove(u)
    dt = (time.time() - t0) * 1e3
    info = [l for l in text.split("\n") if l.startswith("info depth")][-1]
    assert 100 <= int(re.search(r" time (\d+) ", info).group(1)) < 1500 and dt >= 100, (info, dt)
    # a new command while pondering stops the search first (UCI::go / setoption call stop(), uci.cc:160, 217)
    u.command("go ponder nodes 100")
    text, _ = u.command("position startpos")
    assert "bestmove" in text and not u.busy()
    assert u.command("go nodes 80")[0].strip().split("\n")[-1].startswith("bestmove (")
    u.close()


def test_go_without_a_board_on_turn_says_none(hm):
    """neither of the team's boards is on turn and sitting out is not allowed: Agent::run_search prints `bestmove (none)`
    (agent.cc:438-450); the CPU restatement refuses the search as well"""
    moves = ["1e2e4", "2d2d4", "1e7e5"]
    u = hm.Uci(DeviceHashNet(hm), max_nodes=500)
    u.command("setoption name Team value black")
    u.command("setoption name Mode value sit")
    u.command("position startpos moves " + " ".join(moves))
    assert u.command("go nodes 100")[0].strip().split("\n")[-1] == "bestmove (none)"
    assert not O.Search(1, 1).run(_oracle_board(None, moves), 1, True, 100)
    u.close()


def test_go_movetime_returns_a_legal_best_move(hm):
    from hivemind_amd import net as N
    torch.manual_seed(0)
    u = hm.Uci(N.FusedNet(N.rise_v3_small()), max_nodes=50000)
    u.command("position startpos moves 1e2e4 2e2e4")
    u.command("go nodes 64")                      # first launches of this process (code objects, the network's first forward) are not timed
    u.command("ucinewgame")
    u.command("position startpos moves 1e2e4 2e2e4")
    text, _ = u.command("go movetime 120")
    last = text.strip().split("\n")[-1]
    m = re.match(r"bestmove \((\S+),(\S+)\)(?: ponder \(\S+,\S+\))?$", last)
    assert m, text
    b = _oracle_board(None, ["1e2e4", "2e2e4"])
    for bd, mv in ((0, m.group(1)), (1, m.group(2))):
        assert mv == "pass" or b.find_move(bd, mv) != 0, (bd, mv)
    infos = [l for l in text.split("\n") if l.startswith("info depth")]
    t = int(re.search(r" time (\d+) ", infos[-1]).group(1))
    assert 55 <= t < 1500, infos[-1]                                         # ran towards the deadline (early stopping may cut it), then finished the batches in flight
    # "Report each completed depth once" (agent.cc:680-711): one line per new maximum depth while the clock runs, then the final line
    depths = [int(l.split()[2]) for l in infos]
    assert len(infos) >= 2 and depths[:-1] == sorted(set(depths[:-1])) and depths[-1] >= depths[-2], depths
    assert all(" pv (" in l for l in infos)
    u.close()


def test_stop_and_isready_reach_an_ordinary_search(hm):
    """Every `go` runs on the engine's worker thread (mainSearchThread, uci.cc:192-205): the C ABI returns at once, `isready` is
    answered while the search runs, `stop` ends a long `go movetime` promptly (UCI::stop, uci.cc:64-75) and bestmove follows."""
    import time
    u = hm.Uci(DeviceHashNet(hm), max_nodes=100000)
    u.command("position startpos moves 1e2e4 2d2d4")
    u.command("go nodes 64")                                                # warm-up (first launches of the process)
    t0 = time.perf_counter()
    assert u.command("go movetime 60000", wait=False) == ("", False)
    assert time.perf_counter() - t0 < 1.0 and u.busy()
    assert u.command("isready")[0].endswith("readyok\n") and u.busy()       # the search goes on
    time.sleep(0.05)
    text, _ = u.command("stop")
    assert time.perf_counter() - t0 < 10.0 and not u.busy()
    assert re.match(r"bestmove \((\S+),(\S+)\)", text.strip().split("\n")[-1]), text
    # and a node-limited one, stopped long before its budget
    assert u.command("go nodes 90000", wait=False) == ("", False)
    time.sleep(0.05)
    text, _ = u.command("stop")
    nodes = int(re.search(r" nodes (\d+) ", [l for l in text.split("\n") if l.startswith("info depth")][-1]).group(1))
    assert nodes < 90000 and text.strip().split("\n")[-1].startswith("bestmove ("), text
    u.close()


def test_cli_loop_sees_two_commands_written_at_once():
    """the stdin loop of `python -m hivemind_amd.uci`: `go ponder ...` and `ponderhit` arriving in ONE pipe write must both be
    seen (a select() on the descriptor next to a buffered readline would leave the second line unseen and never print bestmove)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.Popen([sys.executable, "-m", "hivemind_amd.uci", "--model", "small", "--max-nodes", "4000"], cwd=root,
                         stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    import queue
    import threading
    got = queue.Queue()
    threading.Thread(target=lambda: [got.put(l) for l in iter(p.stdout.readline, b"")], daemon=True).start()
    try:
        p.stdin.write(b"position startpos moves 1e2e4 2d2d4\ngo ponder nodes 64\nponderhit\n")
        p.stdin.flush()
        best = None
        try:
            while best is None:                                             # nothing else is written until bestmove has arrived
                line = got.get(timeout=180).decode()
                if line.startswith("bestmove ("):
                    best = line
        except queue.Empty:
            pass
        p.stdin.write(b"quit\n")
        p.stdin.flush()
        p.wait(timeout=60)
    finally:
        if p.poll() is None:
            p.kill()
    assert best is not None, p.stderr.read().decode()[-2000:]
    assert p.returncode == 0


def test_reference_ponder_mode_cases(hm):
    """PonderModeTest.* of the reference (engine/tests/test_move_gen.cc:1304-1330) on the product's UCI layer:
    AgentPonderHitTransitions — ponderhit on an engine that is not searching is a safe no-op and leaves it not pondering;
    SearchOptionsPonderFlags — an ordinary `go` is not a ponder search (it answers by itself), `go ponder` is (silent, running);
    SearchInfoResetStartTime — the clock of a ponder search restarts at ponderhit: pondering for longer than the move time
    does not end the search, and after ponderhit it still gets (about) its whole move time."""
    import time
    u = hm.Uci(DeviceHashNet(hm), max_nodes=50000)
    u.command("position startpos moves 1e2e4 2d2d4")
    assert not u.busy()
    assert u.command("ponderhit") == ("", False) and not u.busy()          # Agent::ponderhit is safe when not running
    text, _ = u.command("go nodes 64")                                      # isPonder = false: answers on its own
    assert text.strip().split("\n")[-1].startswith("bestmove (") and not u.busy()
    assert u.command("go ponder movetime 250") == ("", False) and u.busy()  # isPonder = true: silent, running
    time.sleep(0.6)                                                         # more than twice the move time spent pondering
    assert u.busy() and u.command("")[0] == ""
    t0 = time.perf_counter()
    u.command("ponderhit")                                                  # SearchInfo::reset_start_time: the move time starts now
    text = ""
    while u.busy() and time.perf_counter() - t0 < 10.0:
        time.sleep(0.005)
        text += u.command("")[0]
    text += u.command("")[0]
    dt = time.perf_counter() - t0
    assert text.strip().split("\n")[-1].startswith("bestmove (") and not u.busy(), text
    infos = [l for l in text.split("\n") if l.startswith("info depth")]
    t = int(re.search(r" time (\d+) ", infos[-1]).group(1))
    assert t < 600 and dt < 3.0, (t, dt)                                    # elapsed counts from ponderhit, not from `go`
    u.close()
