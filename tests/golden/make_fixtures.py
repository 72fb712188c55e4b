#!/usr/bin/env python3
"""Generates tests/golden/* from the REFERENCE (run in the build container only).

Two kinds of data, both plain data (no reference source text is stored):
  1. scenarios.json — the reference's own gtest known answers for Board::is_checkmate /
     is_draw / is_in_check (engine/tests/test_mate_detection.cc, test_draw_detection.cc),
     re-expressed as step lists {set_fen | set | push uci | expect_*}.  Expected values are the
     tests' own assertions; each scenario is replayed through oracle/_ref/libhmref.so here and
     dropped (with a note on stdout) if the replay does not agree with the assertion.
  2. ref_playout.npz, perft.json, policy.npz — outputs of the reference build itself
     (oracle/_ref/libhmref.so) on seeded random playouts: compact states, legal move lists in
     list order, mate/draw/check flags, hash_key classes; joint perft(1..3); policy tables.
usage: python tests/golden/make_fixtures.py
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_py as O  # noqa: E402

REF_TESTS = "/root/reference/engine/tests"
START_FEN = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"
BN = {"BOARD_A": 0, "BOARD_B": 1, "0": 0, "1": 1}


def split_tests(text):
    for m in re.finditer(r"TEST_F\(\s*(\w+)\s*,\s*(\w+)\s*\)\s*\{", text):
        depth, i = 1, m.end()
        while depth and i < len(text):
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        yield m.group(2), text[m.end():i - 1]


def strip_comments(body):
    body = re.sub(r"//[^\n]*", "", body)
    return re.sub(r"/\*.*?\*/", "", body, flags=re.S)


def parse_body(body):
    """-> list of steps, or None if the body uses constructs this translator does not cover."""
    body = strip_comments(body)
    if re.search(r"Board\s+future|ASSERT_FALSE\(blackReplies|for\s*\(Stockfish::Move\s+reply", body):
        return None
    # inline form push_move(BOARD_X, find_move(["board, BOARD_X,"] "uci")) -> pair + push
    body = re.sub(r'board\.push_move\(\s*(BOARD_[AB])\s*,\s*find_move\(\s*(?:board\s*,\s*BOARD_[AB]\s*,\s*)?"([^"]+)"\s*\)\s*\)',
                  r'{\1, "\2"} board.push_move(boardNum, mv)', body)
    steps, var_uci, last_cmp = [], {}, None
    # statements in textual order
    tok = re.compile(
        r'board\.set_fen\(\s*(BOARD_[AB])\s*,\s*(?:"([^"]*)"|board\.startingFen)\s*\)'
        r'|board\.set\(\s*((?:"[^"]*"\s*)+)\)'
        r'|\{\s*(BOARD_[AB])\s*,\s*"([^"]+)"\s*\}'
        r'|uci_move\(\s*(BOARD_[AB])\s*,\s*\w+\s*\)\s*==\s*"([^"]+)"'
        r'|(\w+)\s*=\s*m\s*;'
        r'|std::string\s+(\w+)\s*=\s*"([^"]+)"\s*;'
        r'|Stockfish::Move\s+(\w+)\s*=\s*Stockfish::UCI::to_move\(\s*\*board\.pos\[(BOARD_[AB])\]\s*,\s*(\w+)\s*\)'
        r'|Stockfish::Move\s+(\w+)\s*=\s*find_move\(\s*board\s*,\s*(BOARD_[AB])\s*,\s*"([^"]+)"\s*\)'
        r'|board\.push_move\(\s*(BOARD_[AB]|boardNum)\s*,\s*(\w+(?:\[0\])?)\s*\)'
        r'|EXPECT_(TRUE|FALSE)\(\s*board\.is_checkmate\(\s*Stockfish::(WHITE|BLACK)\s*(?:,\s*(true|false))?\s*\)\s*\)'
        r'|EXPECT_(TRUE|FALSE)\(\s*board\.is_draw\(\s*(\w*)\s*\)\s*\)'
        r'|EXPECT_(TRUE|FALSE)\(\s*board\.is_in_check\(\s*(BOARD_[AB])\s*\)\s*\)'
        r'|EXPECT_EQ\(\s*board\.repetition_count\(\s*(BOARD_[AB])\s*\)\s*,\s*(\d+)\s*\)')
    pending_pairs = []
    for m in tok.finditer(body):
        g = m.groups()
        if g[0]:
            steps.append({"op": "set_fen", "board": BN[g[0]], "fen": g[1] if g[1] is not None else START_FEN})
        elif g[2]:
            steps.append({"op": "set", "fen": "".join(re.findall(r'"([^"]*)"', g[2]))})
        elif g[3]:
            pending_pairs.append((BN[g[3]], g[4]))
        elif g[5]:
            last_cmp = g[6]
        elif g[7]:
            if last_cmp is not None and g[7] not in ("moves",):
                var_uci[g[7]] = last_cmp
        elif g[8]:
            var_uci[g[8]] = g[9]
        elif g[10]:
            if g[12] in var_uci:
                var_uci[g[10]] = var_uci[g[12]]
            elif g[12] in ("uci",):
                var_uci[g[10]] = "__pair__"
            else:
                return None
        elif g[13]:
            var_uci[g[13]] = g[15]
        elif g[16]:
            b, v = g[16], g[17]
            if b == "boardNum":
                if not pending_pairs:
                    return None
                for pb, pu in pending_pairs:
                    steps.append({"op": "push", "board": pb, "uci": pu})
                pending_pairs = []
            elif v.endswith("[0]"):
                steps.append({"op": "push_first", "board": BN[b]})
            elif v in var_uci and var_uci[v] != "__pair__":
                steps.append({"op": "push", "board": BN[b], "uci": var_uci[v]})
            else:
                return None
        elif g[18]:
            steps.append({"op": "expect_mate", "side": 0 if g[19] == "WHITE" else 1,
                          "adv": g[20] == "true", "value": g[18] == "TRUE"})
        elif g[21]:
            arg = g[22]
            ply = BN.get(arg, int(arg) if arg.isdigit() else 0)
            steps.append({"op": "expect_draw", "ply": ply, "value": g[21] == "TRUE"})
        elif g[23]:
            steps.append({"op": "expect_in_check", "board": BN[g[24]], "value": g[23] == "TRUE"})
        elif g[25]:
            steps.append({"op": "expect_repetition", "board": BN[g[25]], "value": int(g[26])})
    if pending_pairs:
        return None
    if not any(s["op"].startswith("expect") for s in steps):
        return None
    return steps


def replay(steps, impl):
    """Runs a scenario; returns list of (step index, got, want) mismatches."""
    b = O.Board(impl)
    bad = []
    for i, s in enumerate(steps):
        op = s["op"]
        if op == "set_fen":
            b.set_fen(s["board"], s["fen"])
        elif op == "set":
            b.set(s["fen"])
        elif op == "push":
            m = b.find_move(s["board"], s["uci"])
            if not m:
                bad.append((i, "no such move " + s["uci"], None))
                break
            b.push(s["board"], m)
        elif op == "push_first":
            b.push(s["board"], b.legal_moves(s["board"])[0])
        elif op == "expect_mate":
            got = b.is_checkmate(s["side"], s["adv"])
            if got != s["value"]:
                bad.append((i, got, s["value"]))
        elif op == "expect_draw":
            got = b.is_draw(s["ply"])
            if got != s["value"]:
                bad.append((i, got, s["value"]))
        elif op == "expect_in_check":
            got = b.in_check(s["board"])
            if got != s["value"]:
                bad.append((i, got, s["value"]))
        elif op == "expect_repetition":
            got = b.repetition_count(s["board"])
            if got != s["value"]:
                bad.append((i, got, s["value"]))
    return bad


def make_scenarios():
    out, skipped = [], []
    for fname in ("test_mate_detection.cc", "test_draw_detection.cc"):
        text = open(os.path.join(REF_TESTS, fname)).read()
        for name, body in split_tests(text):
            steps = parse_body(body)
            if steps is None:
                skipped.append(f"{fname}:{name} (construct not covered by the translator)")
                continue
            bad = replay(steps, "ref")
            if bad:
                skipped.append(f"{fname}:{name} (replay through the reference build disagrees: {bad[:2]})")
                continue
            out.append({"source": f"engine/tests/{fname}", "test": name, "steps": steps})
    return out, skipped


def make_playout(n_games=24, max_plies=90, seed=2024):
    rng = np.random.RandomState(seed)
    boards, moves, offs, flags, hkeys = [], [], [0], [], []
    for g in range(n_games):
        b = O.Board("ref")
        for ply in range(max_plies):
            team, adv = int(rng.randint(2)), int(rng.randint(2))
            boards.append(b.compact(team, adv)[0])
            lm = [b.legal_moves(0), b.legal_moves(1)]
            for l in lm:
                moves.append(l)
                offs.append(offs[-1] + len(l))
            flags.append([b.is_checkmate(0, False), b.is_checkmate(0, True), b.is_checkmate(1, False),
                          b.is_checkmate(1, True), b.is_draw(0), b.is_draw(1), b.in_check(0), b.in_check(1)])
            hkeys.append([b.hash_key(False), b.hash_key(True)])
            bd = int(rng.randint(2))
            if len(lm[bd]) == 0:
                bd ^= 1
            if len(lm[bd]) == 0:
                break
            # prefer captures a little so that pockets fill
            cand = lm[bd][rng.randint(len(lm[bd]))]
            for _ in range(2):
                alt = lm[bd][rng.randint(len(lm[bd]))]
                if O.ref.ref_is_capture(b.h, bd, int(alt)):
                    cand = alt
                    break
            b.push(bd, cand)
    hk = np.array(hkeys, dtype=np.uint64)
    _, cls = np.unique(hk.reshape(-1), return_inverse=True)       # equivalence classes only
    return dict(boards=np.array(boards, dtype=O.BOARD_DTYPE).view(np.uint8).reshape(len(boards), 208),
                moves=np.concatenate(moves).astype(np.uint32), offsets=np.array(offs, dtype=np.int64),
                flags=np.array(flags, dtype=np.uint8), hash_class=cls.reshape(-1, 2).astype(np.int32))


def main():
    if O.ref is None:
        sys.exit("oracle/_ref/libhmref.so missing: run `make -C oracle ref` in the build container")
    sc, skipped = make_scenarios()
    json.dump(sc, open(os.path.join(HERE, "scenarios.json"), "w"), indent=1)
    print(f"scenarios: {len(sc)} kept, {len(skipped)} skipped")
    for s in skipped:
        print("  skipped:", s)
    po = make_playout()
    np.savez_compressed(os.path.join(HERE, "ref_playout.npz"), **po)
    print("ref_playout:", po["boards"].shape, "moves", po["moves"].shape)
    b = O.Board("ref")
    perft = {"joint": {str(d): b.perft(d) for d in (1, 2, 3)},
             "joint_published": {"4": 39324713225},          # BASELINE.md §2 (survey probe of the reference)
             "single_board": {str(d): b.perft_single(0, d) for d in (1, 2, 3)}}  # test_move_gen.cc:1526-1571
    assert perft["single_board"] == {"1": 20, "2": 400, "3": 8902}
    json.dump(perft, open(os.path.join(HERE, "perft.json"), "w"), indent=1)
    normal, drop = O.policy_tables("ref")
    np.savez_compressed(os.path.join(HERE, "policy.npz"), normal=normal, drop=drop)
    pw = {"nonroot": [O.ref.ref_pw_allowed_children(v, 0) for v in range(0, 2001)],
          "root": [O.ref.ref_pw_allowed_children(v, 1) for v in range(0, 2001)]}
    json.dump(pw, open(os.path.join(HERE, "pw_schedule.json"), "w"))
    print("done")


if __name__ == "__main__":
    main()
