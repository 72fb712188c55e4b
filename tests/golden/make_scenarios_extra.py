#!/usr/bin/env python3
"""Writes tests/golden/scenarios_extra.json: the 14 Board::is_checkmate / is_draw gtest scenarios of the reference
(engine/tests/test_mate_detection.cc, test_draw_detection.cc) that make_fixtures.py's regex translator does not cover
(conditional expectations, make_moves, explicit make_drop moves, token-list replays, "after every reply" loops,
Board::legal_moves(Color, bool)), hand-transcribed as DATA: positions, moves and the asserted values.
Replayed on the oracle (and on the reference build where it exists) by tests/test_oracle_golden.py.

usage: python tests/golden/make_scenarios_extra.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
W, B = 0, 1
START = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"
KNIGHT, ROOK = 2, 4
C5, H8 = 34, 63


def drop(sq, pt):
    return {"drop": [sq, pt]}


S = [
    dict(source="engine/tests/test_mate_detection.cc:131-164", test="ReportedG4H5LeavesUnavoidableKnightDropMate", steps=[
        dict(op="set", fen="3q1r1k/1p4b1/p1r2p1p/3p1b1n/1npP1pB1/N1N1Q2P/PPP2PP1/1R3KR1[BPp] w - - 0 1|"
                           "5r1k/1p2q1b1/p1r2p1p/3p1b1n/1npP1pB1/N1N4P/PPPQ1PP1/1R3KR1[BPp] w - - 0 1"),
        dict(op="push", board=1, uci="g4h5"), dict(op="push", board=0, uci="g4h5"),
        dict(op="expect_count", board=0, cmp="gt", value=0),
        dict(op="for_each_reply", board=0, then=[dict(op="push", board=1, uci="N@h2"), dict(op="expect_mate", side=B, adv=False, value=True)])]),
    dict(source="engine/tests/test_mate_detection.cc:274-293", test="OnlyBackRankBlockingSquare", steps=[
        dict(op="set_fen", board=0, fen="Q6k/8/8/8/8/8/8/7K b - - 0 1"), dict(op="set_fen", board=1, fen="4k3/8/8/8/8/8/4p3/4K2R w - - 0 1"),
        dict(op="if_no_moves", board=0, then=[dict(op="expect_mate", side=B, adv=False, value=True)])]),
    dict(source="engine/tests/test_mate_detection.cc:516-526", test="LegalMovesEmptyWhenMated", steps=[
        dict(op="set_fen", board=0, fen="r1bqkb1r/pppp1Qpp/2n2n2/4p3/2B1P3/8/PPPP1PPP/RNB1K1NR b KQkq - 0 4"), dict(op="set_fen", board=1, fen=START),
        dict(op="expect_team_moves", side=B, adv=False, cmp="eq", value=0)]),
    dict(source="engine/tests/test_mate_detection.cc:529-537", test="LegalMovesNonEmptyWhenNotMated", steps=[
        dict(op="set_fen", board=0, fen=START), dict(op="set_fen", board=1, fen=START),
        dict(op="expect_team_moves", side=W, adv=False, cmp="gt", value=0)]),
    dict(source="engine/tests/test_mate_detection.cc:543-565", test="CaptureProvidesPartnerKnightDropMate", steps=[
        dict(op="set_fen", board=0, fen="2rq1rk1/pppnb1p1/4p1p1/3pP1pp/4P3/2N1P1B1/PPP2NPP/R2Q1RK1/NN b - - 0 3"),
        dict(op="set_fen", board=1, fen="r4rk1/ppp2p1p/4bB1p/8/6b1/2P5/P1PB1PPP/R3R1K1/qbbnnppPB w"),
        dict(op="make_moves", a="f8f2", b=None),
        dict(op="expect_hand", board=1, color=W, piece=KNIGHT, value=1),
        dict(op="expect_count", board=0, cmp="gt", value=0),
        dict(op="make_moves", a={"first": 0}, b=None),
        dict(op="make_moves", a=None, b="N@e7"),
        dict(op="expect_mate", side=W, adv=False, value=True)]),
    dict(source="engine/tests/test_mate_detection.cc:568-587", test="LegalMovesForTeamWhenNotMated", steps=[
        dict(op="set_fen", board=0, fen="5k1R/pppbrp2/2p1pQ2/8/2B1P3/2PN4/PPP3K1/8[RB] b - - 4 31"),
        dict(op="set_fen", board=1, fen="r6r/pppk1Ppp/2n1q3/3n2N1/8/2P5/P1P1NPPP/R1B1K2R[QBNPPPPqrbbbnnppppp] w KQ - 0 18"),
        dict(op="expect_team_moves", side=B, adv=False, cmp="gt", value=0),
        dict(op="expect_count", board=1, cmp="gt", value=0)]),
    dict(source="engine/tests/test_mate_detection.cc:670-685", test="AdjacentCheckCantBeBlocked", steps=[
        dict(op="set_fen", board=0, fen="4k3/3P4/8/8/8/8/8/4K3 b - - 0 1"), dict(op="set_fen", board=1, fen="4k3/8/8/8/3q4/8/8/4K3 w - - 0 1"),
        dict(op="if_no_moves", board=0, then=[dict(op="expect_mate", side=B, adv=False, value=True)])]),
    dict(source="engine/tests/test_mate_detection.cc:734-766", test="CheckmateAfterNonBlockingPartnerMove", steps=[
        dict(op="set_fen", board=0, fen="5k1R/pppbrp2/2p1pQ2/8/2B1P3/2PN4/PPP3K1/8[RB] b - - 4 31"),
        dict(op="set_fen", board=1, fen="r6r/pppk1Ppp/2n1q3/3n2N1/8/2P5/P1P1NPPP/R1B1K2R[QBNPPPPqrbbbnnppppp] w KQ - 0 18"),
        dict(op="expect_mate", side=B, adv=False, value=False),
        dict(op="make_moves", a=None, b=drop(C5, KNIGHT)),
        dict(op="expect_stm", board=0, value=B), dict(op="expect_stm", board=1, value=B),
        dict(op="expect_in_check", board=0, value=True),
        dict(op="expect_mate", side=B, adv=False, value=True)]),
    dict(source="engine/tests/test_mate_detection.cc:770-793", test="RookDropBackRankMate", steps=[
        dict(op="set_fen", board=0, fen="6k1/pppbrp2/2p1pQ2/8/2B1P3/2PN4/PPP3K1/8[R] w - - 0 1"),
        dict(op="set_fen", board=1, fen="r6r/pppk1Ppp/2n1q3/2Nn4/8/2P5/P1P1NPPP/R1B1K2R[QBNPPPPrbbbnnppppp] b KQ - 0 1"),
        dict(op="expect_stm", board=0, value=W),
        dict(op="make_moves", a=drop(H8, ROOK), b=None),
        dict(op="expect_in_check", board=0, value=True), dict(op="expect_count", board=0, cmp="eq", value=0),
        dict(op="expect_mate", side=B, adv=False, value=True)]),
    dict(source="engine/tests/test_mate_detection.cc:851-912", test="ReportedQueenDropD2IsMateInOne", steps=[
        dict(op="set", fen=START + "|" + START),
        dict(op="push_tokens", tokens=(
            "1e2e4 1e7e5 1g1f3 1b8c6 1f1c4 2e2e4 2c7c6 2b1c3 2d7d5 2e4d5 1P@e6 1d2d4 1e5d4 1c1g5 1f8e7 1g5e7 1d8e7 1e4e5 1d7d6 2c6d5 "
            "1P@f6 1g8f6 1e5f6 1g7f6 1e1g1 2d2d4 2g8f6 2g1f3 2b8c6 2f1b5 2c8g4 2P@a6 2d8c7 2a6b7 2c7b7 2P@a6 2b7c7 2b5c6 1N@f4 1P@g3 "
            "2c7c6 2B@b5 2g4f3 2d1f3 1B@h3 1N@g7 1e8d8 1B@h1 1h3g2 1h1g2 2c6b5 2c3b5 2B@d6 2b5d6 1B@h3 1B@h1 1h3g2 1h1g2 1Q@h3 2e7d6 "
            "2P@b7 2a8d8 2b7b8q 2d8b8 2P@c7 2b8c8 2B@a4 2B@d7 2a4d7 2f6d7 1B@h1 1f4g2 1h1g2 1h3g2 1g1g2 1B@g4 1h2h3 1g4h3 1g2h3 1P@g4 "
            "1h3g4 1h7h5 1g4h3 2f3d5 2c8c7 2d5a8 2B@d8 2B@g5 2B@f6 2g5f6 1B@g4 1h3h2 1g4f3 1d1f3 2d7f6 2B@b5 2P@d7 2c1g5 2f8e7 2g5f6 "
            "1N@g4 1h2g2 1c6e5 1g7e6 1c8e6 2e7f6 2N@d5 2B@d2 2e1d2 2N@e4 2d2d1").split()),
        dict(op="remember_opponent_of_mover", board=1),          # opponentTeam = side to move on board 2 (team colours of board B are mirrored)
        dict(op="push", board=1, uci="Q@d2"),
        dict(op="expect_in_check", board=1, value=True), dict(op="expect_count", board=1, cmp="eq", value=0),
        dict(op="expect_mate", side="opponent", adv=False, value=True), dict(op="expect_mate", side="opponent", adv=True, value=True)]),
    dict(source="engine/tests/test_mate_detection.cc:914-928", test="ReportedQueenDropE8IsMateInOne", steps=[
        dict(op="set", fen="r1bk1b1r/ppp1p1pp/8/6Nn/B7/2Nn4/PP1B1PPP/5K1R/PPNBRQpbbbq w - - 0 2|"
                           "r2qr1k1/p1p1ppP1/2p3nQ/3p2Pp/3P3n/2N1PP2/PPPp3P/R2K2R1/pP w - - 1 2"),
        dict(op="push", board=0, uci="Q@e8"),
        dict(op="expect_in_check", board=0, value=True), dict(op="expect_count", board=0, cmp="eq", value=0),
        dict(op="expect_mate", side=B, adv=False, value=True), dict(op="expect_mate", side=B, adv=True, value=True)]),
    dict(source="engine/tests/test_draw_detection.cc:112-122", test="RepetitionKeyIgnoresPocketPieces", steps=[
        dict(op="set_fen", board=0, fen="4k3/8/8/8/8/8/8/4K3 w - - 0 1"), dict(op="remember_rep_key", board=0),
        dict(op="set_fen", board=0, fen="4k3/8/8/8/8/8/8/4K3[P] w - - 0 1"), dict(op="expect_rep_key_same", board=0)]),
    dict(source="engine/tests/test_draw_detection.cc:124-154", test="ReportedKnightKingCycleIsOnlyTwofold", steps=[
        dict(op="push_tokens", tokens=(
            "1e2e4 1e7e5 1g1f3 1b8c6 1f1c4 1f8e7 1b1c3 2e2e4 2g8f6 2b1c3 2d7d5 2e4d5 1P@e6 1d2d3 1g8f6 2f6d5 1P@h6 1h8g8 1h6g7 1g8g7 "
            "1c1h6 1g7g2 2g1f3 2b8c6 2d2d4 2e7e6 2f1d3 2f8b4 2c1d2 2P@f4 2P@h6 2g7h6 1P@g7 1g2g7 1h6g7 2c3d5 1N@g2 1e1d2 2d8d5 2d2b4 "
            "1B@f4 2c6b4 1B@e3 1g2e3 1f2e3 1f4e3 1d2e3 1f6g4 1e3d2 1e7g5 1d2e1 2P@g7 2b4d3 2d1d3 1N@g2 1e1e2 1g2f4 1e2e1 2h8g8 2B@e4 "
            "2d5d8 2e4h7 1P@f2 1e1d2 1f4g2 1d2e2 1g2f4 1e2d2 1f4g2 1d2e2 2g8g7 2P@g6").split()),
        dict(op="expect_repetition", board=0, value=2),
        dict(op="expect_draw_on_board", board=0, value=False), dict(op="expect_draw", ply=0, value=False)]),
    dict(source="engine/tests/test_draw_detection.cc:315-346", test="ReportedBoardTwoMoveCompletesThreefoldRepetition", steps=[
        dict(op="push_tokens", tokens=(
            "1g1f3 1d7d5 1d2d4 1b8c6 1b1c3 1c8g4 1c1f4 1e7e6 1h2h3 1g4f3 1e2f3 1f8d6 1f4d6 1c7d6 2e2e4 2g8f6 2b1c3 2b8c6 2g1f3 2d7d5 "
            "2e4d5 2f6d5 2d2d4 2e7e5 2c3d5 2d8d5 1N@h5 1N@f5 1P@g4 1P@e3 1g4f5 2N@e3 2B@a5 2B@c3 2a5c3 2b2c3 1B@h4 1g2g3 1e3f2 1e1f2 "
            "1h4g3 1f2g3 1d8g5 2d5a5 2d4e5 2a5c3 1P@g4 1g5h5 1g4h5 1g8f6 1d1e2 1f6h5 1g3f2 1e8g8 1h1g1 1P@g3 1g1g3 1h5g3 1f2g3 1c6d4 "
            "1e2e3 2c1d2 2c3c5 2N@e4 2c5e7 2P@f6 2e7d8 2f6g7 2f8g7 2P@f6 2B@f8 2f6g7 2f8g7 2P@f6 2B@f8").split()),
        dict(op="expect_draw", ply=0, value=False),
        dict(op="push", board=1, uci="f6g7"),
        dict(op="expect_draw_on_board", board=1, value=True), dict(op="expect_draw", ply=0, value=True)]),
]

if __name__ == "__main__":
    out = os.path.join(HERE, "scenarios_extra.json")
    json.dump(S, open(out, "w"), indent=1)
    print(f"wrote {len(S)} scenarios to {out}")
