"""Replays tests/golden/search_cases.json (the reference's gtest known answers for the search core,
engine/tests/test_move_gen.cc) on the CPU restatement through oracle/oracle_lab.cc.
TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import json
import math
import os

import numpy as np

import oracle_py as O

L = O.lib
_vp, _i, _u32, _u64, _f, _d = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float, C.c_double
_P = C.POINTER
for name, res, args in [
    ("ora_lab_new", _vp, [_i, _i]), ("ora_lab_free", None, [_vp]), ("ora_lab_config", _i, [_vp, C.c_char_p, _d]),
    ("ora_lab_config_get", _d, [_vp, C.c_char_p]), ("ora_lab_node_new", _i, [_vp, _i, _u64]),
    ("ora_lab_init_expand", _i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    ("ora_lab_update", None, [_vp, _i, _i, _f]), ("ora_lab_update_terminal", None, [_vp, _i, _f]),
    ("ora_lab_apply_vl", None, [_vp, _i, _i]), ("ora_lab_remove_vl", None, [_vp, _i, _i]), ("ora_lab_mark", None, [_vp, _i, _i, _i]),
    ("ora_lab_set_value", None, [_vp, _i, _f]), ("ora_lab_set_depth", None, [_vp, _i, _i]), ("ora_lab_child", _i, [_vp, _i, _i]),
    ("ora_lab_replace_child", None, [_vp, _i, _i, _i]),
    ("ora_lab_expand_next", _i, [_vp, _i, _i, _u64, _i, _P(_i), _P(_i), _P(_u32), _P(_u32)]),
    ("ora_lab_should_expand", _i, [_vp, _i]), ("ora_lab_has_unexpanded", _i, [_vp, _i]),
    ("ora_lab_peek_next", None, [_vp, _i, _P(_u32), _P(_u32), _P(_f)]),
    ("ora_lab_joint_action", None, [_vp, _i, _i, _P(_u32), _P(_u32), _P(_f), _P(_f)]),
    ("ora_lab_select", _i, [_vp, _i, _P(_i), _P(_i), _P(_i)]), ("ora_lab_reserve", _i, [_vp, _i]), ("ora_lab_release", None, [_vp, _i]),
    ("ora_lab_init_types", None, [_vp, _i]), ("ora_lab_update_type", _i, [_vp, _i, _i, _i]),
    ("ora_lab_backup", None, [_vp, _vp, _vp, _i, _f]), ("ora_lab_cancel_vl", None, [_vp, _vp, _vp, _i]),
    ("ora_lab_best_move", _i, [_vp, _i, _f, _f]), ("ora_lab_get", _d, [_vp, _i, _i, _i]), ("ora_lab_set_root", None, [_vp, _i]),
    ("ora_lab_tt_insert_or_get", _i, [_vp, _u64, _i]), ("ora_lab_tt_hits", _i, [_vp]),
    ("ora_lab_store_candidates", None, [_vp, _vp, _i]), ("ora_lab_retained_count", _i, [_vp]), ("ora_lab_try_reuse", _i, [_vp, _vp, _i, _i]),
    ("ora_lab_select_and_expand", _i, [_vp, _vp, _i, _P(_i), _P(_i), _P(_i)]), ("ora_lab_shape_value", _f, [_vp, _f, _vp, _f]),
    ("ora_normalize_logits", None, [_vp, _i, _i, _vp]), ("ora_normalized_probability", None, [_vp, _i, _vp, _i, _i, _i, _vp]),
    ("ora_is_policy_move_representable", _i, [_u32]), ("ora_policy_index_of_label", _i, [C.c_char_p]), ("ora_f32_to_f16", C.c_uint16, [_f]),
    ("ora_get_cpuct3", _f, [_f, _f, _f]), ("ora_allowed_children3", _i, [_i, _f, _f]),
    ("ora_board_record_position", None, [_vp, _i]), ("ora_board_add_to_hand", None, [_vp, _i, _i, _i]),
    ("ora_board_count_in_hand", _i, [_vp, _i, _i, _i]), ("ora_board_rule50", _i, [_vp, _i]), ("ora_board_stm", _i, [_vp, _i]),
    ("ora_board_history_len", _i, [_vp, _i]), ("ora_board_prefix_len", _i, [_vp, _i]), ("ora_board_last_move", _u32, [_vp, _i]),
    ("ora_board_is_legal_move", _i, [_vp, _i, _u32]), ("ora_board_uci_to_move", _u32, [_vp, _i, C.c_char_p]),
]:
    fn = getattr(L, name)
    fn.restype, fn.argtypes = res, args

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "search_cases.json")
FIELD = dict(q=0, visits=1, type=2, end_in_ply=3, n_children=4, team=5, child_q=6, child_visits=7, virtual_loss=8, expanded=9,
             pending=10, expanded_count=11, child_prior=12, value_sum=13, depth=14)


def load_cases():
    return json.load(open(GOLDEN))["cases"]


def float_eq(a, b):
    """gtest EXPECT_FLOAT_EQ: within 4 units in the last place."""
    a, b = np.float32(a), np.float32(b)
    if np.isnan(a) or np.isnan(b):
        return False
    ia, ib = (int(np.array(x, np.float32).view(np.int32)) for x in (a, b))
    ia, ib = (0x80000000 - x if x < 0 else x + 0x80000000 for x in (ia, ib))
    return abs(ia - ib) <= 4


def _num(x):
    return {"nan": math.nan, "inf": math.inf, "-inf": -math.inf}.get(x, x) if isinstance(x, str) else x


class Runner:
    """One case = one fresh lab (Search with default RuntimeConfig, tie/exp modes of the reference: 0/0)."""

    def __init__(self, tie_mode=0, exp_mode=0):
        self.h = L.ora_lab_new(tie_mode, exp_mode)
        self.exp_mode = exp_mode
        self.nodes, self.boards, self.vars, self.last_made = {}, {}, {}, {}
        self.remembered = {}

    def close(self):
        L.ora_lab_free(self.h)

    # ---- helpers ----
    def move(self, m, board=None):
        if isinstance(m, dict):
            if "uci" in m:
                b, u = m["uci"]
                mv = self.boards[board].find_move(b, u)
                assert mv, f"{u} not legal"
                return mv
            b, k = m["legal"]
            return int(self.boards[board].legal_moves(b)[k])
        return int(m)

    def nid(self, name):
        return self.nodes[name]

    def name_of(self, node_id):
        for k, v in self.nodes.items():
            if v == node_id:
                return k
        return None

    def val(self, x):
        return self.vars[x] if isinstance(x, str) and x in self.vars else x

    def get(self, name, what, idx=-1):
        if what == "should_expand":
            return bool(L.ora_lab_should_expand(self.h, self.nid(name)))
        if what == "has_unexpanded":
            return bool(L.ora_lab_has_unexpanded(self.h, self.nid(name)))
        return L.ora_lab_get(self.h, self.nid(name), FIELD[what], int(self.val(idx)))

    def board_state(self, b, adv):
        bd = self.boards[b]
        return (bd.compact(0, False).tobytes(), bd.hash_key(adv), L.ora_board_history_len(bd.h, 0), L.ora_board_history_len(bd.h, 1),
                L.ora_board_prefix_len(bd.h, 0), L.ora_board_prefix_len(bd.h, 1))

    def traj(self, t):
        ids = np.array([self.nid(n) for n, _ in t], np.int32)
        idx = np.array([i for _, i in t], np.int32)
        return ids, idx

    # ---- the interpreter ----
    def run(self, steps):
        for s in steps:
            self.step(s)

    def step(self, s):
        op, h = s["op"], self.h
        if op == "repeat":
            for _ in range(s["times"]):
                self.step(s["step"])
        elif op == "config":
            assert L.ora_lab_config(h, s["key"].encode(), float(s["value"])) == 0, s
        elif op == "config_expect":
            v = L.ora_lab_config_get(h, s["key"].encode())
            assert (float_eq(v, s["float_eq"]) if "float_eq" in s else v == s["eq"]), (s, v)
        elif op == "node":
            hv = s.get("hash", 0)
            self.nodes[s["as"]] = L.ora_lab_node_new(h, s["team"], int(self.val(hv)))
            if "depth" in s:
                L.ora_lab_set_depth(h, self.nodes[s["as"]], s["depth"])
        elif op == "init_expand":
            a = np.array([self.move(m, s.get("board")) for m in s["a"]], np.uint32)
            b = np.array([self.move(m, s.get("board")) for m in s["b"]], np.uint32)
            pa, pb = np.array(s["pa"], np.float32), np.array(s["pb"], np.float32)
            r = L.ora_lab_init_expand(h, self.nid(s["node"]), a.ctypes.data, len(a), b.ctypes.data, len(b), pa.ctypes.data, pb.ctypes.data,
                                      int(s["adv"]), int(s["a_on"]), int(s["b_on"]), None, None)
            assert bool(r) == s["expect"], s
        elif op == "child":
            c = L.ora_lab_child(h, self.nid(s["node"]), s["idx"])
            assert c >= 0, s
            self.nodes[s["as"]] = c
        elif op in ("update", "update_terminal", "apply_vl", "remove_vl", "mark", "set_value", "release", "init_types"):
            n = self.nid(s["node"])
            {"update": lambda: L.ora_lab_update(h, n, s["idx"], s["value"]), "update_terminal": lambda: L.ora_lab_update_terminal(h, n, s["value"]),
             "apply_vl": lambda: L.ora_lab_apply_vl(h, n, s["idx"]), "remove_vl": lambda: L.ora_lab_remove_vl(h, n, s["idx"]),
             "mark": lambda: L.ora_lab_mark(h, n, s["type"], s["ply"]), "set_value": lambda: L.ora_lab_set_value(h, n, s["value"]),
             "release": lambda: L.ora_lab_release(h, n), "init_types": lambda: L.ora_lab_init_types(h, n)}[op]()
        elif op == "reserve":
            assert bool(L.ora_lab_reserve(h, self.nid(s["node"]))) == s["expect"], s
        elif op == "update_type":
            assert bool(L.ora_lab_update_type(h, self.nid(s["node"]), s["idx"], s["type"])) == s["expect"], s
        elif op == "best_move":
            got = L.ora_lab_best_move(h, self.nid(s["node"]), L.ora_lab_config_get(h, b"qVetoDelta"), L.ora_lab_config_get(h, b"qValueWeight"))
            assert got == s["expect"], (s, got)
        elif op == "replace_child":
            L.ora_lab_replace_child(h, self.nid(s["node"]), s["idx"], self.nid(s["child"]))
        elif op == "expand_next":
            idx, res, ma, mb = _i(-1), _i(0), _u32(0), _u32(0)
            ex = self.nid(s["existing"]) if s.get("existing") else -1
            c = L.ora_lab_expand_next(h, self.nid(s["node"]), ex, int(self.val(s.get("hash", 0))), int(s.get("reserve", False)), idx, res, ma, mb)
            if "expect_non_null" in s:
                assert (c >= 0) == s["expect_non_null"], s
            if "expect_idx" in s:
                assert idx.value == s["expect_idx"], (s, idx.value)
            if "idx_as" in s:
                self.vars[s["idx_as"]] = idx.value
            if s.get("as"):
                self.nodes[s["as"]] = c
        elif op == "expand_all":
            n, count = self.nid(s["node"]), 0
            while L.ora_lab_has_unexpanded(h, n):
                idx, res, ma, mb = _i(-1), _i(0), _u32(0), _u32(0)
                c = L.ora_lab_expand_next(h, n, -1, 0, 0, idx, res, ma, mb)
                if c >= 0:
                    count += 1
                    ga, gb, fa, fb = _u32(0), _u32(0), _f(0), _f(0)
                    L.ora_lab_joint_action(h, n, idx.value, ga, gb, fa, fb)
                    assert ga.value == ma.value, (idx.value, ga.value, ma.value)
            assert count == s["expect_count"], count
        elif op == "select":
            idx, res, pend = _i(-1), _i(0), _i(-1)
            c = L.ora_lab_select(h, self.nid(s["node"]), idx, res, pend)
            self.vars["_selected_idx"] = idx.value
            if "expect_child" in s:
                assert c == (self.nid(s["expect_child"]) if s["expect_child"] else -1), (s, c)
            if s.get("expect_child_non_null"):
                assert c >= 0, s
            if "expect_idx" in s:
                assert idx.value == s["expect_idx"], (s, idx.value)
            if "expect_reserved" in s:
                assert bool(res.value) == s["expect_reserved"], s
            if "expect_pending" in s:
                assert pend.value == (self.nid(s["expect_pending"]) if s["expect_pending"] else -1), (s, pend.value)
            if s.get("as"):
                self.nodes[s["as"]] = c
        elif op == "remove_vl_selected":
            L.ora_lab_remove_vl(h, self.nid(s["node"]), self.vars["_selected_idx"])
        elif op == "backup":
            ids, idx = self.traj(s["traj"])
            L.ora_lab_backup(h, ids.ctypes.data, idx.ctypes.data, len(ids), s["value"])
        elif op == "cancel_vl":
            ids, idx = self.traj(s["traj"])
            L.ora_lab_cancel_vl(h, ids.ctypes.data, idx.ctypes.data, len(ids))
        elif op == "expect":
            got = self.get(s["node"], s["what"], s.get("idx", -1))
            if "float_eq" in s:
                assert float_eq(got, s["float_eq"]), (s, got)
            else:
                assert got == s["eq"], (s, got)
        elif op == "expect_same":
            assert self.nid(s["a"]) == self.nid(s["b"]), s
        elif op == "set_root":
            L.ora_lab_set_root(h, self.nid(s["node"]))
        elif op == "tt_insert":
            got = L.ora_lab_tt_insert_or_get(h, int(self.val(s["hash"])), self.nid(s["node"]))
            assert got == self.nid(s["expect"]), (s, got)
        elif op == "expect_tt_hits":
            assert L.ora_lab_tt_hits(h) == s["eq"], L.ora_lab_tt_hits(h)
        elif op == "select_and_expand":
            res, pend, tl = _i(0), _i(-1), _i(0)
            leaf = L.ora_lab_select_and_expand(h, self.boards[s["board"]].h, int(s["root_adv"]), res, pend, tl)
            assert leaf == (self.nid(s["expect_leaf"]) if s["expect_leaf"] else -1), (s, leaf)
            if "expect_reserved" in s:
                assert bool(res.value) == s["expect_reserved"], s
            if "expect_pending" in s:
                assert pend.value == (self.nid(s["expect_pending"]) if s["expect_pending"] else -1), (s, pend.value)
        elif op == "store_candidates":          # Agent::store_next_root_candidates (agent.cc:1373-1451) on the lab's root
            L.ora_lab_store_candidates(h, self.boards[s["board"]].h, int(s["adv"]))
            if "expect_retained" in s:
                assert L.ora_lab_retained_count(h) == s["expect_retained"], (s, L.ora_lab_retained_count(h))
        elif op == "try_reuse":                 # Agent::try_reuse_tree (agent.cc:1345-1371) for the position on `board`
            got = L.ora_lab_try_reuse(h, self.boards[s["board"]].h, int(s["adv"]), s["team"])
            assert got == (self.nid(s["expect"]) if s["expect"] else -1), (s, got)
            if s.get("as") and got >= 0:
                self.nodes[s["as"]] = got
        elif op == "joint_make":                # Board::make_moves of joint action idx of `node`, kept on the board
            ma, mb, fa, fb = _u32(0), _u32(0), _f(0), _f(0)
            L.ora_lab_joint_action(h, self.nid(s["node"]), s["idx"], ma, mb, fa, fb)
            assert self.boards[s["board"]].make_moves(ma.value, mb.value) == 0
        elif op == "peek_make_hash" or op == "action_make_hash":
            ma, mb = _u32(0), _u32(0)
            if op == "peek_make_hash":
                pr = _f(0)
                L.ora_lab_peek_next(h, self.nid(s["node"]), ma, mb, pr)
            else:
                fa, fb = _f(0), _f(0)
                L.ora_lab_joint_action(h, self.nid(s["node"]), s["idx"], ma, mb, fa, fb)
            bd = self.boards[s["board"]]
            assert bd.make_moves(ma.value, mb.value) == 0
            self.vars[s["as"]] = bd.hash_key(s["adv"])
            bd.unmake_moves(ma.value, mb.value)
        # ---- Board ----
        elif op == "board":
            self.boards[s["as"]] = O.Board()
        elif op == "set_fen":
            self.boards[s["board"]].set_fen(s["which"], s["fen"])
        elif op == "set":
            self.boards[s["board"]].set(s["fen"])
        elif op == "push_uci":
            bd = self.boards[s["board"]]
            m = L.ora_board_uci_to_move(bd.h, s["which"], s["uci"].encode())
            assert m, s
            bd.push(s["which"], m)
        elif op == "pop":
            self.boards[s["board"]].pop(s["which"])
        elif op == "record_position":
            L.ora_board_record_position(self.boards[s["board"]].h, s["which"])
        elif op == "add_to_hand":
            L.ora_board_add_to_hand(self.boards[s["board"]].h, s["which"], s["color"], s["piece"])
        elif op == "make_moves":
            a, b = self.move(s["a"], s["board"]), self.move(s["b"], s["board"])
            assert self.boards[s["board"]].make_moves(a, b) == 0
            self.last_made[s["board"]] = (a, b)
        elif op == "unmake_moves":
            self.boards[s["board"]].unmake_moves(*self.last_made[s["board"]])
        elif op == "remember":
            self.remembered[s["as"]] = {adv: self.board_state(s["board"], adv) for adv in (False, True)}
        elif op == "expect_unchanged":
            assert self.board_state(s["board"], s["adv"]) == self.remembered[s["since"]][s["adv"]], s
        elif op == "expect_changed":
            assert self.board_state(s["board"], s["adv"])[1] != self.remembered[s["since"]][s["adv"]][1], s
        elif op == "hash_compare":
            ka, kb = (self.boards[n].hash_key(adv) for n, adv in (s["a"], s["b"]))
            assert (ka == kb) == s["equal"], s
        elif op == "rep_key_compare":
            ka, kb = (int(L.ora_rep_key(self.boards[n].h, s["which"])) for n in (s["a"], s["b"]))
            assert (ka == kb) == s["equal"], s
        elif op == "rule50_compare":
            ka, kb = (L.ora_board_rule50(self.boards[n].h, s["which"]) for n in (s["a"], s["b"]))
            assert (ka == kb) == s["equal"], s
        elif op == "expect_board":
            bd, w = self.boards[s["board"]], s["what"]
            if w == "is_draw":
                got = bd.is_draw(s["ply"])
            elif w == "is_checkmate":
                got = bd.is_checkmate(s["side"], s["adv"])
            elif w == "stm":
                got = L.ora_board_stm(bd.h, s["which"])
            elif w == "gives_check":
                got = bool(L.ora_gives_check(bd.h, s["which"], bd.find_move(s["which"], s["uci"])))
            elif w == "hash_key":
                got = bd.hash_key(s["adv"])
            elif w == "count_in_hand":
                got = L.ora_board_count_in_hand(bd.h, s["which"], s["color"], s["piece"])
            elif w == "prefix_len_delta":
                got = L.ora_board_prefix_len(bd.h, s["which"]) - self.remembered[s["since"]][False][4 + s["which"]]
            elif w == "last_move":
                got = L.ora_board_last_move(bd.h, s["which"])
            elif w == "last_move_uci":
                # the move was made from the previous position: compare with the remembered move int instead of its text
                got = s["eq"] if L.ora_board_last_move(bd.h, s["which"]) == self.last_made[s["board"]][s["which"]] != 0 else None
            else:
                raise KeyError(w)
            if "as" in s:                       # remember the value (e.g. a hash key for a node) instead of comparing it
                self.vars[s["as"]] = got
            else:
                want = self.vars[s["eq_var"]] if "eq_var" in s else s["eq"]
                assert got == want, (s, got)
        elif op == "classify":
            bd = self.boards[s["board"]]
            r = self.classify(bd, s["team"], s["root_team"], s["root_adv"], s["ply"])
            assert (r & 0xff) == s["outcome"], (s, r)
            if "end_in_ply" in s:
                assert (r >> 8) == s["end_in_ply"], (s, r)
        elif op == "representable":
            bd = self.boards[s["board"]]
            m = bd.find_move(s["which"], s["uci"])
            assert m, s
            assert bool(L.ora_is_policy_move_representable(m)) == s["eq"], s
        elif op == "planes":
            p = O.planes(self.boards[s["board"]].compact(s["team"], s["adv"]), "f32")[0].reshape(74, 64)
            for ch, sq in s.get("ones", []):
                assert float_eq(p[ch, sq], 1.0), (s, ch, sq)
            for ch in s.get("zero_planes", []):
                assert not p[ch].any(), (s, ch)
        # ---- policy ----
        elif op == "normalize_logits":
            self.vars[s["as"]] = self.normalize_logits(np.array([_num(x) for x in s["logits"]], np.float32))
        elif op == "policy":
            pol = np.full(4672, s.get("fill", 0.0), np.float32)
            if s.get("pattern") == "mod17":
                pol = ((np.arange(4672) % 17).astype(np.float32) - np.float32(8)) * np.float32(0.25)
            for label, v in s.get("set", {}).items():
                idx = L.ora_policy_index_of_label(label.encode())
                assert idx >= 0, label
                pol[idx] = v
            self.vars[s["as"]] = pol
        elif op == "normalized_probability":
            bd = self.boards[s["board"]]
            acts = list(bd.legal_moves(s["which"])) + ([0] if s["with_pass"] else [])
            self.vars["_actions"] = acts
            self.vars[s["as"]] = self.normalized_probability(self.vars[s["policy"]], bool(s.get("half")), acts, L.ora_board_stm(bd.h, s["which"]))
        elif op == "expect_vec":
            v = self.vars[s["vec"]]
            if s.get("all_finite"):
                assert np.isfinite(v).all(), v
            if "sum_near" in s:
                assert abs(float(np.sum(v, dtype=np.float32)) - s["sum_near"][0]) <= s["sum_near"][1], v
            if "gt" in s:
                assert v[s["gt"][0]] > v[s["gt"][1]], v
            if "float_eq" in s:
                assert len(v) == len(s["float_eq"]) and all(float_eq(a, b) for a, b in zip(v, s["float_eq"])), v
            if s.get("size_is_actions"):
                assert len(v) == len(self.vars["_actions"])
            if "last_lt" in s:
                assert v[-1] < s["last_lt"], v[-1]
            if "float_eq_vec" in s:
                w = self.vars[s["float_eq_vec"]]
                assert len(v) == len(w) and all(float_eq(a, b) for a, b in zip(v, w))
            if "action_lt" in s:
                bd = self.boards[s["board"]]
                (ba, ua), (bb, ub) = s["action_lt"]
                ia, ib = self.vars["_actions"].index(bd.find_move(ba, ua)), self.vars["_actions"].index(bd.find_move(bb, ub))
                assert v[ia] < v[ib], (v[ia], v[ib])
        elif op == "generator_first":
            bd = self.boards[s["board"]]
            acts, probs = self.vars["_actions"], self.vars[s["probs"]]
            import test_oracle_search as TS
            pairs, pri = TS.enumerate_gen(acts, s["b"], probs, s["pb"], s["adv"], s["a_on"], s["b_on"], tie=0)
            assert bd.uci(s["which"], pairs[0][0]) == s["expect_uci"], pairs[0]
            want = probs[acts.index(bd.find_move(s["which"], s["expect_prior_of"]))] * np.float32(s["pb"][0])
            assert float_eq(pri[0], want), (pri[0], want)
        elif op == "cpuct_ne":
            assert L.ora_get_cpuct3(*s["a"]) != L.ora_get_cpuct3(*s["b"])
        elif op == "allowed_children":
            assert L.ora_allowed_children3(s["visits"], s["coef"], L.ora_lab_config_get(h, b"pwExponent")) == s["eq"], s
        elif op == "allowed_children_gt":
            e = L.ora_lab_config_get(h, b"pwExponent")
            assert L.ora_allowed_children3(s["visits"], s["coef_a"], e) > L.ora_allowed_children3(s["visits"], s["coef_b"], e)
        elif op == "shape_value":
            self.vars[s["as"]] = float(L.ora_lab_shape_value(h, s["value"], None, s["moves_left"]))
        elif op == "expect_scalar":
            key = "gt" if "gt" in s else "lt"
            a, b = (self.val(x) for x in s[key])
            assert (a > b) if key == "gt" else (a < b), (s, a, b)
        else:
            raise KeyError(op)

    # overridable pieces (the GPU replay swaps these for the product's entry points)
    def classify(self, bd, team, root_team, root_adv, ply):
        return int(L.ora_classify(bd.h, team, root_team, int(root_adv), ply))

    def normalize_logits(self, logits):
        out = np.zeros(len(logits), np.float32)
        L.ora_normalize_logits(logits.ctypes.data, len(logits), self.exp_mode, out.ctypes.data)
        return out

    def normalized_probability(self, policy_f32, half, actions, stm):
        acts = np.array(actions, np.uint32)
        out = np.zeros(len(acts), np.float32)
        if half:
            pol = policy_f32.astype(np.float16).view(np.uint16)
            L.ora_normalized_probability(pol.ctypes.data, 1, acts.ctypes.data, len(acts), stm, self.exp_mode, out.ctypes.data)
        else:
            pol = np.ascontiguousarray(policy_f32, np.float32)
            L.ora_normalized_probability(pol.ctypes.data, 0, acts.ctypes.data, len(acts), stm, self.exp_mode, out.ctypes.data)
        return out
