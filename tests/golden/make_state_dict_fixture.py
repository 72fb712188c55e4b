#!/usr/bin/env python3
"""Writes tests/golden/risev33_state_dict.json — the state_dict key / shape list of the reference's deployed network
(get_rise_v33_model with the arguments of train_loop.get_model_args under the supervised TrainConfig) — WITHOUT importing the
reference: `timm` is absent from this image, so `import src.architectures.rise_mobile_v3` raises ModuleNotFoundError.

The reference files are read as TEXT and parsed with `ast`:
  * the factory's literal arguments come out of the source itself — the kernels / se_types lists of get_rise_v33_model
    (rise_mobile_v3.py:820-849: `kernels = [3] * 15`, `kernels[7] = 5`, ...), the keyword literals of its RiseV3(...) call
    (channels, channels_operating_init, channel_expansion, channels_value_head, value_fc_size), NUM_BUGHOUSE_CHANNELS
    (constants.py:8-13), channels_policy_head (train_loop.py:95-116) and the use_wdl / use_plys_to_end flags the supervised
    run sets (train_loop.py:236-239);
  * the module tree is then spelled out below, one rule per reference module, each rule citing the constructor lines whose
    attribute names and layer order it follows; the script checks that every cited line range still contains the layer
    constructors the rule assumes, so a changed reference fails here instead of silently producing stale keys.

No reference source text is stored in the fixture: it holds names and shapes only.  usage: python tests/golden/make_state_dict_fixture.py
(needs /root/reference; tests/test_host_logic.py compares the committed JSON with hivemind_amd.net.rise_v33().state_dict()).
"""
import ast
import json
import math
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"


def src(rel):
    return open(os.path.join(REF, rel)).read()


def func(tree, name):
    for n in ast.walk(tree):
        if isinstance(n, ast.FunctionDef) and n.name == name:
            return n
    raise KeyError(name)


def lit(node, env):
    """the few expression forms the factory uses: constants, names, [x] * n, len(name)"""
    if isinstance(node, ast.Constant):
        return node.value
    if isinstance(node, ast.Name):
        return env[node.id]
    if isinstance(node, ast.List):
        return [lit(e, env) for e in node.elts]
    if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Mult):
        return lit(node.left, env) * lit(node.right, env)
    if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Add):
        return lit(node.left, env) + lit(node.right, env)
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == "len":
        return len(lit(node.args[0], env))
    raise ValueError(ast.dump(node))


def factory_arguments():
    """-> dict of the literal arguments get_rise_v33_model passes to RiseV3 (rise_mobile_v3.py:820-849)"""
    f = func(ast.parse(src("architectures/rise_mobile_v3.py")), "get_rise_v33_model")
    env, kw = {}, {}
    for st in f.body:
        if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name) and not (isinstance(st.value, ast.Call) and getattr(st.value.func, "id", "") == "RiseV3"):
            env[st.targets[0].id] = lit(st.value, env)
        elif isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Subscript):               # kernels[7] = 5
            env[st.targets[0].value.id][lit(st.targets[0].slice, env)] = lit(st.value, env)
        elif isinstance(st, ast.Assign) and isinstance(st.value, ast.Call) and getattr(st.value.func, "id", "") == "RiseV3":
            for k in st.value.keywords:
                try:
                    kw[k.arg] = lit(k.value, env)
                except (ValueError, KeyError):
                    pass                                                                              # args.* : filled from get_model_args below
    return kw


def constants():
    env = {}
    for st in ast.parse(src("constants.py")).body:
        if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name):
            try:
                env[st.targets[0].id] = lit(st.value, env)
            except (ValueError, KeyError):
                pass
    return env


def model_args():
    """channels_policy_head of train_loop.get_model_args (:95-116) and the head flags of the supervised TrainConfig (:236-239)"""
    text = src("training/train_loop.py")
    out = {"channels_policy_head": int(re.search(r"self\.channels_policy_head\s*=\s*(\d+)", text).group(1))}
    for flag in ("use_wdl", "use_plys_to_end"):
        out[flag] = re.search(rf"tc\.{flag}\s*=\s*True", text) is not None
    out["use_mlp_wdl_ply"] = re.search(r"tc\.use_mlp_wdl_ply\s*=\s*True", text) is not None         # never set: False
    assert re.search(r"self\.shared_policy_trunk\s*=\s*bool\(train_config and train_config\.use_wdl\)", text)
    out["shared_policy_trunk"] = out["use_wdl"]
    return out


def cited(rel, first, last, *needles):
    """the rule below follows these reference lines: they must still hold the constructors it names, in this order"""
    lines = "\n".join(src(rel).split("\n")[first - 1:last])
    pos = 0
    for n in needles:
        k = lines.find(n, pos)
        assert k >= 0, f"{rel}:{first}-{last} no longer contains {n!r} after offset {pos}"
        pos = k + len(n)


def bn(prefix, c):          # torch.nn.BatchNorm2d: weight, bias, running_mean, running_var, num_batches_tracked
    return [(f"{prefix}.weight", [c]), (f"{prefix}.bias", [c]), (f"{prefix}.running_mean", [c]), (f"{prefix}.running_var", [c]),
            (f"{prefix}.num_batches_tracked", [])]


def entries():
    a, m, K = factory_arguments(), model_args(), constants()
    cin = K["NUM_BUGHOUSE_CHANNELS"]
    C, cop, exp = a["channels"], a["channels_operating_init"], a["channel_expansion"]
    kernels, se = a["kernels"], a["se_types"]
    assert len(kernels) == len(se) == 15 and m["use_wdl"] and m["use_plys_to_end"] and not m["use_mlp_wdl_ply"] and m["shared_policy_trunk"]
    e = []
    # RiseV3.body_spatial = Sequential(_Stem, *res_blocks) (rise_mobile_v3.py:179-182)
    cited("architectures/rise_mobile_v3.py", 179, 183, "self.body_spatial = Sequential(", "_Stem(", "*self.res_blocks")
    # _Stem.body = Sequential(Conv2d(nb_input_channels -> channels, 3x3, bias=False), BatchNorm2d, act) (builder_util.py:165-169)
    cited("architectures/builder_util.py", 165, 169, "self.body = Sequential(", "Conv2d(in_channels=nb_input_channels, out_channels=channels, kernel_size=(3, 3)", "bias=False", "BatchNorm2d(")
    e += [("body_spatial.0.body.0.weight", [C, cin, 3, 3])] + bn("body_spatial.0.body.1", C)
    # _get_res_blocks (rise_mobile_v3.py:59-101): channels_operating starts at channels_operating_init and grows by channel_expansion per
    # block; a 5x5 block runs with channels_operating - 32 * (idx // 2) (kernel_5_channel_ratio is None)
    cited("architectures/rise_mobile_v3.py", 59, 101, "channels_operating = channels_operating_init", "if kernel == 5:", "channels_operating - 32 * (idx // 2)",
          "_BottlekneckResidualBlock(channels=channels", "channels_operating += channel_expansion")
    # _BottlekneckResidualBlock (builder_util.py:445-473): optional self.se = get_se(...), then self.body = Sequential(1x1 conv, BN, act,
    # depthwise kxk conv (groups = channels_operating), BN, act, 1x1 conv, BN) -> parameter indices 0, 1, 3, 4, 6, 7
    cited("architectures/builder_util.py", 457, 473, "self.se = get_se(", "self.body = Sequential(Conv2d(in_channels=channels, out_channels=channels_operating, kernel_size=(1, 1), bias=False)",
          "BatchNorm2d(num_features=channels_operating)", "kernel_size=(kernel, kernel)", "groups=groups", "BatchNorm2d(num_features=channels_operating)",
          "Conv2d(in_channels=channels_operating, out_channels=channels, kernel_size=(1, 1), bias=False)", "BatchNorm2d(num_features=channels)")
    # _EfficientChannelAttentionModule (builder_util.py:49-69): self.body = Sequential(Conv1d(channels -> channels, kernel, bias=True), act),
    # kernel = t if t odd else t + 1 with t = int(abs((log2(channels) + 1) / 2))
    cited("architectures/builder_util.py", 49, 69, "t = int(abs((math.log(channels, 2) + b) / gamma))", "kernel = t if t % 2 else t + 1", "self.body = Sequential(",
          "Conv1d(in_channels=channels, out_channels=channels, kernel_size=kernel", "bias=True")
    for idx, k in enumerate(kernels):
        p = f"body_spatial.{idx + 1}"
        co = cop - 32 * (idx // 2) if k == 5 else cop
        if se[idx]:
            assert se[idx] == "eca_se"
            t = int(abs((math.log(C, 2) + 1) / 2))
            ek = t if t % 2 else t + 1
            e += [(f"{p}.se.body.0.weight", [C, C, ek]), (f"{p}.se.body.0.bias", [C])]
        e += [(f"{p}.body.0.weight", [co, C, 1, 1])] + bn(f"{p}.body.1", co)
        e += [(f"{p}.body.3.weight", [co, 1, k, k])] + bn(f"{p}.body.4", co)
        e += [(f"{p}.body.6.weight", [C, co, 1, 1])] + bn(f"{p}.body.7", C)
        cop += exp
    # _ValueHead (builder_util.py:268-301): body = Sequential(Conv2d(channels -> channels_value_head, 1x1, bias=False), BN, act);
    # body_wdl = Sequential(Linear(nb_flatten -> 3)); body_plys = Sequential(Linear(nb_flatten -> 1), sigmoid);
    # body_final = Sequential(Linear(nb_flatten -> fc0), act, Linear(fc0 -> 1), tanh) (the non-mlp branch)
    cited("architectures/builder_util.py", 268, 301, "self.body = Sequential(Conv2d(in_channels=channels, out_channels=channels_value_head, kernel_size=(1, 1), bias=False)",
          "self.nb_flatten = board_height*board_width*channels_value_head", "self.body_wdl = Sequential(Linear(", "out_features=3",
          "self.body_plys = Sequential(Linear(", "out_features=1", "else:", "self.body_final = Sequential(Linear(", "out_features=fc0", "Linear(in_features=fc0, out_features=1)")
    cv, fc = a["channels_value_head"], a["value_fc_size"]
    flat = 8 * 8 * cv
    e += [("value_head.body.0.weight", [cv, C, 1, 1])] + bn("value_head.body.1", cv)
    e += [("value_head.body_wdl.0.weight", [3, flat]), ("value_head.body_wdl.0.bias", [3]),
          ("value_head.body_plys.0.weight", [1, flat]), ("value_head.body_plys.0.bias", [1]),
          ("value_head.body_final.0.weight", [fc, flat]), ("value_head.body_final.0.bias", [fc]),
          ("value_head.body_final.2.weight", [1, fc]), ("value_head.body_final.2.bias", [1])]
    # _SharedPolicyHeads (rise_mobile_v3.py:36-49): shared_body = Sequential(Conv2d(C -> C, 3x3, bias=False), BN, act);
    # board_projections = ModuleList([Conv2d(C -> policy_channels, 3x3, bias=False)] x 2)
    cited("architectures/rise_mobile_v3.py", 36, 49, "self.shared_body = Sequential(", "Conv2d(channels, channels, kernel_size=3, padding=1, bias=False)", "BatchNorm2d(channels)",
          "self.board_projections = nn.ModuleList([", "Conv2d(channels, policy_channels, kernel_size=3, padding=1, bias=False)", "Conv2d(channels, policy_channels, kernel_size=3, padding=1, bias=False)")
    pc = m["channels_policy_head"]
    e += [("policy_heads.shared_body.0.weight", [C, C, 3, 3])] + bn("policy_heads.shared_body.1", C)
    e += [("policy_heads.board_projections.0.weight", [pc, C, 3, 3]), ("policy_heads.board_projections.1.weight", [pc, C, 3, 3])]
    return e


if __name__ == "__main__":
    ent = entries()
    trainable = sum(math.prod(s) for k, s in ent if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert trainable == 14122085, trainable          # counted on the imported reference model during the survey
    out = os.path.join(HERE, "risev33_state_dict.json")
    old = json.load(open(out)) if os.path.exists(out) else None
    doc = dict(source="generated by tests/golden/make_state_dict_fixture.py from the reference sources read as text: factory literals of "
                      "src/architectures/rise_mobile_v3.py:820-849 + src/training/train_loop.py:95-116,236-239 + src/constants.py:8-13; module attribute "
                      "paths of rise_mobile_v3.py:36-49,59-101,179-205 and builder_util.py:49-69,165-169,268-301,445-473 (each rule checks its cited lines); "
                      "14 122 085 trainable parameters (SURVEY §6, counted on the imported reference model)",
               trainable_parameters=trainable, entries=dict((k, s) for k, s in ent))
    json.dump(doc, open(out, "w"), indent=0)
    print(f"wrote {len(ent)} entries, {trainable} trainable parameters to {out}")
    if old is not None:
        print("entries identical to the previous fixture" if old["entries"] == doc["entries"] and list(old["entries"]) == list(doc["entries"]) else "ENTRIES CHANGED")
