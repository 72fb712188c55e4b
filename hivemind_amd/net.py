"""RISEv3 policy/value network under PyTorch-ROCm (the evaluator behind the Engine seam).

Architecture follows the reference definition (src/architectures/rise_mobile_v3.py:105-230, 820-849;
builder_util.py:49-80, 154-178, 246-326, 445-483) with the factory arguments of
src/training/train_loop.py:95-116 (shared policy trunk, WDL + plys-to-end heads).  Module and
parameter names match the reference so a reference ``state_dict`` loads unchanged
(``RiseV3.load_state_dict``).  ``InferenceNet`` is the deployed form: BatchNorm folded into the
convolutions, ECA's length-1 Conv1d reduced to its centre tap, channels-last fp16, optional HIP-graph
replay; it returns exactly the tensors the reference's TensorRT engine exposes
(nn/engine.cc:444-461): value, pi_a, pi_b, wdl (loss, draw, win logits), moves_left.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

NB_INPUT_CHANNELS, POLICY_CHANNELS = 74, 73


class _Stem(nn.Module):
    def __init__(self, channels, nb_input_channels):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(nb_input_channels, channels, 3, padding=1, bias=False),
                                  nn.BatchNorm2d(channels), nn.ReLU())

    def forward(self, x):
        return self.body(x)


class _ECA(nn.Module):
    """_EfficientChannelAttentionModule with hard-sigmoid (builder_util.py:49-80, :460)."""

    def __init__(self, channels, gamma=2, b=1):
        super().__init__()
        t = int(abs((math.log(channels, 2) + b) / gamma))
        k = t if t % 2 else t + 1
        self.body = nn.Sequential(nn.Conv1d(channels, channels, k, padding=k // 2, bias=True), nn.Hardsigmoid())

    def forward(self, x):
        n, c = x.shape[:2]
        g = self.body(x.mean(dim=(2, 3)).view(n, c, 1)).view(n, c, 1, 1)
        return x * g


class _Bottleneck(nn.Module):
    """_BottlekneckResidualBlock (builder_util.py:445-483): SE gates the block INPUT."""

    def __init__(self, channels, channels_operating, kernel, se):
        super().__init__()
        self.se_type = "eca_se" if se else None
        if se:
            self.se = _ECA(channels)
        self.body = nn.Sequential(
            nn.Conv2d(channels, channels_operating, 1, bias=False), nn.BatchNorm2d(channels_operating), nn.ReLU(),
            nn.Conv2d(channels_operating, channels_operating, kernel, padding=kernel // 2, bias=False, groups=channels_operating),
            nn.BatchNorm2d(channels_operating), nn.ReLU(),
            nn.Conv2d(channels_operating, channels, 1, bias=False), nn.BatchNorm2d(channels))

    def forward(self, x):
        if self.se_type:
            x = self.se(x)
        return x + self.body(x)


class _ValueHead(nn.Module):
    def __init__(self, channels, channels_value_head, fc0):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(channels, channels_value_head, 1, bias=False),
                                  nn.BatchNorm2d(channels_value_head), nn.ReLU())
        self.nb_flatten = 64 * channels_value_head
        self.body_wdl = nn.Sequential(nn.Linear(self.nb_flatten, 3))
        self.body_plys = nn.Sequential(nn.Linear(self.nb_flatten, 1), nn.Sigmoid())
        # present in the reference module (unused when use_mlp_wdl_ply is False); kept for state_dict parity
        self.body_final = nn.Sequential(nn.Linear(self.nb_flatten, fc0), nn.ReLU(), nn.Linear(fc0, 1), nn.Tanh())

    def forward(self, x):
        x = self.body(x).reshape(-1, self.nb_flatten)
        wdl = self.body_wdl(x)
        plys = self.body_plys(x)
        sm = torch.softmax(wdl, dim=1)
        return sm[:, 2:3] - sm[:, 0:1], wdl, plys


class _SharedPolicyHeads(nn.Module):
    def __init__(self, channels, policy_channels):
        super().__init__()
        self.shared_body = nn.Sequential(nn.Conv2d(channels, channels, 3, padding=1, bias=False),
                                         nn.BatchNorm2d(channels), nn.ReLU())
        self.board_projections = nn.ModuleList([nn.Conv2d(channels, policy_channels, 3, padding=1, bias=False),
                                                nn.Conv2d(channels, policy_channels, 3, padding=1, bias=False)])

    def forward(self, x):
        s = self.shared_body(x)
        return tuple(p(s).reshape(x.shape[0], -1) for p in self.board_projections)


class RiseV3(nn.Module):
    def __init__(self, channels=384, channels_operating_init=256, channel_expansion=64, kernels=None, se=None,
                 channels_value_head=16, value_fc_size=512):
        super().__init__()
        kernels = kernels or [3] * 15
        se = se or [False] * len(kernels)
        blocks, cop = [], channels_operating_init
        for idx, k in enumerate(kernels):
            active = cop - 32 * (idx // 2) if k == 5 else cop          # rise_mobile_v3.py:68-72
            blocks.append(_Bottleneck(channels, active, k, se[idx]))
            cop += channel_expansion
        self.body_spatial = nn.Sequential(_Stem(channels, NB_INPUT_CHANNELS), *blocks)
        self.value_head = _ValueHead(channels, channels_value_head, value_fc_size)
        self.policy_heads = _SharedPolicyHeads(channels, POLICY_CHANNELS)
        self.config = dict(channels=channels, kernels=list(kernels))

    def forward(self, x):
        out = self.body_spatial(x)
        value, wdl, plys = self.value_head(out)
        pi_a, pi_b = self.policy_heads(out)
        return value, pi_a, pi_b, wdl, plys


def rise_v33() -> RiseV3:
    """get_rise_v33_model (rise_mobile_v3.py:820-849): 15 blocks, 384 ch, 14 122 085 parameters."""
    kernels = [3] * 15
    for i in (7, 11, 12, 13):
        kernels[i] = 5
    se = [False] * 15
    for i in (5, 8, 12, 13, 14):
        se[i] = True
    return RiseV3(384, 256, 64, kernels, se, 16, 512)


def rise_v3_small() -> RiseV3:
    """"RISEv3-small" of BASELINE.json configs[2].  The reference has no model of that name
    (SURVEY.md §8d); defined here as the same RiseV3 class with 128 channels and six 3x3 blocks
    (ECA on the last two)."""
    return RiseV3(128, 128, 32, [3] * 6, [False, False, False, False, True, True], 8, 256)


def load_checkpoint(path, model: RiseV3 = None, map_location="cpu") -> RiseV3:
    """Trained-weight ingestion: the reference trainer's checkpoint file, written by save_torch_state
    (src/training/trainer_agent.py:871-890) as torch.save({'model_state_dict', 'optimizer_state_dict',
    ['training_iteration', 'evaluation_step', 'batch_steps']}) and restored by restore_torch_state (:866-868) with a strict
    load_state_dict.  A bare state_dict is accepted too; 'module.' / '_orig_mod.' prefixes (DataParallel, torch.compile) are
    stripped.  Returns the model in eval mode, ready for InferenceNet / FusedNet."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    sd = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    if not isinstance(sd, dict) or not sd:
        raise ValueError(f"{path}: no model_state_dict")
    clean = {}
    for k, v in sd.items():
        for pre in ("module.", "_orig_mod."):
            while k.startswith(pre):
                k = k[len(pre):]
        clean[k] = v
    if model is None:
        ch = clean["body_spatial.0.body.0.weight"].shape[0]
        model = rise_v33() if ch == 384 else rise_v3_small()
    model.load_state_dict(clean, strict=True)
    return model.eval()


def flops_per_position(model: RiseV3) -> float:
    """Multiply-accumulate count x2 of the conv / linear layers at 8x8."""
    total = 0
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            total += 2 * 64 * m.out_channels * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]
        elif isinstance(m, nn.Conv1d):
            total += 2 * m.out_channels * m.in_channels          # centre tap only on a length-1 sequence
        elif isinstance(m, nn.Linear):
            total += 2 * m.in_features * m.out_features
    fin = model.value_head.body_final
    total -= 2 * (fin[0].in_features * fin[0].out_features + fin[2].in_features * fin[2].out_features)
    return float(total)


def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    w = conv.weight.detach().float()
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    return w * scale.view(-1, 1, 1, 1), bn.bias.detach().float() - bn.running_mean.detach().float() * scale


class InferenceNet(nn.Module):
    """Deployed form of a RiseV3 (eval mode): folded BN, fp16 channels-last, fixed output contract."""

    def __init__(self, model: RiseV3, dtype=torch.float16, device="cuda"):
        super().__init__()
        model = model.eval()
        self.dtype = dtype
        mk = lambda w, b, **kw: (w.to(device=device, dtype=dtype).contiguous(memory_format=torch.channels_last),
                                 b.to(device=device, dtype=dtype), kw)
        stem = model.body_spatial[0].body
        self.stem = mk(*_fold(stem[0], stem[1]), padding=1)
        self.blocks = []
        for blk in list(model.body_spatial)[1:]:
            b = blk.body
            eca = None
            if blk.se_type:
                c1 = blk.se.body[0]
                k = c1.kernel_size[0]
                eca = (c1.weight.detach()[:, :, k // 2].to(device=device, dtype=dtype).contiguous(),
                       c1.bias.detach().to(device=device, dtype=dtype))
            self.blocks.append(dict(
                eca=eca,
                c1=mk(*_fold(b[0], b[1])),
                dw=mk(*_fold(b[3], b[4]), padding=b[3].padding[0], groups=b[3].groups),
                c2=mk(*_fold(b[6], b[7]))))
        vh = model.value_head
        self.vconv = mk(*_fold(vh.body[0], vh.body[1]))
        # fused value-head linear: rows 0..2 = wdl logits, row 3 = plys logit; NCHW flatten order
        # (reference .view(-1, nb_flatten) on an NCHW tensor) is kept by permuting the weight instead
        # of the activation: channels-last activations flatten as (h, w, c).
        wl = torch.cat([vh.body_wdl[0].weight.detach(), vh.body_plys[0].weight.detach()], 0).float()
        cv = vh.body[0].out_channels
        wl = wl.view(4, cv, 64).permute(0, 2, 1).reshape(4, 64 * cv)
        self.vlin_w = wl.to(device=device, dtype=dtype).contiguous()
        self.vlin_b = torch.cat([vh.body_wdl[0].bias.detach(), vh.body_plys[0].bias.detach()]).to(device=device, dtype=dtype)
        ph = model.policy_heads
        self.pshared = mk(*_fold(ph.shared_body[0], ph.shared_body[1]), padding=1)
        wp = torch.cat([ph.board_projections[0].weight.detach(), ph.board_projections[1].weight.detach()], 0).float()
        self.pproj = mk(wp, torch.zeros(wp.shape[0]), padding=1)
        self._graph = None
        self._static_in = None
        self._static_out = None

    @staticmethod
    def _conv(x, p, relu):
        w, b, kw = p
        y = F.conv2d(x, w, b, **kw)
        return F.relu_(y) if relu else y

    def forward(self, planes: torch.Tensor):
        x = planes.to(self.dtype).contiguous(memory_format=torch.channels_last)
        x = self._conv(x, self.stem, True)
        for blk in self.blocks:
            if blk["eca"] is not None:
                w, b = blk["eca"]
                g = F.hardsigmoid(F.linear(x.mean(dim=(2, 3)), w, b))
                x = x * g[:, :, None, None]
            y = self._conv(x, blk["c1"], True)
            y = self._conv(y, blk["dw"], True)
            y = self._conv(y, blk["c2"], False)
            x = x + y
        n = x.shape[0]
        v = self._conv(x, self.vconv, True)
        v = v.permute(0, 2, 3, 1).reshape(n, -1)                     # (h, w, c) flatten, matches vlin_w
        lin = F.linear(v, self.vlin_w, self.vlin_b).float()
        wdl = lin[:, :3]
        sm = torch.softmax(wdl, dim=1)
        value = (sm[:, 2] - sm[:, 0])
        plys = torch.sigmoid(lin[:, 3])
        s = self._conv(x, self.pshared, True)
        p = self._conv(s, self.pproj, False)                          # [n, 146, 8, 8] channels-last
        p = p.contiguous(memory_format=torch.contiguous_format)      # plane-major [73][64] per head
        pi_a = p[:, :POLICY_CHANNELS].reshape(n, -1)
        pi_b = p[:, POLICY_CHANNELS:].reshape(n, -1)
        h = torch.float16
        return (value.to(h).contiguous(), pi_a.to(h).contiguous(), pi_b.to(h).contiguous(),
                wdl.to(h).contiguous(), plys.to(h).contiguous())

    # ---- HIP-graph replay for a fixed batch (launch-bound at small batch) -----------------
    def capture(self, batch: int):
        self._static_in = torch.zeros((batch, NB_INPUT_CHANNELS, 8, 8), dtype=torch.float16, device=self.vlin_w.device)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(3):
                self.forward(self._static_in)
        torch.cuda.current_stream().wait_stream(s)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph), torch.no_grad():
            self._static_out = self.forward(self._static_in)
        return self

    @torch.no_grad()
    def __call__(self, planes: torch.Tensor):
        if self._graph is not None and planes.shape == self._static_in.shape:
            self._static_in.copy_(planes)
            self._graph.replay()
            return self._static_out
        return self.forward(planes)


# ---------------------------------------------------------------------------------------------
# Fused single-launch forward (hivemind_amd/csrc/hm_net.hip)
# ---------------------------------------------------------------------------------------------
def _pack_frag(wt: torch.Tensor) -> torch.Tensor:
    """[K, N] (K % 16 == 0, N % 32 == 0) -> v_mfma_f32_32x32x16_f16 fragment order
    [kstep][tile][lane][8]: lane l holds W[kstep*16 + 8*(l>>5) + j][tile*32 + (l&31)]."""
    k, n = wt.shape
    assert k % 16 == 0 and n % 32 == 0, (k, n)
    x = wt.reshape(k // 16, 2, 8, n // 32, 32)          # ks, h, j, t, c
    return x.permute(0, 3, 1, 4, 2).contiguous().reshape(-1)   # ks, t, h, c, j


class FusedNet:
    """RiseV3 (eval) -> one hm_net_forward launch per batch.  Same output contract as InferenceNet."""

    MAXB, CIN_PAD = 16, 80

    def __init__(self, model: RiseV3, device="cuda"):
        import ctypes as C
        import numpy as np
        from ._lib import lib, check
        import copy
        self._C, self._lib, self._check = C, lib, check
        model = copy.deepcopy(model).cpu().float().eval()       # packing happens on the host
        halfs, floats = [], []
        hoff, foff = [0], [0]

        def addh(t):
            t = t.detach().float().reshape(-1)
            pad = (-t.numel()) % 8
            if pad:
                t = torch.cat([t, torch.zeros(pad)])
            off = hoff[0]
            halfs.append(t)
            hoff[0] += t.numel()
            return off

        def addf(t):
            t = t.detach().float().reshape(-1)
            off = foff[0]
            floats.append(t)
            foff[0] += t.numel()
            return off

        def conv3_wt(w, cin_pad=None):          # [co][ci][3][3] -> Wt[(ky*3+kx)*cin + ci][co]
            co, ci = w.shape[:2]
            if cin_pad and cin_pad != ci:
                w = torch.cat([w, torch.zeros(co, cin_pad - ci, 3, 3)], 1)
                ci = cin_pad
            return w.permute(2, 3, 1, 0).reshape(9 * ci, co)

        def padn(wt, n):
            return wt if wt.shape[1] == n else torch.cat([wt, torch.zeros(wt.shape[0], n - wt.shape[1])], 1)

        stem = model.body_spatial[0].body
        w, b = _fold(stem[0], stem[1])
        ch = w.shape[0]
        assert ch % 64 == 0
        desc = dict(C=ch, cin_pad=self.CIN_PAD)
        desc["stem_w"] = addh(_pack_frag(conv3_wt(w, self.CIN_PAD)))
        desc["stem_b"] = addf(b)
        blocks = []
        for blk in list(model.body_spatial)[1:]:
            body = blk.body
            w1, b1 = _fold(body[0], body[1])
            wd, b2 = _fold(body[3], body[4])
            w2, b3 = _fold(body[6], body[7])
            cop, k = w1.shape[0], wd.shape[-1]
            assert cop % 32 == 0 and k in (3, 5)
            d = dict(cop=cop, k=k, eca=0, ecaw=0, ecab=0)
            d["w1"] = addh(_pack_frag(w1.reshape(cop, ch).t().contiguous()))
            d["b1"] = addf(b1)
            d["dw"] = addh(wd.reshape(cop, k * k))
            d["b2"] = addf(b2)
            d["w2"] = addh(_pack_frag(w2.reshape(ch, cop).t().contiguous()))
            d["b3"] = addf(b3)
            if blk.se_type:
                c1 = blk.se.body[0]
                kk = c1.kernel_size[0]
                d["eca"] = 1
                d["ecaw"] = addh(c1.weight.detach()[:, :, kk // 2].t().contiguous())     # [ci][co]
                d["ecab"] = addf(c1.bias)
            blocks.append(d)
        assert len(blocks) <= self.MAXB
        vh = model.value_head
        wv, bv = _fold(vh.body[0], vh.body[1])
        cv = wv.shape[0]
        assert cv <= 32
        desc["cv"] = cv
        desc["v_w"] = addh(_pack_frag(padn(wv.reshape(cv, ch).t().contiguous(), 32)))
        desc["v_b"] = addf(torch.cat([bv, torch.zeros(32 - cv)]))
        desc["vl_w"] = addh(torch.cat([vh.body_wdl[0].weight, vh.body_plys[0].weight], 0))
        desc["vl_b"] = addf(torch.cat([vh.body_wdl[0].bias, vh.body_plys[0].bias]))
        ph = model.policy_heads
        ws, bs = _fold(ph.shared_body[0], ph.shared_body[1])
        desc["ps_w"] = addh(_pack_frag(conv3_wt(ws)))
        desc["ps_b"] = addf(bs)
        wp = torch.cat([ph.board_projections[0].weight.detach().float(), ph.board_projections[1].weight.detach().float()], 0)
        desc["pp_w"] = addh(_pack_frag(padn(conv3_wt(wp), 160)))
        ints = [desc["C"], len(blocks), desc["cv"], desc["cin_pad"], desc["stem_w"], desc["ps_w"], desc["pp_w"], desc["v_w"],
                desc["vl_w"], desc["stem_b"], desc["ps_b"], desc["v_b"], desc["vl_b"]]
        for i in range(self.MAXB):
            if i < len(blocks):
                d = blocks[i]
                ints += [d["cop"], d["k"], d["eca"], 0, d["w1"], d["dw"], d["w2"], d["ecaw"], d["b1"], d["b2"], d["b3"], d["ecab"]]
            else:
                ints += [0] * 12
        self.desc = np.asarray(ints, dtype=np.int32)
        self.wh = torch.cat(halfs).to(device=device, dtype=torch.float16).contiguous()
        self.wf = torch.cat(floats).to(device=device, dtype=torch.float32).contiguous()
        self.device = self.wh.device
        self._out = {}
        h = C.c_void_p()
        check(lib.hm_net_create(self.desc.ctypes.data, self.desc.size, self.wh.data_ptr(), self.wf.data_ptr(), C.byref(h)))
        self.handle = h            # hm_net*: descriptor on the device, launch geometry resolved once

    def close(self):
        if getattr(self, "handle", None):
            self._lib.hm_net_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def save(self, path: str):
        """Packed-network file (the plan-file counterpart Engine::loadNetwork reads back, nn/engine.cc:290-401)."""
        wh = self.wh.cpu().numpy()
        wf = self.wf.cpu().numpy()
        self._check(self._lib.hm_net_save_file(str(path).encode(), self.desc.ctypes.data, self.desc.size, wh.ctypes.data, wh.nbytes,
                                               wf.ctypes.data, wf.nbytes))

    def _buffers(self, n):
        if n not in self._out:
            f = dict(dtype=torch.float16, device=self.device)
            self._out[n] = (torch.empty(n, **f), torch.empty((n, POLICY_CHANNELS * 64), **f), torch.empty((n, POLICY_CHANNELS * 64), **f),
                            torch.empty((n, 3), **f), torch.empty(n, **f))
        return self._out[n]

    def __call__(self, planes: torch.Tensor, out=None, group_rows=None, group=8):
        """out = (value[n], pi_a[n,4672], pi_b[n,4672], wdl[n,3], moves_left[n]) fp16 tensors to fill in place.
        group_rows (int32 CUDA tensor [n / group], from SearchEngine.collect): only the first group_rows[g] rows of
        each group of `group` rows are evaluated; the other rows' outputs are left as they are."""
        if planes.dtype != torch.float16 or not planes.is_contiguous():
            planes = planes.to(torch.float16).contiguous()
        n = planes.shape[0]
        v, a, b, w, m = out if out is not None else self._buffers(n)
        st = torch.cuda.current_stream().cuda_stream
        if group_rows is None:
            self._check(self._lib.hm_net_forward(self.handle, planes.data_ptr(), n, v.data_ptr(), a.data_ptr(), b.data_ptr(), w.data_ptr(),
                                                 m.data_ptr(), self._C.c_void_p(st)))
        else:
            if group_rows.dtype != torch.int32 or group_rows.numel() * group != n or group_rows.device.type != "cuda":
                raise ValueError("group_rows must be an int32 CUDA tensor with n / group entries")
            self._check(self._lib.hm_net_forward_groups(self.handle, planes.data_ptr(), n, group_rows.data_ptr(), group, v.data_ptr(), a.data_ptr(),
                                                        b.data_ptr(), w.data_ptr(), m.data_ptr(), self._C.c_void_p(st)))
        return v, a, b, w, m
