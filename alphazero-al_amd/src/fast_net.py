"""Inference twin of the Connect4 network, laid out for MI355X.

Computes exactly what Connect4/Network.py::CNN.forward computes (same weights, same maths),
but arranged so that a 32 768-leaf batch costs a handful of well-shaped kernels instead of
~60 small ones:

* activations stay in ONE layout - tokens (B, 42, C), which is channels-last for the 3x3
  convolutions - so the NCHW<->NHWC transposes and the flatten/transpose before attention
  disappear (MIOpen's NHWC bf16 implicit-GEMM kernels run on the buffers as they are);
* weights are cast to the compute dtype once, not on every call by autocast;
* the per-head gate projection (N=4) rides in the QKV GEMM (N=192+4), and the three N=1
  linears (row gate, policy logit, moves-left) are dot products, not GEMMs;
* GroupNorm(1, C) is LayerNorm over the sample's 42*C values followed by the channel affine,
  RMSNorm uses the fused kernel (weight in the activation dtype).

Statistics of the normalisations and the softmaxes accumulate in fp32 as they do under the
reference's bf16 autocast.  `FastConnect4Net.from_module(net)` snapshots any module with the
reference's parameter names (the reference CNN itself or az_net.Connect4Net).
"""
import ctypes as C
import os

import torch
import torch.nn.functional as F

ROWS, COLS, CELLS = 6, 7, 42

_GLUE = None


def glue():
    """libaz_mcts.so's fused glue kernels (include/az_nn.h), or None when unavailable."""
    global _GLUE
    if _GLUE is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib", "libaz_mcts.so")
        try:
            L = C.CDLL(path)
            vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
            L.az_nn_embed.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp, vp, vp]
            L.az_nn_groupnorm1.argtypes = [vp, vp, vp, vp, i64, i32, f32, vp]
            L.az_nn_silu_add.argtypes = [vp, vp, i32, vp, vp, i64, vp]
            L.az_nn_rmsnorm64.argtypes = [vp, vp, vp, i64, f32, vp]
            L.az_nn_qkv_prep.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i64, f32, vp]
            L.az_nn_attn_post.argtypes = [vp, vp, vp, i64, vp]
            L.az_nn_heads_prep.argtypes = [vp, vp, vp, f32, vp, vp, i64, f32, vp]
            L.az_nn_conv_block.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, i64, f32, vp, vp]
            L.az_nn_conv_block2.argtypes = [vp, vp, vp, vp, vp, i64, f32, vp, vp]
            L.az_nn_attn_block.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, f32, vp, vp]
            L.az_nn_heads.argtypes = [vp, C.POINTER(HeadsWeights), vp, vp, vp, vp, i64, f32, vp, vp, vp]
            L.az_nn_stem_embed.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp]
            L.az_nn_stem_folded.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp]
            L.az_nn_model_create.argtypes = [C.POINTER(ModelWeights), C.POINTER(vp)]
            L.az_nn_model_destroy.argtypes = [vp]
            L.az_nn_model_destroy.restype = None
            L.az_nn_model_scratch_bytes.argtypes = [vp, i64]
            L.az_nn_model_scratch_bytes.restype = C.c_uint64
            L.az_nn_model_forward.argtypes = [vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, C.c_uint64, vp]
            _GLUE = L
        except OSError:
            _GLUE = False
    return _GLUE or None


def fold_block(w, bias, gamma, beta):
    """What az_nn_conv_block2 (include/az_nn.h) takes instead of (weight, bias, gamma, beta) of a residual block
    y = x + silu(conv(GroupNorm1(x) * gamma + beta) + bias): the weight with gamma folded in and rounded to bf16 once
    (OHWI), t1 (9, 64) = per border class the sum of those ROUNDED weights over the taps inside the board,
    t2_scaled (9, 64) = log2(e) * (bias + the same sum of weight * beta).  Border class = 3 * rowclass + colclass,
    0 = first row / column (the tap at -1 falls off the board), 1 = inner, 2 = last."""
    w32 = w.float().contiguous()                                    # (O, I, 3, 3), the bf16 values the first kernel multiplies
    wf = (w32 * gamma.float().view(1, -1, 1, 1)).to(torch.bfloat16)
    wff = wf.float()
    wb = w32 * beta.float().view(1, -1, 1, 1)
    t1 = torch.zeros((9, w.shape[0]), dtype=torch.float32, device=w.device)
    t2 = torch.zeros_like(t1)
    taps = {0: (1, 2), 1: (0, 1, 2), 2: (0, 1)}
    for rc in range(3):
        for cc in range(3):
            kh, kw = list(taps[rc]), list(taps[cc])
            t1[3 * rc + cc] = wff[:, :, kh][:, :, :, kw].sum((1, 2, 3))
            t2[3 * rc + cc] = bias.float() + wb[:, :, kh][:, :, :, kw].sum((1, 2, 3))
    return (wf.contiguous(memory_format=torch.channels_last), t1.contiguous(), (t2 * 1.4426950408889634).contiguous())


def fold_stem(weight, bias, emb_own, emb_opp, pos_map):
    """What az_nn_stem_folded (include/az_nn.h, nn_stem.hip) takes instead of the embedding tables and the stem's weight:
    the tokens are own * e_own + opp * e_opp + pos with own / opp in {0, 1} (Network.py:226-239) and the convolution is
    linear, so  conv(tokens)[o, cell] + bias = pmap[cell][o] + sum over the taps inside the board of
    own[n] * T[o][2 tap] + opp[n] * T[o][2 tap + 1].  weight (O, E, 3, 3) and bias are rounded to bf16 as the reference's
    autocast rounds them; the embeddings stay fp32 (the reference adds them in fp32).  Returns (w_frag bf16 [2][4][64][8]:
    T split into a bf16 high and low part in MFMA fragment order, pmap float32 [48][68])."""
    w = weight.to(torch.bfloat16).float()                           # (64, E, 3, 3)
    out_c = w.shape[0]
    assert out_c == 64 and tuple(w.shape[2:]) == (3, 3) and pos_map.shape[0] == CELLS
    a = torch.einsum("ocyx,c->oyx", w, emb_own.float()).reshape(out_c, 9)
    b = torch.einsum("ocyx,c->oyx", w, emb_opp.float()).reshape(out_c, 9)
    t = torch.zeros((out_c, 32), dtype=torch.float32, device=w.device)
    t[:, 0:18:2] = a
    t[:, 1:18:2] = b
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    lane = torch.arange(64, device=w.device)
    # row r of channel tile i is channel 32 (i // 2) + 8 (r // 4) + 4 (i % 2) + r % 4: a lane's rows of two neighbouring
    # tiles are eight consecutive channels (one 16-byte store)
    ti, r = torch.arange(4, device=w.device).view(4, 1), (lane & 15).view(1, 64)
    rows = 32 * (ti // 2) + 8 * (r // 4) + 4 * (ti % 2) + r % 4                                     # (4, 64): channel of [tile][lane]
    cols = ((lane >> 4) * 8).view(64, 1) + torch.arange(8, device=w.device).view(1, 8)              # (64, 8): k of [lane][j]
    frag = torch.stack([part[rows.view(4, 64, 1), cols.view(1, 64, 8)] for part in (hi, lo)]).contiguous()    # (2, 4, 64, 8)
    pm = torch.nn.functional.conv2d(pos_map.float().t().reshape(1, -1, 6, COLS), w, bias.to(torch.bfloat16).float(), padding=1)
    pmap = torch.zeros((48, 68), dtype=torch.float32, device=w.device)
    pmap[:CELLS, :out_c] = pm.reshape(out_c, CELLS).t()
    return frag, pmap.contiguous()


class HeadsWeights(C.Structure):
    """az_nn_heads_weights of include/az_nn.h"""
    _PTRS = ("p_norm", "p_gate_w", "p_fc_w", "p_fc_b", "p_out_w", "d_pool_norm", "d_pool_w", "d_pool_b", "d_norm",
             "d_fc_w", "d_fc_b", "d_out_norm", "d_val_w", "d_val_b", "d_aux_w")
    _fields_ = [(n, C.c_void_p) for n in _PTRS] + [(n, C.c_float) for n in ("p_gate_b", "p_out_b", "d_aux_b", "aux_scale")]


MAX_BLOCKS = 8          # AZ_NN_MAX_BLOCKS


class ModelWeights(C.Structure):
    """az_nn_model_weights of include/az_nn.h"""
    _fields_ = ([(n, C.c_void_p) for n in ("emb_own", "emb_opp", "pos", "stem_w", "stem_b")] + [("n_blocks", C.c_int32)] +
                [(n, C.c_void_p * MAX_BLOCKS) for n in ("block_w", "block_b", "block_gamma", "block_beta")] +
                [(n, C.c_void_p) for n in ("pre_w", "qkvg_w", "qn_w", "kn_w", "o_w")] +
                [("heads", HeadsWeights), ("eps", C.c_float), ("stem_frag", C.c_void_p), ("stem_pmap", C.c_void_p)])


class FastConnect4Net(torch.nn.Module):
    aux_target_offset = 42
    n_actions = COLS

    def __init__(self, state_dict, device, dtype=torch.bfloat16, heads=4):
        super().__init__()
        self.device = torch.device(device)
        self.dtype = dtype
        self.heads = heads
        sd = {k: v.detach().to(self.device) for k, v in state_dict.items()}
        c = lambda t: t.to(dtype).contiguous()   # noqa: E731
        f = lambda t: t.to(torch.float32).contiguous()   # noqa: E731

        self.embed_dim = sd["piece_emb.weight"].shape[1]
        self.register_buffer("emb_own", c(sd["piece_emb.weight"][0]))
        self.register_buffer("emb_opp", c(sd["piece_emb.weight"][1]))
        self.register_buffer("pos", c(sd["pos_emb.weight"][sd["orbit_map"].long()]))       # (42, E)

        def conv_w(w):      # OIHW -> channels_last weight
            return c(w).contiguous(memory_format=torch.channels_last)
        self.register_buffer("stem_w", conv_w(sd["hidden.0.weight"]))
        self.register_buffer("stem_b", c(sd["hidden.0.bias"]))
        # the stem as a K = 18 GEMM on the 0/1 planes (nn_stem.hip); AZ_STEM_FOLDED=0: embedding + K = 288 convolution
        self.folded_stem = os.environ.get("AZ_STEM_FOLDED", "1") != "0" and sd["hidden.0.weight"].shape[0] == 64
        if self.folded_stem:
            frag, pmap = fold_stem(sd["hidden.0.weight"], sd["hidden.0.bias"], sd["piece_emb.weight"][0], sd["piece_emb.weight"][1],
                                   sd["pos_emb.weight"][sd["orbit_map"].long()])
            self.register_buffer("stem_frag", frag)
            self.register_buffer("stem_pmap", pmap)
        self.h_dim = sd["hidden.0.weight"].shape[0]
        self.res = []
        i = 2
        while f"hidden.{i}.conv.weight" in sd:
            names = (f"res{i}_w", f"res{i}_b", f"res{i}_g", f"res{i}_beta")
            self.register_buffer(names[0], conv_w(sd[f"hidden.{i}.conv.weight"]))
            self.register_buffer(names[1], c(sd[f"hidden.{i}.conv.bias"]))
            self.register_buffer(names[2], c(sd[f"hidden.{i}.norm.weight"]))
            self.register_buffer(names[3], c(sd[f"hidden.{i}.norm.bias"]))
            self.res.append(names)
            i += 1
        a = f"hidden.{i}.attn."
        self.register_buffer("pre_w", c(sd[a + "prenorm.weight"]))
        qkvg = torch.cat([sd[a + "qkv_proj.weight"], sd[a + "gate_proj.weight"]], 0)
        self.register_buffer("qkvg_w", c(qkvg))
        self.register_buffer("o_w", c(sd[a + "o_proj.weight"]))
        self.register_buffer("qn_w", c(sd[a + "q_norm.weight"]))
        self.register_buffer("kn_w", c(sd[a + "k_norm.weight"]))
        p = "policy_head."
        self.register_buffer("p_norm", c(sd[p + "norm.weight"]))
        self.register_buffer("p_gate_w", c(sd[p + "row_gate.weight"].reshape(-1)))
        self.register_buffer("p_gate_b", f(sd[p + "row_gate.bias"]))
        self.register_buffer("p_fc_w", c(sd[p + "fc.weight"]))
        self.register_buffer("p_fc_b", c(sd[p + "fc.bias"]))
        self.register_buffer("p_out_w", c(sd[p + "out.weight"].reshape(-1)))
        self.register_buffer("p_out_b", f(sd[p + "out.bias"]))
        d = "dual_head."
        self.register_buffer("d_pool_norm", c(sd[d + "pool_norm.weight"]))
        self.register_buffer("d_pool_w", c(sd[d + "pool_fc.weight"]))
        self.register_buffer("d_pool_b", c(sd[d + "pool_fc.bias"]))
        self.register_buffer("d_norm", c(sd[d + "norm.weight"]))
        self.register_buffer("d_fc_w", c(sd[d + "fc.weight"]))
        self.register_buffer("d_fc_b", c(sd[d + "fc.bias"]))
        self.register_buffer("d_out_norm", c(sd[d + "out_norm.weight"]))
        self.register_buffer("d_val_w", c(sd[d + "value_out.weight"]))
        self.register_buffer("d_val_b", c(sd[d + "value_out.bias"]))
        self.register_buffer("d_aux_w", c(sd[d + "aux_out.weight"].reshape(-1)))
        self.register_buffer("d_aux_b", f(sd[d + "aux_out.bias"]))
        self.p_gate_b_host = float(sd[p + "row_gate.bias"].reshape(-1)[0].item())
        # the all-HIP forward pass (include/az_nn.h: stem with the embedding, residual blocks, attention, heads) on
        # the GPU with bf16 activations and the reference's shapes; plain torch operations otherwise (CPU, fp32
        # builds for the parity tests, other widths).  One switch: tests flip `hip` to compare the two.
        self.hip = (self.device.type == "cuda" and dtype == torch.bfloat16 and self.embed_dim == 32
                    and self.h_dim == 64 and heads == 4 and glue() is not None)
        self.mfma_conv = self.mfma_attn = self.fused_stem = self.fused_heads = self.hip
        self._heads_w = None
        if self.fused_heads:
            hw = HeadsWeights()
            for n in HeadsWeights._PTRS:
                setattr(hw, n, getattr(self, n).data_ptr())
            hw.p_gate_b = self.p_gate_b_host
            hw.p_out_b = float(self.p_out_b.reshape(-1)[0].item())
            hw.d_aux_b = float(self.d_aux_b.reshape(-1)[0].item())
            hw.aux_scale = float(self.aux_target_offset)
            self._heads_w = hw

    def native_model(self):
        """az_nn_model* over this twin's weight buffers (include/az_nn.h): the whole forward pass as
        one C call, what az_mcts_dev_search runs inside its loop.  None when the all-HIP path is off."""
        if not self.supports_compact or not self.fused_stem or len(self.res) > MAX_BLOCKS:
            return None
        if self.__dict__.get("_model") is None:
            w = ModelWeights()
            for n in ("emb_own", "emb_opp", "pos", "stem_w", "stem_b", "pre_w", "qkvg_w", "qn_w", "kn_w", "o_w"):
                setattr(w, n, getattr(self, n).data_ptr())
            w.n_blocks = len(self.res)
            for i, names in enumerate(self.res):
                for field, name in zip(("block_w", "block_b", "block_gamma", "block_beta"), names):
                    getattr(w, field)[i] = getattr(self, name).data_ptr()
            w.heads = self._heads_w
            w.eps = 1e-5
            if self.folded_stem:
                w.stem_frag, w.stem_pmap = self.stem_frag.data_ptr(), self.stem_pmap.data_ptr()
            h = C.c_void_p()
            if glue().az_nn_model_create(C.byref(w), C.byref(h)) != 0:
                raise RuntimeError("az_nn_model_create refused the weights")
            self.__dict__["_model"] = h
        return self.__dict__["_model"]

    def __del__(self):
        # may run while the interpreter is being torn down: no nn.Module machinery, no exceptions
        try:
            h = self.__dict__.get("_model")
            if h is not None and _GLUE:
                self.__dict__["_model"] = None
                _GLUE.az_nn_model_destroy(h)
        except Exception:
            pass

    @classmethod
    def from_module(cls, net, dtype=torch.bfloat16, device=None):
        sd = net.state_dict()
        if device is None:
            device = next(iter(sd.values())).device
        return cls(sd, device, dtype)

    @staticmethod
    def recognises(net):
        need = ("piece_emb", "pos_emb", "hidden", "policy_head", "dual_head", "orbit_map")
        return all(hasattr(net, n) for n in need)

    # ------------------------------------------------------------------ pieces
    def _conv(self, tokens, w, b):
        """3x3 same convolution on (B, 42, Cin) tokens == NHWC image; returns (B, 42, Cout)."""
        bsz, _, cin = tokens.shape
        img = tokens.view(bsz, ROWS, COLS, cin).permute(0, 3, 1, 2)          # NCHW view, channels_last strides
        out = F.conv2d(img, w, b, padding=1)
        return out.permute(0, 2, 3, 1).reshape(bsz, CELLS, -1)

    def _group_norm1(self, tokens, gamma, beta):
        bsz = tokens.shape[0]
        y = F.layer_norm(tokens.reshape(bsz, -1), (tokens.shape[1] * tokens.shape[2],), eps=1e-5)
        return torch.addcmul(beta, y.view_as(tokens), gamma)

    def _rms(self, x, w):
        return F.rms_norm(x, (x.shape[-1],), w, 1e-5)

    # ------------------------------------------------------------------ GPU path
    @property
    def supports_compact(self):
        """predict_device(..., rows=, n_rows=, out=): evaluate only the listed rows"""
        return bool(self.hip)

    @torch.no_grad()
    def predict_device(self, x, action_mask=None, rows=None, n_rows=None, out=None):
        """(probs (B,7), wdl (B,3) relative [draw, win, loss], moves_left (B,)) fp32 on the device:
        what the reference's `predict` returns (Network.py:267-288) without the host copies.
        Compact form (HIP path only): `rows` int32 (B,) and `n_rows` int64 () on the device name
        the rows of x / action_mask to evaluate - the first n_rows entries of rows - and only those
        rows of `out` = (probs, wdl, ml) are written; the cost follows n_rows, not B."""
        if rows is not None:
            assert self.supports_compact and out is not None and n_rows is not None
            t, bsz, L, s = self._body_hip(x, rows, n_rows)
            probs, wdl, ml = out
            m = action_mask.contiguous()
            L.az_nn_heads(t.data_ptr(), C.byref(self._heads_w), m.data_ptr(), probs.data_ptr(), wdl.data_ptr(),
                          ml.data_ptr(), bsz, 1e-5, rows.data_ptr(), n_rows.data_ptr(), s)
            self._keep_mask = m
            return probs, wdl, ml
        if not (self.hip and x.is_cuda):
            lp, v, st = self(x, action_mask)
            return lp.exp(), v.exp(), st * float(self.aux_target_offset)
        if not self.fused_heads:
            lp, v, st = self._forward_hip(x, action_mask)
            return lp.exp(), v.exp(), st * float(self.aux_target_offset)
        t, bsz, L, s = self._body_hip(x)
        probs = torch.empty((bsz, COLS), dtype=torch.float32, device=self.device)
        wdl = torch.empty((bsz, 3), dtype=torch.float32, device=self.device)
        ml = torch.empty((bsz,), dtype=torch.float32, device=self.device)
        m = None
        if action_mask is not None:
            m = action_mask if action_mask.dtype in (torch.uint8, torch.bool) else action_mask.to(torch.bool)
            m = m.contiguous()
        L.az_nn_heads(t.data_ptr(), C.byref(self._heads_w), None if m is None else m.data_ptr(), probs.data_ptr(),
                      wdl.data_ptr(), ml.data_ptr(), bsz, 1e-5, None, None, s)
        self._keep_mask = m
        return probs, wdl, ml

    @torch.no_grad()
    def _forward_hip(self, x, action_mask):
        t, bsz, L, s = self._body_hip(x)
        return self._heads_hip(t, action_mask, bsz, L, s)

    @torch.no_grad()
    def _body_hip(self, x, rows=None, n_rows=None):
        """embedding, stem, residual blocks and attention on the MFMA kernels -> final tokens
        (of the compact batch when rows / n_rows are given)"""
        L = glue()
        gp = None if rows is None else rows.data_ptr()
        np_ = None if n_rows is None else n_rows.data_ptr()
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        bsz = x.shape[0]
        dev, bf = self.device, torch.bfloat16
        c_dim = self.h_dim
        x = x.contiguous().float()
        t = torch.empty((bsz, CELLS, c_dim), dtype=bf, device=dev)
        # embedding + stem convolution in one kernel: the tokens are built in LDS
        if self.folded_stem:
            L.az_nn_stem_folded(x.data_ptr(), self.stem_frag.data_ptr(), self.stem_pmap.data_ptr(), t.data_ptr(), bsz, gp, np_, s)
        else:
            L.az_nn_stem_embed(x.data_ptr(), self.emb_own.data_ptr(), self.emb_opp.data_ptr(), self.pos.data_ptr(),
                               self.stem_w.data_ptr(), self.stem_b.data_ptr(), t.data_ptr(), bsz, gp, np_, s)
        for w, b, g, beta in self.res:
            t2 = torch.empty_like(t)
            L.az_nn_conv_block(t.data_ptr(), c_dim, getattr(self, w).data_ptr(), getattr(self, b).data_ptr(),
                               getattr(self, g).data_ptr(), getattr(self, beta).data_ptr(), 1, t2.data_ptr(),
                               bsz, 1e-5, np_, s)
            t = t2
        t2 = torch.empty_like(t)
        L.az_nn_attn_block(t.data_ptr(), self.pre_w.data_ptr(), self.qkvg_w.data_ptr(), self.qn_w.data_ptr(),
                           self.kn_w.data_ptr(), self.o_w.data_ptr(), t2.data_ptr(), bsz, 1e-5, np_, s)
        return t2, bsz, L, s

    @torch.no_grad()
    def _heads_hip(self, t, action_mask, bsz, L, s):
        """the heads as az_nn_heads_prep + torch operations: `forward`'s log-probabilities, and what the fused
        heads kernel (az_nn_heads, used by predict_device) is compared with in the tests"""
        dev, bf, c_dim = self.device, torch.bfloat16, self.h_dim
        col = torch.empty((bsz, COLS, c_dim), dtype=bf, device=dev)
        mean = torch.empty((bsz, c_dim), dtype=bf, device=dev)
        L.az_nn_heads_prep(t.data_ptr(), self.p_norm.data_ptr(), self.p_gate_w.data_ptr(), self.p_gate_b_host,
                           col.data_ptr(), mean.data_ptr(), bsz, 1e-5, s)
        col = F.silu(F.linear(col, self.p_fc_w, self.p_fc_b))
        logits = (col.float() * self.p_out_w.float()).sum(-1) + self.p_out_b
        if action_mask is not None:
            logits = logits.masked_fill(~action_mask.to(torch.bool), -1e9)
        log_prob = F.log_softmax(logits, dim=-1)
        g = mean + F.silu(F.linear(self._rms(mean, self.d_pool_norm), self.d_pool_w, self.d_pool_b))
        hh = self._rms(F.silu(F.linear(self._rms(g, self.d_norm), self.d_fc_w, self.d_fc_b)), self.d_out_norm)
        value = F.log_softmax(F.linear(hh, self.d_val_w, self.d_val_b).float(), dim=-1)
        steps = torch.sigmoid((hh.float() * self.d_aux_w.float()).sum(-1) + self.d_aux_b)
        return log_prob, value, steps

    @torch.no_grad()
    def forward(self, x, action_mask=None):
        """x: (B, 3, 6, 7) relative planes (any float dtype).  Returns fp32
        (log_prob (B,7), value_log_prob (B,3), steps_norm (B,)) like the reference forward."""
        if self.hip and x.is_cuda:
            return self._forward_hip(x, action_mask)
        bsz = x.shape[0]
        dt = self.dtype
        own = x[:, 0].reshape(bsz, CELLS, 1).to(dt)
        opp = x[:, 1].reshape(bsz, CELLS, 1).to(dt)
        t = torch.addcmul(torch.addcmul(self.pos, own, self.emb_own), opp, self.emb_opp)   # (B, 42, E)

        t = F.silu(self._conv(t, self.stem_w, self.stem_b))
        for w, b, g, beta in self.res:
            y = self._group_norm1(t, getattr(self, g), getattr(self, beta))
            t = t + F.silu(self._conv(y, getattr(self, w), getattr(self, b)))

        # gated attention over the 42 cells
        h = self._rms(t, self.pre_w)
        qkvg = F.linear(h, self.qkvg_w)                                       # (B, 42, 3C + heads)
        c_dim, nh = self.h_dim, self.heads
        hd = c_dim // nh
        q, k, v = qkvg[..., :3 * c_dim].view(bsz, CELLS, 3, nh, hd).unbind(2)
        gate = qkvg[..., 3 * c_dim:]                                          # (B, 42, heads)
        q = self._rms(q, self.qn_w).transpose(1, 2)
        k = self._rms(k, self.kn_w).transpose(1, 2)
        a = F.scaled_dot_product_attention(q, k, v.transpose(1, 2))           # (B, heads, 42, hd)
        a = a * torch.sigmoid(gate).transpose(1, 2).unsqueeze(-1)
        t = F.linear(a.transpose(1, 2).reshape(bsz, CELLS, c_dim), self.o_w) + t

        # column policy head
        pn = self._rms(t, self.p_norm).view(bsz, ROWS, COLS, c_dim)
        scores = (pn.float() * self.p_gate_w.float()).sum(-1) + self.p_gate_b     # (B, rows, cols)
        wts = torch.softmax(scores, dim=1).to(dt)                                  # over rows
        col = (wts.unsqueeze(-1) * pn).sum(dim=1)                                  # (B, cols, C)
        col = F.silu(F.linear(col, self.p_fc_w, self.p_fc_b))
        logits = (col.float() * self.p_out_w.float()).sum(-1) + self.p_out_b       # (B, cols)
        if action_mask is not None:
            logits = logits.masked_fill(~action_mask.to(torch.bool), -1e9)
        log_prob = F.log_softmax(logits, dim=-1)

        # value / moves-left head
        g = t.float().mean(dim=1).to(dt)
        g = g + F.silu(F.linear(self._rms(g, self.d_pool_norm), self.d_pool_w, self.d_pool_b))
        hh = self._rms(F.silu(F.linear(self._rms(g, self.d_norm), self.d_fc_w, self.d_fc_b)), self.d_out_norm)
        value = F.log_softmax(F.linear(hh, self.d_val_w, self.d_val_b).float(), dim=-1)
        steps = torch.sigmoid((hh.float() * self.d_aux_w.float()).sum(-1) + self.d_aux_b)
        return log_prob, value, steps

    @torch.no_grad()
    def predict(self, state, action_mask=None):
        import numpy as np
        t = torch.as_tensor(np.asarray(state), dtype=torch.float32, device=self.device)
        m = None if action_mask is None else torch.as_tensor(np.asarray(action_mask), device=self.device)
        probs, wdl, ml = self.predict_device(t, m)
        return probs.cpu().numpy(), wdl.cpu().numpy(), ml.view(-1, 1).cpu().numpy()
