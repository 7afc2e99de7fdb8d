"""Inference twin of the Othello network (az_net.OthelloNet / the reference's Othello CNN) for the
device loop: its eight 256-channel 3x3 convolutions - ~97 % of the ~1 GFLOP a leaf costs - run on
the hand-written MFMA kernel of csrc/nn_othello.hip (include/az_nn.h: az_nn_othello_conv), with
the BatchNorms as affines inside it; the embedding and the thin ends of the two heads (a 1x1
convolution, an 8-channel bottleneck, three small linears) stay torch operations.

Activations are NHWC bf16 between layers; weights are packed once into the kernel's fragment
order.  `predict_device(x, mask)` returns what the reference's `predict` returns
(Othello/Network.py:229-261) without leaving the device: probabilities over the 65 actions,
relative WDL, and the score utility atan(disc difference / score_scale) * 2/pi.
"""
import ctypes as C
import math
import os

import torch
import torch.nn.functional as F

from src.fast_net import glue


class _HeadsW(C.Structure):          # az_nn_othello_heads_weights (include/az_nn.h)
    _PTRS = ("board_w", "pass_norm_w", "pass_fc_w", "v_conv_w", "v_bn_s", "v_bn_b", "v_fc_w", "v_fc_b", "a_fc_wt", "a_fc_b",
             "a_norm_w", "a_out_w")
    _fields_ = ([(n, C.c_void_p) for n in _PTRS] + [(n, C.c_float) for n in ("board_b", "pass_fc_b", "a_out_b", "aux_to_score", "eps")] +
                [("a_fc_w16", C.c_void_p)])


class _ConvLayer(C.Structure):       # az_nn_othello_conv_layer
    _fields_ = [(n, C.c_void_p) for n in ("w_packed", "pre_scale", "pre_shift", "post_scale", "post_shift")] + \
               [(n, C.c_int32) for n in ("residual", "c_in", "h_in", "pad")]


class _OthelloW(C.Structure):        # az_nn_othello_weights
    _fields_ = [("embed_table", C.c_void_p), ("n_body", C.c_int32), ("n_convs", C.c_int32), ("conv", _ConvLayer * 16),
                ("dual_w16", C.c_void_p), ("dual_scale16", C.c_void_p), ("dual_shift16", C.c_void_p), ("heads", _HeadsW)]


def _bn_affine(bn):
    scale = (bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps))
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    return scale.contiguous(), shift.contiguous()


def pack_conv_weight(w):
    """(256, C_in, 3, 3) -> bf16 in the kernel's A-fragment order [tap][k chunk of 32][channel tile]
    [lane = 16 * k group + channel][8 consecutive input channels]."""
    co, ci = w.shape[0], w.shape[1]
    assert co % 16 == 0 and ci % 32 == 0 and tuple(w.shape[2:]) == (3, 3)
    t = w.detach().to(torch.bfloat16).permute(2, 3, 1, 0).reshape(9, ci // 32, 4, 8, co // 16, 16)   # tap, kc, g, j, tile, tl
    return t.permute(0, 1, 4, 2, 5, 3).contiguous()


class FastOthelloNet(torch.nn.Module):
    aux_target_offset = 64
    n_actions = 65
    # compact batches (only the rows named by a device-side list: the live leaves, or the leaves that missed
    # the transposition table) are supported by gathering them on the host's say-so: the list's length is
    # read back once per call - 20 us against an ~18 ms evaluation - so not inside a captured graph
    supports_compact = True
    compact_needs_host = True

    @staticmethod
    def recognises(net):
        need = ("piece_emb", "pos_emb", "legal_emb", "stem", "policy_head", "dual_head", "orbit_map")
        if not all(hasattr(net, n) for n in need):
            return False
        w = net.stem[0].weight
        return tuple(w.shape) == (256, 32, 3, 3) and glue() is not None and hasattr(glue(), "az_nn_othello_conv")

    def __init__(self, net):
        super().__init__()
        self.net = net
        self.device = net.stem[0].weight.device
        self.score_scale = float(getattr(net, "score_scale", 8.0))
        L = glue()
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
        L.az_nn_othello_conv.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp]
        self._L = L
        dev = self.device
        ones = torch.ones(256, device=dev)
        zeros = torch.zeros(256, device=dev)
        mods = list(net.stem)
        n_blocks = sum(1 for m in mods if hasattr(m, "conv1"))
        self.layers = []        # (packed weight, pre, post, residual?, c_in, h_in, pad)

        def add(conv, pre, post, res, c_in, h_in, pad):
            wp = pack_conv_weight(conv.weight)
            pre = tuple(t.to(dev) for t in pre) if pre is not None else None
            post = tuple(t.to(dev) for t in post) if post is not None else (ones, zeros)
            self.layers.append((wp, pre, post, res, c_in, h_in, pad))

        add(mods[0], None, _bn_affine(mods[1]), False, 32, 8, 2)
        for blk in mods[3:3 + n_blocks]:
            add(blk.conv1, _bn_affine(blk.norm1), None, False, 256, 10, 1)
            add(blk.conv2, _bn_affine(blk.norm2), None, True, 256, 10, 1)
        add(mods[3 + n_blocks], None, _bn_affine(mods[4 + n_blocks]), False, 256, 10, 1)
        self.n_body = len(self.layers)
        # the embedding as one table lookup (Othello/Network.py:201-211): a cell shows its orbit's
        # position vector plus exactly one of {own stone, opponent stone, empty + legal, empty + illegal}
        with torch.no_grad():
            pos = net.pos_emb(net.orbit_map).float()                                               # (64, 32)
            kinds = torch.stack([net.piece_emb.weight[0], net.piece_emb.weight[1], net.legal_emb.weight[1],
                                 net.legal_emb.weight[0]]).float()                                 # (4, 32)
            self.embed_table = (pos[:, None, :] + kinds[None, :, :]).to(torch.bfloat16).reshape(256, 32).contiguous()
        self.cell4 = (torch.arange(64, device=dev) * 4).view(1, 64)
        ps = net.policy_head.stem
        add(ps[0], None, _bn_affine(ps[1]), False, 256, 10, 0)
        add(ps[4], None, _bn_affine(ps[5]), False, 256, 8, 1)
        ph = net.policy_head
        self.board_w = ph.board_out.weight.detach().to(torch.bfloat16).reshape(256).contiguous()   # bf16, as under autocast
        self.board_b = float(ph.board_out.bias.detach().float().item())
        # the dual head's 8-channel bottleneck (3x3, no padding, 10x10 -> 8x8): its own narrow kernel
        dh = net.dual_head
        w16 = torch.zeros((16, 256, 3, 3), device=dev)
        w16[:8] = dh.stem[0].weight.detach().float()
        s8, b8 = _bn_affine(dh.stem[1])
        self.dual_w = pack_conv_weight(w16)
        self.dual_s = torch.ones(16, device=dev)
        self.dual_b = torch.zeros(16, device=dev)
        self.dual_s[:8], self.dual_b[:8] = s8.to(dev), b8.to(dev)
        L.az_nn_othello_conv_narrow.argtypes = [vp, vp, vp, vp, vp, i64, vp, vp]
        self.v_conv_w = dh.value_out[0].weight.detach().float().reshape(8, 72).t().contiguous()    # (72, 8)
        self.v_bn = tuple(t.to(dev).view(1, 8, 1) for t in _bn_affine(dh.value_out[1]))

    def native_model(self):
        """az_nn_model* (kind OTHELLO_CNN, include/az_nn.h) over this twin's buffers: embedding, the
        convolutions, the bottleneck and both heads as one C call - what az_mcts_dev_search runs inside its
        loop for Othello; it reads the leaves' bitboards and evaluates only the rows a device-side list names."""
        if self.__dict__.get("_model") is not None:
            return self.__dict__["_model"]
        dev, f32 = self.device, torch.float32
        net = self.net
        keep = {}

        def t(x):
            y = x.detach().to(dev, f32).contiguous()
            keep[len(keep)] = y
            return y.data_ptr()
        w = _OthelloW()
        w.embed_table = self.embed_table.data_ptr()
        w.n_body, w.n_convs = self.n_body, self.n_body + 2
        assert w.n_convs <= 16 and w.n_convs == len(self.layers)
        for i, (wp, pre, post, res, c_in, h_in, pad) in enumerate(self.layers):
            L = w.conv[i]
            L.w_packed = wp.data_ptr()
            L.pre_scale, L.pre_shift = (pre[0].data_ptr(), pre[1].data_ptr()) if pre is not None else (None, None)
            L.post_scale, L.post_shift = post[0].data_ptr(), post[1].data_ptr()
            L.residual, L.c_in, L.h_in, L.pad = int(bool(res)), c_in, h_in, pad
        w.dual_w16, w.dual_scale16, w.dual_shift16 = self.dual_w.data_ptr(), self.dual_s.data_ptr(), self.dual_b.data_ptr()
        ph, dh, h = net.policy_head, net.dual_head, w.heads
        h.board_w = self.board_w.data_ptr()
        h.pass_norm_w, h.pass_fc_w = t(ph.pass_norm.weight), t(ph.pass_fc.weight.reshape(-1))
        h.v_conv_w = self.v_conv_w.data_ptr()
        h.v_bn_s, h.v_bn_b = t(self.v_bn[0].reshape(-1)), t(self.v_bn[1].reshape(-1))
        h.v_fc_w, h.v_fc_b = t(dh.value_out[5].weight), t(dh.value_out[5].bias)
        h.a_fc_wt, h.a_fc_b = t(dh.aux_out[1].weight.t()), t(dh.aux_out[1].bias)
        if os.environ.get("AZ_OTH_HEADS_MFMA", "1") != "0":
            # the same weight for the matrix cores: bf16 (as under autocast), input index in the bottleneck's NHWC order
            # (the Linear takes the (8 channels, 8, 8) map flattened channel-major: in = 64 c + cell -> 8 cell + c)
            w16 = dh.aux_out[1].weight.detach().to(dev, f32).reshape(512, 8, 64).transpose(1, 2).reshape(512, 512)
            keep["a_fc_w16"] = w16.to(torch.bfloat16).contiguous()
            h.a_fc_w16 = keep["a_fc_w16"].data_ptr()
        h.a_norm_w, h.a_out_w = t(dh.aux_out[2].weight), t(dh.aux_out[5].weight.reshape(-1))
        h.board_b, h.pass_fc_b, h.a_out_b = self.board_b, float(ph.pass_fc.bias.item()), float(dh.aux_out[5].bias.item())
        h.aux_to_score = float(self.aux_target_offset) / self.score_scale
        h.eps = 1e-5
        L = self._L
        L.az_nn_model_create_othello.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        handle = C.c_void_p()
        if L.az_nn_model_create_othello(C.byref(w), C.byref(handle)) != 0:
            raise RuntimeError("az_nn_model_create_othello refused the weights")
        self.__dict__["_model_keep"] = keep
        self.__dict__["_model"] = handle
        return handle

    def __del__(self):
        try:
            h = self.__dict__.get("_model")
            if h is not None:
                self.__dict__["_model"] = None
                self._L.az_nn_model_destroy(h)
        except Exception:
            pass

    def _conv(self, x, layer, residual, stream):
        wp, pre, post, res, c_in, h_in, pad = layer
        bsz = x.shape[0]
        ho = h_in + 2 * pad - 2
        y = torch.empty((bsz, ho, ho, 256), dtype=torch.bfloat16, device=self.device)
        rc = self._L.az_nn_othello_conv(x.data_ptr(), wp.data_ptr(), None if pre is None else pre[0].data_ptr(),
                                        None if pre is None else pre[1].data_ptr(), post[0].data_ptr(), post[1].data_ptr(),
                                        residual.data_ptr() if res else None, y.data_ptr(), bsz, c_in, h_in, pad, 1, None, stream)
        if rc != 0:
            raise RuntimeError("az_nn_othello_conv refused its arguments (%d)" % rc)
        return y

    @torch.no_grad()
    def body(self, x, action_mask):
        """(B, 3, 8, 8) planes + (B, 65) mask -> hidden (B, 10, 10, 256) NHWC bf16"""
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        mask = action_mask.view(torch.bool) if action_mask.dtype == torch.uint8 else action_mask.to(torch.bool)
        bsz = x.shape[0]
        own = x[:, 0].reshape(bsz, 64) > 0.5
        opp = x[:, 1].reshape(bsz, 64) > 0.5
        kind = torch.where(own, 0, torch.where(opp, 1, torch.where(mask[:, :64], 2, 3)))
        t = self.embed_table[(kind + self.cell4).reshape(-1)].view(bsz, 8, 8, 32)                  # NHWC bf16
        h = self._conv(t, self.layers[0], None, s)
        i = 1
        while i < self.n_body - 1:
            y1 = self._conv(h, self.layers[i], None, s)
            h = self._conv(y1, self.layers[i + 1], h, s)
            i += 2
        return self._conv(h, self.layers[self.n_body - 1], None, s), s

    @torch.no_grad()
    def forward(self, x, action_mask=None):
        """(log_prob (B, 65), log wdl (B, 3), aux (B,)) as the module's forward returns them"""
        net = self.net
        hidden, s = self.body(x, action_mask)
        p = self._conv(hidden, self.layers[self.n_body], None, s)
        p = self._conv(p, self.layers[self.n_body + 1], None, s)                                   # (B, 8, 8, 256)
        pf = p.reshape(p.shape[0], 64, 256)
        squares = (pf @ self.board_w).float() + self.board_b                                       # the 1x1 convolution
        ph = net.policy_head
        skip = ph.pass_fc(ph.pass_norm(pf.mean(dim=1, dtype=torch.float32))).float()
        log_prob = F.log_softmax(torch.cat([squares, skip], dim=1), dim=-1)
        # dual head (Othello/Network.py:78-104) on the 8-channel bottleneck: a (B, 8, 8, 8) tensor
        dh = net.dual_head
        h8 = torch.empty((hidden.shape[0], 8, 8, 8), dtype=torch.bfloat16, device=self.device)
        if self._L.az_nn_othello_conv_narrow(hidden.data_ptr(), self.dual_w.data_ptr(), self.dual_s.data_ptr(),
                                             self.dual_b.data_ptr(), h8.data_ptr(), hidden.shape[0], None, s) != 0:
            raise RuntimeError("az_nn_othello_conv_narrow refused its arguments")
        h8 = h8.permute(0, 3, 1, 2).float().contiguous()                                            # (B, 8 channels, 8, 8)
        # 3x3 stride-2 convolution 8 -> 8 on the 8x8 map as strided window views + one small GEMM (the
        # library convolution and F.unfold both work sample by sample here)
        win = h8.unfold(2, 3, 2).unfold(3, 3, 2)                                      # (B, c, oy, ox, ky, kx) view
        v = (win.permute(0, 2, 3, 1, 4, 5).reshape(h8.shape[0], 9, 72) @ self.v_conv_w).transpose(1, 2)   # (B, 8, 9)
        v = F.silu(v * self.v_bn[0] + self.v_bn[1])
        value = F.log_softmax(dh.value_out[5](v.flatten(1)), dim=-1)
        a = dh.aux_out[1](h8.flatten(1))
        aux = torch.tanh(dh.aux_out[5](F.silu(dh.aux_out[2](a)))).squeeze(-1)
        return log_prob, value.float(), aux.float()

    @torch.no_grad()
    def predict_device(self, x, action_mask=None, rows=None, n_rows=None, out=None):
        if rows is not None:
            n = int(n_rows.item())                                   # host round trip: see compact_needs_host
            if n == 0:
                return out
            idx = rows[:n].long()
            p, w, u = self.predict_device(x.index_select(0, idx), action_mask.index_select(0, idx))
            out[0].index_copy_(0, idx, p); out[1].index_copy_(0, idx, w); out[2].index_copy_(0, idx, u)
            return out
        log_prob, value, aux = self.forward(x, action_mask)
        utility = torch.atan(aux * (float(self.aux_target_offset) / self.score_scale)) * (2.0 / math.pi)
        res = (log_prob.exp().contiguous(), value.exp().contiguous(), utility.reshape(-1).contiguous())
        if out is not None:
            for dst, src in zip(out, res):
                dst.copy_(src)
            return out
        return res
