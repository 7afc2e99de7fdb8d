"""Inference-side Connect4 policy / value / moves-left network (and, below, the Othello network).

Same architecture, parameter names and numerics as the reference's
src/environments/Connect4/Network.py::CNN (embedding 14-93 and 226-246, body 163-175, heads
96-141), so `load_state_dict(strict=True)` accepts the reference's checkpoints and the
reference's own CNN can be swapped for this one.  Only the forward / predict side exists here:
optimiser, schedulers and losses belong to the learner, which is out of scope.

The search never depends on this file: `BatchedMCTS.batch_playout` takes any evaluator.  It
exists so that bench.py and the tests can evaluate "the reference's network" on a box that
does not have the reference checkout.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROWS, COLS, CELLS = 6, 7, 42


def _orbit_map():
    """Cell -> left/right-mirror orbit id: 6 rows x 4 distinct columns (Network.py:8-21)."""
    col = torch.tensor([0, 1, 2, 3, 2, 1, 0])
    return (torch.arange(ROWS)[:, None] * 4 + col[None, :]).reshape(-1)


class _Res(nn.Module):
    """Pre-norm residual conv block (Network.py:27-48): x + silu(conv(groupnorm_1(x)))."""

    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)
        self.norm = nn.GroupNorm(1, ch)

    def forward(self, x):
        return x + F.silu(self.conv(self.norm(x)))


class _GatedAttention(nn.Module):
    """RMS-prenorm multi-head attention with per-head sigmoid output gates and QK-norm
    (Network.py:51-93)."""

    def __init__(self, dim, heads):
        super().__init__()
        self.heads, self.hd = heads, dim // heads
        self.prenorm = nn.RMSNorm(dim, eps=1e-5)
        self.qkv_proj = nn.Linear(dim, 3 * dim, bias=False)
        self.gate_proj = nn.Linear(dim, heads, bias=False)
        self.o_proj = nn.Linear(dim, dim, bias=False)
        self.q_norm = nn.RMSNorm(self.hd, eps=1e-5)
        self.k_norm = nn.RMSNorm(self.hd, eps=1e-5)

    def forward(self, x):                                   # x: (B, T, dim)
        b, t, dim = x.shape
        h = self.prenorm(x)
        q, k, v = self.qkv_proj(h).view(b, t, 3, self.heads, self.hd).unbind(2)
        gate = self.gate_proj(h)                            # (B, T, heads)
        q = self.q_norm(q).transpose(1, 2)
        k = self.k_norm(k).transpose(1, 2)
        a = F.scaled_dot_product_attention(q, k, v.transpose(1, 2))
        a = a * gate.transpose(1, 2).unsqueeze(-1).sigmoid()
        return self.o_proj(a.transpose(1, 2).reshape(b, t, dim)) + x


class _AttnBlock(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.attn = _GatedAttention(dim, heads)

    def forward(self, x):                                   # (B, C, H, W) -> (B, H*W, C)
        return self.attn(x.flatten(2).transpose(1, 2))


class _ColumnPolicy(nn.Module):
    """Softmax-over-rows pooling of each column, then a 2-layer head per column
    (Network.py:96-118)."""

    def __init__(self, dim):
        super().__init__()
        self.norm = nn.RMSNorm(dim, eps=1e-5)
        self.row_gate = nn.Linear(dim, 1)
        self.fc = nn.Linear(dim, dim)
        self.out = nn.Linear(dim, 1)

    def forward(self, tokens, action_mask=None):
        b, _, dim = tokens.shape
        x = self.norm(tokens).reshape(b, ROWS, COLS, dim).transpose(1, 2)      # (B, cols, rows, dim)
        w = torch.softmax(self.row_gate(x).squeeze(-1), dim=-1)
        col = (w.unsqueeze(-1) * x).sum(dim=2)
        logits = self.out(F.silu(self.fc(col))).squeeze(-1)
        if action_mask is not None:
            logits = logits.masked_fill(~action_mask, -1e9)
        return F.log_softmax(logits, dim=-1)


class _ValueAux(nn.Module):
    """Mean-pooled tokens -> WDL log-probabilities and a sigmoid moves-left fraction
    (Network.py:121-141)."""

    def __init__(self, dim, n_classes=3):
        super().__init__()
        self.pool_norm = nn.RMSNorm(dim, eps=1e-5)
        self.pool_fc = nn.Linear(dim, dim)
        self.norm = nn.RMSNorm(dim, eps=1e-5)
        self.fc = nn.Linear(dim, dim)
        self.out_norm = nn.RMSNorm(dim, eps=1e-5)
        self.value_out = nn.Linear(dim, n_classes)
        self.aux_out = nn.Linear(dim, 1)

    def forward(self, tokens):
        x = tokens.mean(dim=1)
        x = x + F.silu(self.pool_fc(self.pool_norm(x)))
        h = self.out_norm(F.silu(self.fc(self.norm(x))))
        return F.log_softmax(self.value_out(h), dim=-1), torch.sigmoid(self.aux_out(h).squeeze(-1))


class Connect4Net(nn.Module):
    aux_target_offset = 42          # moves-left head predicts a fraction of 42 plies (Network.py:145)
    n_actions = COLS

    def __init__(self, embed_dim=32, h_dim=64, num_res_blocks=3, heads=4, device='cpu'):
        super().__init__()
        self.embed_dim = embed_dim
        self.in_dim = 3
        self.device = device
        self.piece_emb = nn.Embedding(2, embed_dim)         # own / opponent stone
        self.pos_emb = nn.Embedding(24, embed_dim)          # mirror-orbit of the cell
        self.register_buffer('orbit_map', _orbit_map())
        self.hidden = nn.Sequential(
            nn.Conv2d(embed_dim, h_dim, 3, padding=1),
            nn.SiLU(),
            *[_Res(h_dim) for _ in range(num_res_blocks)],
            _AttnBlock(h_dim, heads),
        )
        self.policy_head = _ColumnPolicy(h_dim)
        self.dual_head = _ValueAux(h_dim)
        self.reset_parameters()
        self.to(device)

    def reset_parameters(self):
        """The reference's initialisation (Network.py:181-185,207-214): orthogonal embeddings,
        fan-in Kaiming-normal conv/linear weights with zero bias, and ZERO output weights of
        the three heads - a fresh network predicts a uniform policy, WDL = 1/3 each and
        21 moves left."""
        for m in self.modules():
            if isinstance(m, nn.Embedding):
                nn.init.orthogonal_(m.weight)
            elif isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight, mode='fan_in', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        for lin in (self.policy_head.out, self.dual_head.value_out, self.dual_head.aux_out):
            nn.init.zeros_(lin.weight)

    def embed(self, state):
        """(B, 3, 6, 7) relative planes -> (B, embed_dim, 6, 7): piece embedding where a stone
        is, plus the positional embedding everywhere (Network.py:226-239)."""
        b = state.size(0)
        own = state[:, 0].reshape(b, CELLS, 1)
        opp = state[:, 1].reshape(b, CELLS, 1)
        x = own * self.piece_emb.weight[0] + opp * self.piece_emb.weight[1] + self.pos_emb(self.orbit_map)
        return x.transpose(1, 2).reshape(b, self.embed_dim, ROWS, COLS)

    def forward(self, x, action_mask=None):
        if action_mask is not None:
            if isinstance(action_mask, np.ndarray):
                action_mask = torch.from_numpy(action_mask)
            if action_mask.ndim == 1:
                action_mask = action_mask.unsqueeze(0)
            action_mask = action_mask.to(device=x.device, dtype=torch.bool)
        tokens = self.hidden(self.embed(x))
        log_prob = self.policy_head(tokens, action_mask)
        value, steps = self.dual_head(tokens)
        return log_prob, value, steps

    def name(self):
        return 'CNN'

    def _dev_type(self):
        return torch.device(self.device).type

    @torch.no_grad()
    def predict(self, state, action_mask=None):
        """numpy in / numpy out contract of the reference (Network.py:267-288): probabilities,
        RELATIVE wdl [draw, win, loss] and expected remaining plies (n, 1); bf16 autocast off
        the CPU."""
        t = torch.from_numpy(state) if isinstance(state, np.ndarray) else state
        t = t.to(self.device, dtype=torch.float32)
        dev = self._dev_type()
        with torch.autocast(dev, dtype=torch.bfloat16, enabled=dev != 'cpu'):
            log_prob, value_lp, steps = self(t, action_mask=action_mask)
        probs = log_prob.float().exp()
        wdl = value_lp.exp().float()
        ml = (steps * float(self.aux_target_offset)).float().view(-1, 1)
        return probs.cpu().numpy(), wdl.cpu().numpy(), ml.cpu().numpy()

    @torch.no_grad()
    def policy(self, state, action_mask=None):
        return self.predict(state, action_mask)[0]

    @torch.no_grad()
    def value(self, state, action_mask=None):
        return self.predict(state, action_mask)[1]


# ---------------------------------------------------------------------------------------------
# Othello (src/environments/Othello/Network.py::CNN): inference side, state-dict compatible

def _othello_orbits():
    """Cell -> orbit of the square's dihedral symmetry group: 10 classes on the 8x8 board
    (Othello/Network.py:10-19).  Fold the board onto one octant; the octant's triangle
    a <= b <= 3 is numbered row by row."""
    fold = torch.tensor([0, 1, 2, 3, 3, 2, 1, 0])
    a = torch.minimum(fold[:, None], fold[None, :])
    b = torch.maximum(fold[:, None], fold[None, :])
    start = torch.tensor([0, 4, 7, 9])                     # first id of the triangle's row a
    return (start[a] + (b - a)).reshape(-1)


class _BnRes(nn.Module):
    """x -> silu(conv2(bn2(silu(conv1(bn1(x))))) + x)   (Othello/Network.py:22-37, dropout is
    identity at inference)."""

    def __init__(self, ch):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(ch)
        self.conv1 = nn.Conv2d(ch, ch, 3, padding=1, bias=False)
        self.norm2 = nn.BatchNorm2d(ch)
        self.conv2 = nn.Conv2d(ch, ch, 3, padding=1, bias=False)

    def forward(self, x):
        y = F.silu(self.conv1(self.norm1(x)))
        return F.silu(self.conv2(self.norm2(y)) + x)


def _conv_bn_silu(cin, cout, **kw):
    """[conv, bn, silu, (dropout)]: four slots, so that Sequential indices equal the reference's"""
    return [nn.Conv2d(cin, cout, 3, bias=False, **kw), nn.BatchNorm2d(cout), nn.SiLU(), nn.Identity()]


class _OthelloPolicy(nn.Module):
    """64 square logits from a 1x1 convolution of two more conv layers (the first unpadded: the
    body works on a 10x10 map) and one pass logit from the pooled features; log-softmax over all
    65 WITHOUT masking (Othello/Network.py:40-66 - the masked_fill there is commented out)."""

    def __init__(self, ch):
        super().__init__()
        self.stem = nn.Sequential(*_conv_bn_silu(ch, ch), *_conv_bn_silu(ch, ch, padding=1))
        self.board_out = nn.Conv2d(ch, 1, 1)
        self.pass_norm = nn.RMSNorm(ch, eps=1e-5)
        self.pass_fc = nn.Linear(ch, 1)

    def forward(self, x):
        x = self.stem(x)
        squares = self.board_out(x).flatten(1)
        skip = self.pass_fc(self.pass_norm(x.mean(dim=(2, 3))))
        return F.log_softmax(torch.cat([squares, skip], dim=1), dim=-1)


class _OthelloValueAux(nn.Module):
    """8-channel bottleneck -> WDL (strided conv + linear) and the disc-difference scalar
    (tanh of a 512-wide MLP), Othello/Network.py:78-104."""

    def __init__(self, ch):
        super().__init__()
        self.stem = nn.Sequential(*_conv_bn_silu(ch, 8)[:3])
        self.value_out = nn.Sequential(nn.Conv2d(8, 8, 3, stride=2, bias=False), nn.BatchNorm2d(8), nn.SiLU(),
                                       nn.Identity(), nn.Flatten(), nn.Linear(72, 3))
        self.aux_out = nn.Sequential(nn.Flatten(), nn.Linear(512, 512), nn.RMSNorm(512, eps=1e-5), nn.SiLU(),
                                     nn.Identity(), nn.Linear(512, 1))

    def forward(self, x):
        h = self.stem(x)
        return F.log_softmax(self.value_out(h), dim=-1), torch.tanh(self.aux_out(h).squeeze(-1))


class OthelloNet(nn.Module):
    """The reference's Othello CNN at inference (Othello/Network.py:107-261): embedding of stones,
    square orbit and - on empty squares - legality, a stem convolution with padding 2 (the body
    lives on a 10x10 map), `num_res_blocks` BatchNorm residual blocks, one more convolution, and
    the two heads.  Parameter and buffer names equal the reference's, so its checkpoints load with
    strict=True.  The evaluator contract is the same as Connect4Net's; `forward` REQUIRES the
    legal-move mask (it is an input feature here)."""
    aux_target_offset = 64          # the aux head predicts disc difference / 64
    score_scale = 8.0
    n_actions = 65

    def __init__(self, embed_dim=32, h_dim=256, num_res_blocks=3, device='cpu'):
        super().__init__()
        self.embed_dim = embed_dim
        self.in_dim = 3
        self.device = device
        self.piece_emb = nn.Embedding(2, embed_dim)
        self.pos_emb = nn.Embedding(10, embed_dim)
        self.legal_emb = nn.Embedding(2, embed_dim)
        self.register_buffer('orbit_map', _othello_orbits())
        self.stem = nn.Sequential(*_conv_bn_silu(embed_dim, h_dim, padding=2)[:3],
                                  *[_BnRes(h_dim) for _ in range(num_res_blocks)],
                                  *_conv_bn_silu(h_dim, h_dim, padding=1))
        self.policy_head = _OthelloPolicy(h_dim)
        self.dual_head = _OthelloValueAux(h_dim)
        self.reset_parameters()
        self.to(device)
        self.eval()

    def reset_parameters(self):
        """Othello/Network.py:143-145,176-183 and the heads' reset_output_parameters: zero output
        layers, so a fresh network is uniform over the 65 actions, WDL 1/3 each, disc difference 0."""
        for m in self.modules():
            if isinstance(m, nn.Embedding):
                nn.init.orthogonal_(m.weight)
            elif isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight, mode='fan_in', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        for m in (self.policy_head.board_out, self.policy_head.pass_fc, self.dual_head.value_out[-1],
                  self.dual_head.aux_out[-1]):
            nn.init.zeros_(m.weight)
            nn.init.zeros_(m.bias)

    @staticmethod
    def _mask(action_mask, device):
        if action_mask is None:
            raise ValueError('the Othello network needs the legal-move mask (it is an input feature)')
        if isinstance(action_mask, np.ndarray):
            action_mask = torch.from_numpy(action_mask)
        if action_mask.ndim == 1:
            action_mask = action_mask.unsqueeze(0)
        return action_mask.to(device=device, dtype=torch.bool)

    def embed(self, state, action_mask):
        """(B, 3, 8, 8) relative planes + (B, 65) mask -> (B, embed_dim, 8, 8), Othello/Network.py:201-211"""
        b = state.size(0)
        own = (state[:, 0].reshape(b, 64) > 0.5).unsqueeze(-1)
        opp = (state[:, 1].reshape(b, 64) > 0.5).unsqueeze(-1)
        x = self.pos_emb(self.orbit_map).unsqueeze(0) + own * self.piece_emb.weight[0] + opp * self.piece_emb.weight[1]
        x = x + (~(own | opp)) * self.legal_emb(action_mask[:, :64].long())
        return x.transpose(1, 2).reshape(b, self.embed_dim, 8, 8)

    def forward(self, x, action_mask=None):
        action_mask = self._mask(action_mask, x.device)
        hidden = self.stem(self.embed(x, action_mask))
        value, aux = self.dual_head(hidden)
        return self.policy_head(hidden), value, aux

    def name(self):
        return 'CNN'

    @torch.no_grad()
    def predict(self, state, action_mask=None):
        """numpy in / numpy out (Othello/Network.py:229-261): probabilities over 65 actions,
        RELATIVE wdl, and the expected score utility atan(disc difference / score_scale) * 2/pi
        as an (n, 1) column; bf16 autocast off the CPU."""
        t = torch.from_numpy(state) if isinstance(state, np.ndarray) else state
        t = t.to(self.device, dtype=torch.float32)
        dev = torch.device(self.device).type
        with torch.autocast(dev, dtype=torch.bfloat16, enabled=dev != 'cpu'):
            log_prob, value_lp, aux = self(t, action_mask=action_mask)
        diff = aux.float() * float(self.aux_target_offset)
        utility = torch.atan(diff / float(getattr(self, 'score_scale', 8.0))) * (2.0 / float(np.pi))
        return (log_prob.float().exp().cpu().numpy(), value_lp.exp().float().cpu().numpy(),
                utility.view(-1, 1).cpu().numpy())

    @torch.no_grad()
    def policy(self, state, action_mask=None):
        return self.predict(state, action_mask)[0]

    @torch.no_grad()
    def value(self, state, action_mask=None):
        return self.predict(state, action_mask)[1]


def load_reference_weights(net, arrays):
    """arrays: mapping name -> ndarray/tensor with the reference checkpoint's tensors (e.g.
    tests/golden/g7_checkpoint_weights.npz, or torch.load(model.pt, weights_only=True))."""
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in arrays.items()}
    net.load_state_dict(sd, strict=True)
    return net
