"""The algebra behind the folded kernels, in torch fp32 on the CPU (no GPU, no HIP): what fast_net.fold_stem hands
nn_stem.hip and what fast_net.fold_block hands nn_conv2.hip reproduce the layers they replace."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [p for p in (os.path.join(ROOT, "alphazero-al_amd"), ROOT) if p not in sys.path]


def _unfrag(frag):
    """(2, 4, 64, 8) fragments -> the (64, 32) table hi + lo they encode (include/az_nn.h, az_nn_stem_folded)."""
    t = torch.zeros((64, 32), dtype=torch.float64)
    for i in range(4):
        for lane in range(64):
            r = lane & 15
            ch = 32 * (i // 2) + 8 * (r // 4) + 4 * (i % 2) + r % 4
            for j in range(8):
                k = 8 * (lane >> 4) + j
                t[ch, k] = float(frag[0, i, lane, j]) + float(frag[1, i, lane, j])
    return t


def test_fold_stem_reproduces_embedding_and_convolution():
    """conv3x3(own * e_own + opp * e_opp + pos) + bias == pmap[cell] + sum over the taps inside the board of
    own[n] * T[:, 2 tap] + opp[n] * T[:, 2 tap + 1]   (Network.py:168-170, 226-239) on random boards; the bf16 high + low
    split of T carries it to ~2^-16 relative."""
    from src.fast_net import fold_stem
    torch.manual_seed(5)
    w = torch.randn(64, 32, 3, 3) * 0.1
    bias = torch.randn(64) * 0.3
    e_own, e_opp = torch.randn(32), torch.randn(32)
    pos = torch.randn(42, 32)
    frag, pmap = fold_stem(w, bias, e_own, e_opp, pos)
    assert tuple(frag.shape) == (2, 4, 64, 8) and frag.dtype == torch.bfloat16 and tuple(pmap.shape) == (48, 68)
    assert pmap[42:].abs().max() == 0 and pmap[:, 64:].abs().max() == 0
    table = _unfrag(frag)
    assert table[:, 18:].abs().max() == 0
    wb, bb = w.to(torch.bfloat16).double(), bias.to(torch.bfloat16).double()
    state = torch.randint(0, 3, (64, 6, 7))
    own, opp = (state == 1).double(), (state == 2).double()
    tokens = own.reshape(64, 42, 1) * e_own.double() + opp.reshape(64, 42, 1) * e_opp.double() + pos.double()
    want = torch.nn.functional.conv2d(tokens.transpose(1, 2).reshape(64, 32, 6, 7), wb, bb, padding=1)   # (B, 64, 6, 7)
    got = pmap[:42, :64].double().t().reshape(1, 64, 6, 7).repeat(64, 1, 1, 1)
    for tap in range(9):
        dy, dx = tap // 3 - 1, tap % 3 - 1
        for plane, board in ((0, own), (1, opp)):
            sh = torch.zeros_like(board)
            ys, xs = slice(max(0, -dy), 6 - max(0, dy)), slice(max(0, -dx), 7 - max(0, dx))
            yd, xd = slice(max(0, dy), 6 + min(0, dy)), slice(max(0, dx), 7 + min(0, dx))
            sh[:, ys, xs] = board[:, yd, xd]                       # sh[r, c] = board[r + dy, c + dx] inside the board
            got = got + sh.unsqueeze(1) * table[:, 2 * tap + plane].reshape(1, 64, 1, 1)
    err = (got - want).abs().max().item()
    assert err < 2e-4 * want.abs().max().item(), err


def test_fold_block_reproduces_groupnorm_convolution():
    """conv(W, pad(GroupNorm1(x) * gamma + beta)) + bias == rstd * (conv(W * gamma, pad(x)) - mean * t1[class]) + t2[class]
    with the nine border classes of fast_net.fold_block (t2 comes scaled by log2 e)."""
    from src.fast_net import fold_block
    torch.manual_seed(6)
    w = (torch.randn(64, 64, 3, 3) * 0.05).to(torch.bfloat16)
    bias, gamma, beta = torch.randn(64) * 0.2, torch.randn(64) * 0.3 + 1.0, torch.randn(64) * 0.2
    wf, t1, t2s = fold_block(w, bias, gamma, beta)
    x = torch.randn(5, 64, 6, 7, dtype=torch.float64)
    mean = x.mean((1, 2, 3), keepdim=True)
    rstd = 1.0 / torch.sqrt(x.var((1, 2, 3), unbiased=False, keepdim=True) + 1e-5)
    # the folded weight is rounded to bf16 once: compare with the convolution of THAT weight un-folded again
    w_eff = wf.double() / gamma.double().view(1, -1, 1, 1)
    xn = (x - mean) * rstd * gamma.double().view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1)
    want = torch.nn.functional.conv2d(xn, w_eff, None, padding=1)
    w32 = w.double()
    want = want + bias.double().view(1, -1, 1, 1) + torch.nn.functional.conv2d(
        torch.ones(1, 64, 6, 7, dtype=torch.float64) * beta.double().view(1, -1, 1, 1), w32 - w_eff, None, padding=1)
    raw = torch.nn.functional.conv2d(x, wf.double(), None, padding=1)
    cls = torch.zeros(6, 7, dtype=torch.long)
    for r in range(6):
        for c in range(7):
            cls[r, c] = 3 * (0 if r == 0 else 2 if r == 5 else 1) + (0 if c == 0 else 2 if c == 6 else 1)
    t1m = t1.double()[cls].permute(2, 0, 1).unsqueeze(0)          # (1, 64, 6, 7)
    t2m = (t2s.double() / 1.4426950408889634)[cls].permute(2, 0, 1).unsqueeze(0)
    got = rstd * (raw - mean * t1m) + t2m
    assert (got - want).abs().max().item() < 1e-3 * want.abs().max().item()
