"""Errors of the bf16 evaluators against fixture G7 (the reference's fp32 CPU outputs, shipped checkpoint):
the reference module under its own bf16 autocast, and the HIP inference twin."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "alphazero-al_amd"), ROOT, os.path.join(ROOT, "tests")]
import torch
from src import az_net
from src.fast_net import FastConnect4Net
from test_oracle_golden import load
g = load("g7_network"); wts = load("g7_checkpoint_weights")
net = az_net.Connect4Net(device="cuda").eval()
az_net.load_reference_weights(net, {k: wts[k] for k in wts.files})
b, t = g["boards"], g["turns"]
planes = np.stack([(b == t[:, None, None]), (b == -t[:, None, None]), np.ones_like(b) * t[:, None, None]], 1).astype(np.float32)
masks = g["masks"].astype(bool)
def err(tag, p, w, ml):
    qp = np.quantile(np.abs(p - g["ckpt_probs"]), [0.5, 0.9, 0.99, 0.999]); qw = np.quantile(np.abs(w - g["ckpt_wdl"]), [0.5, 0.9, 0.99, 0.999])
    print("   quantiles 50/90/99/99.9 %%: probs %s | wdl %s" % (np.round(qp, 5), np.round(qw, 5)))
    print("%-28s probs max %.4f mean %.5f | wdl max %.4f mean %.5f | ml max %.3f | argmax agree %.3f" % (
        tag, np.abs(p - g["ckpt_probs"]).max(), np.abs(p - g["ckpt_probs"]).mean(), np.abs(w - g["ckpt_wdl"]).max(),
        np.abs(w - g["ckpt_wdl"]).mean(), np.abs(ml.reshape(-1) - g["ckpt_ml"].reshape(-1)).max(),
        (p.argmax(1) == g["ckpt_probs"].argmax(1)).mean()))
err("module, bf16 autocast", *net.predict(planes, masks))
fast = FastConnect4Net.from_module(net)
err("HIP twin (bf16)", *fast.predict(planes, masks))
fast.hip = False
err("twin on torch ops (bf16)", *fast.predict(planes, masks))
err("twin fp32", *FastConnect4Net.from_module(net, dtype=torch.float32).predict(planes, masks))

# ---- where the twin's error comes from: body (stem .. attention) against heads
import ctypes as C
import torch.nn.functional as F
fast = FastConnect4Net.from_module(net)
f32 = FastConnect4Net.from_module(net, dtype=torch.float32)
x = torch.from_numpy(planes).cuda()
m = torch.from_numpy(masks).cuda()


def heads_fp32(t):
    """the two heads of FastConnect4Net.forward in fp32 on given final tokens"""
    s, bsz, c_dim = f32, t.shape[0], 64
    t = t.float()
    pn = s._rms(t, s.p_norm).view(bsz, 6, 7, c_dim)
    scores = (pn * s.p_gate_w).sum(-1) + s.p_gate_b
    wts_ = torch.softmax(scores, dim=1)
    col = (wts_.unsqueeze(-1) * pn).sum(dim=1)
    col = F.silu(F.linear(col, s.p_fc_w, s.p_fc_b))
    logits = (col * s.p_out_w).sum(-1) + s.p_out_b
    logits = logits.masked_fill(~m, -1e9)
    p = torch.softmax(logits, -1)
    gg = t.mean(dim=1)
    gg = gg + F.silu(F.linear(s._rms(gg, s.d_pool_norm), s.d_pool_w, s.d_pool_b))
    hh = s._rms(F.silu(F.linear(s._rms(gg, s.d_norm), s.d_fc_w, s.d_fc_b)), s.d_out_norm)
    w = torch.softmax(F.linear(hh, s.d_val_w, s.d_val_b), -1)
    ml = torch.sigmoid((hh * s.d_aux_w).sum(-1) + s.d_aux_b) * 42.0
    return p.cpu().numpy(), w.cpu().numpy(), ml.cpu().numpy()


def body_fp32(x):
    s, bsz = f32, x.shape[0]
    own = x[:, 0].reshape(bsz, 42, 1); opp = x[:, 1].reshape(bsz, 42, 1)
    t = torch.addcmul(torch.addcmul(s.pos, own, s.emb_own), opp, s.emb_opp)
    t = F.silu(s._conv(t, s.stem_w, s.stem_b))
    stages = [t]
    for w_, b_, g_, be_ in s.res:
        y = s._group_norm1(t, getattr(s, g_), getattr(s, be_))
        t = t + F.silu(s._conv(y, getattr(s, w_), getattr(s, b_)))
        stages.append(t)
    h = s._rms(t, s.pre_w)
    qkvg = F.linear(h, s.qkvg_w)
    q, k, v = qkvg[..., :192].view(bsz, 42, 3, 4, 16).unbind(2)
    gate = qkvg[..., 192:]
    q = s._rms(q, s.qn_w).transpose(1, 2); k = s._rms(k, s.kn_w).transpose(1, 2)
    a = F.scaled_dot_product_attention(q, k, v.transpose(1, 2))
    a = a * torch.sigmoid(gate).transpose(1, 2).unsqueeze(-1)
    t = F.linear(a.transpose(1, 2).reshape(bsz, 42, 64), s.o_w) + t
    stages.append(t)
    return t, stages


with torch.no_grad():
    t_hip, bsz, L, s_ = fast._body_hip(x)
    torch.cuda.synchronize()
    t_ref, stages = body_fp32(x)
    print("final tokens: HIP body vs fp32 body: max %.4f mean %.5f (|t| mean %.3f)" % (
        (t_hip.float() - t_ref).abs().max().item(), (t_hip.float() - t_ref).abs().mean().item(), t_ref.abs().mean().item()))
    err("fp32 heads on HIP body tokens", *heads_fp32(t_hip))
    # HIP heads on the fp32 body's tokens (rounded to bf16 once)
    tb = t_ref.to(torch.bfloat16).contiguous()
    probs = torch.empty((bsz, 7), dtype=torch.float32, device="cuda"); wdl = torch.empty((bsz, 3), dtype=torch.float32, device="cuda")
    ml = torch.empty((bsz,), dtype=torch.float32, device="cuda")
    mm = m.contiguous()
    L.az_nn_heads(tb.data_ptr(), C.byref(fast._heads_w), mm.data_ptr(), probs.data_ptr(), wdl.data_ptr(), ml.data_ptr(), bsz, 1e-5, None, None, s_)
    torch.cuda.synchronize()
    err("HIP heads on fp32 body tokens", probs.cpu().numpy(), wdl.cpu().numpy(), ml.cpu().numpy())
    err("fp32 heads on bf16(fp32 tokens)", *heads_fp32(tb))
    # stage by stage: the HIP kernels fed with the fp32 body's previous stage
    cur = stages[0].to(torch.bfloat16).contiguous()
    for i, (w_, b_, g_, be_) in enumerate(fast.res):
        y = torch.empty_like(cur)
        L.az_nn_conv_block(cur.data_ptr(), 64, getattr(fast, w_).data_ptr(), getattr(fast, b_).data_ptr(), getattr(fast, g_).data_ptr(),
                           getattr(fast, be_).data_ptr(), 1, y.data_ptr(), bsz, 1e-5, None, s_)
        torch.cuda.synchronize()
        d = (y.float() - stages[i + 1]).abs()
        print("block %d alone (input = bf16 of the fp32 stage): max %.4f mean %.5f (|t| mean %.3f)" % (i, d.max().item(), d.mean().item(), stages[i + 1].abs().mean().item()))
        cur = stages[i + 1].to(torch.bfloat16).contiguous()
