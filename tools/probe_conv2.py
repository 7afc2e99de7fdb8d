"""Timing of the residual block's two kernels side by side on synthetic activations (GPU box only):
nn_conv.hip (az_nn_conv_block) against nn_conv2.hip (az_nn_conv_block2), interleaved rounds in one process.

    python tools/probe_conv2.py [leaves] [rounds]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alphazero-al_amd"))
from src.fast_net import fold_block, glue  # noqa: E402

L = glue()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 26368
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
bf, dev = torch.bfloat16, "cuda"
x = (torch.randn(B, 42, 64, device=dev) * 1.2 + 0.3).to(bf)
if os.environ.get("PROBE_ZERO") == "1":          # clock check: on zeros the chip holds its full clock (DVFS give-back)
    x.zero_()
if os.environ.get("PROBE_REAL") == "1":          # activations of the random-init network on random positions (what the bench sees)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import numpy as np
    import scenarios as S
    from src.az_net import Connect4Net
    from src.fast_net import FastConnect4Net
    torch.manual_seed(1234)
    net = FastConnect4Net.from_module(Connect4Net(device=dev).eval())
    bd, tn = S.random_openings(np.random.default_rng(0), 512, 24)
    planes = np.stack([(bd == tn[:, None, None]), (bd == -tn[:, None, None]), np.ones_like(bd) * tn[:, None, None]], 1).astype(np.float32)
    feat = torch.from_numpy(np.tile(planes, (B // 512 + 1, 1, 1, 1))[:B]).to(dev).contiguous()
    L.az_nn_stem_embed.argtypes = [C.c_void_p] * 7 + [C.c_int64] + [C.c_void_p] * 3
    L.az_nn_stem_embed(feat.data_ptr(), net.emb_own.data_ptr(), net.emb_opp.data_ptr(), net.pos.data_ptr(), net.stem_w.data_ptr(),
                       net.stem_b.data_ptr(), x.data_ptr(), B, None, None, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
y = torch.empty_like(x)
w = (torch.randn(64, 64, 3, 3, device=dev) * 0.05).to(bf)
if os.environ.get("PROBE_ZERO") == "1":
    w.zero_()
w_ohwi = w.contiguous(memory_format=torch.channels_last)
b = torch.randn(64, device=dev).to(bf)
g = (1 + 0.1 * torch.randn(64, device=dev)).to(bf)
be = (0.1 * torch.randn(64, device=dev)).to(bf)
if os.environ.get("PROBE_REAL") == "1":
    w = getattr(net, net.res[0][0]).clone(); b = getattr(net, net.res[0][1]).clone()
    g = getattr(net, net.res[0][2]).clone(); be = getattr(net, net.res[0][3]).clone()
    w_ohwi = w.contiguous(memory_format=torch.channels_last)
wf, t1, t2s = fold_block(w, b, g, be)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def v1():
    L.az_nn_conv_block(x.data_ptr(), 64, w_ohwi.data_ptr(), b.data_ptr(), g.data_ptr(), be.data_ptr(), 1, y.data_ptr(), B, 1e-5, None, s)


def v2():
    L.az_nn_conv_block2(x.data_ptr(), wf.data_ptr(), t1.data_ptr(), t2s.data_ptr(), y.data_ptr(), B, 1e-5, None, s)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if len(sys.argv) > 3:                 # under a profiler: a few launches of one kernel
    fn = {"v1": v1, "v2": v2}[sys.argv[3]]
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
else:
    flops = 2.0 * 42 * 64 * 576 * B
    import numpy as np
    for r in range(rounds):
        a, c = timed(v1), timed(v2)
        print("round %d: %d leaves  v1 %6.1f us (%4.1f %% of 2.5 PF)   v2 %6.1f us (%4.1f %%)   dbg=%s" %
              (r, B, a, flops / (a * 1e-6) / 2.5e15 * 100, c, flops / (c * 1e-6) / 2.5e15 * 100, os.environ.get("AZ_NN_CONV2_DBG", "0")), flush=True)
    st = np.zeros(256 * 4 * 4, dtype=np.uint64)
    L.az_nn_conv2_stamps.argtypes = [C.c_void_p, C.c_int]
    L.az_nn_conv2_stamps(st.ctypes.data, st.size)
    st = st.reshape(1024, 4).astype(np.float64)
    st = st[st[:, 2] > 0]
    cyc, ticks, tiles = st[:, 0], st[:, 1], st[:, 2]
    print("   tile loop of the last v2 launch: %.0f cycles per tile (%.1f per MFMA), in-kernel clock %.2f GHz, loop %.1f us, %d-%d tiles per workgroup" %
          (np.median(cyc / tiles), np.median(cyc / tiles) / 108, np.median(cyc / ticks) * 0.1, np.median(ticks) * 0.01, tiles.min(), tiles.max()))

