"""Stand-alone timing of the evaluator kernels on synthetic activations (GPU box only).

    python tools/probe_nn.py                  # all kernels, HIP-event timings at 32768 leaves
    python tools/probe_nn.py conv 0 5         # only the residual conv block, debug mode 0, 5 launches
                                              # (the form to put under rocprofv3 --pmc ...)
Debug modes of the conv block (az_nn_debug): 1 skips the MFMA phase and its epilogue, 2 skips the
epilogue and the stores, 3 both: what is left is staging + GroupNorm.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alphazero-al_amd"))
from src.fast_net import FastConnect4Net, glue  # noqa: E402
from src.az_net import Connect4Net  # noqa: E402

L = glue()
L.az_nn_debug.argtypes = [C.c_int]
B = int(os.environ.get("PROBE_B", 32768))
bf = torch.bfloat16
dev = "cuda"
x = torch.randn(B, 42, 64, device=dev).to(bf)
y = torch.empty_like(x)
x32 = torch.randn(B, 42, 32, device=dev).to(bf)
w = (torch.randn(64, 64, 3, 3, device=dev) * 0.05).to(bf).contiguous(memory_format=torch.channels_last)
w32 = (torch.randn(64, 32, 3, 3, device=dev) * 0.05).to(bf).contiguous(memory_format=torch.channels_last)
b = torch.randn(64, device=dev).to(bf)
g = torch.ones(64, device=dev).to(bf)
be = torch.zeros(64, device=dev).to(bf)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
net = FastConnect4Net.from_module(Connect4Net(device=dev).eval())
probs = torch.empty(B, 7, device=dev)
wdl = torch.empty(B, 3, device=dev)
ml = torch.empty(B, device=dev)
mask = torch.ones(B, 7, dtype=torch.uint8, device=dev)


def conv():
    L.az_nn_conv_block(x.data_ptr(), 64, w.data_ptr(), b.data_ptr(), g.data_ptr(), be.data_ptr(), 1, y.data_ptr(), B, 1e-5, None, s)


def stem():
    L.az_nn_conv_block(x32.data_ptr(), 32, w32.data_ptr(), b.data_ptr(), None, None, 0, y.data_ptr(), B, 1e-5, None, s)


def attn():
    L.az_nn_attn_block(x.data_ptr(), net.pre_w.data_ptr(), net.qkvg_w.data_ptr(), net.qn_w.data_ptr(),
                       net.kn_w.data_ptr(), net.o_w.data_ptr(), y.data_ptr(), B, 1e-5, None, s)


def heads():
    L.az_nn_heads(x.data_ptr(), C.byref(net._heads_w), mask.data_ptr(), probs.data_ptr(), wdl.data_ptr(), ml.data_ptr(),
                  B, 1e-5, None, None, s)


feat = (torch.rand(B, 3, 6, 7, device=dev) > 0.5).float()
L.az_nn_stem_embed.argtypes = [C.c_void_p] * 7 + [C.c_int64] + [C.c_void_p] * 3


def stem_embed():
    L.az_nn_stem_embed(feat.data_ptr(), net.emb_own.data_ptr(), net.emb_opp.data_ptr(), net.pos.data_ptr(),
                       w32.data_ptr(), b.data_ptr(), y.data_ptr(), B, None, None, s)


def stem_folded():
    L.az_nn_stem_folded(feat.data_ptr(), net.stem_frag.data_ptr(), net.stem_pmap.data_ptr(), y.data_ptr(), B, None, None, s)


KERNELS = {"conv": conv, "stem": stem, "stem_embed": stem_embed, "stem_folded": stem_folded, "attn": attn, "heads": heads}


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if len(sys.argv) > 1:
    fn = KERNELS[sys.argv[1]]
    L.az_nn_debug(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 5):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("PROBE_TIME"):
        print("%-20s %7.1f us at %d leaves" % (sys.argv[1], timed(fn), B))
else:
    for mode, name in ((0, "full"), (1, "no MFMA phase"), (2, "no epilogue/store"), (3, "staging + norm only")):
        L.az_nn_debug(mode)
        print("conv %-22s %7.1f us" % (name, timed(conv)))
    L.az_nn_debug(16)
    print("conv with phase stamps      %7.1f us" % timed(conv))
    import numpy as np
    buf = np.zeros(2048 * 8, dtype=np.uint64)
    L.az_nn_conv_profile.argtypes = [C.c_void_p, C.c_int]
    L.az_nn_conv_profile(buf.ctypes.data, buf.size)
    ph = buf.reshape(2048, 8)[:, :6].astype(np.float64)
    names = ("P1 norm->img", "barrier 1", "MFMA+epilogue", "wait staged tile", "barrier 2", "P3 store")
    tot = ph.sum(1).mean()
    for k, nm in enumerate(names):
        print("   %-18s mean %9.0f cycles/wave (%4.1f%%)  min %9.0f max %9.0f" % (nm, ph[:, k].mean(), 100 * ph[:, k].mean() / tot, ph[:, k].min(), ph[:, k].max()))
    print("   total %9.0f cycles per wave (memtime ticks, 100 MHz?)" % tot)
    L.az_nn_debug(0)
    for name in ("stem", "stem_embed", "attn", "heads"):
        print("%-27s %7.1f us" % (name, timed(KERNELS[name])))
