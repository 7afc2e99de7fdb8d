"""Timing of one az_nn_othello_conv layer (256 -> 256 channels, 10x10, pad 1) at 16384 samples.
AZ_OTH_DEBUG=1 keeps the weight stream in L1 (wrong results): the distance to that time is what
the L2 weight traffic costs."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "alphazero-al_amd"), ROOT]
import torch  # noqa: E402
from src.fast_net import glue  # noqa: E402
from src.fast_othello import pack_conv_weight  # noqa: E402

L = glue()
L.az_nn_othello_conv.argtypes = [C.c_void_p] * 8 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
x = torch.randn(B, 10, 10, 256, device="cuda").to(torch.bfloat16)
w = pack_conv_weight(torch.randn(256, 256, 3, 3, device="cuda") * 0.03)
ones, zeros = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
y = torch.empty_like(x)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(pre, res):
    return L.az_nn_othello_conv(x.data_ptr(), w.data_ptr(), ones.data_ptr() if pre else None, zeros.data_ptr() if pre else None,
                                ones.data_ptr(), zeros.data_ptr(), x.data_ptr() if res else None, y.data_ptr(), B, 256, 10, 1, 1, None, s)


for pre, res in ((False, False), (True, False), (True, True)):
    for _ in range(3):
        assert run(pre, res) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run(pre, res)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"pre={pre} res={res}: {ms:.3f} ms, {2 * B * 100 * 256 * 2304 / ms / 1e9:.0f} TFLOP/s (debug {os.environ.get('AZ_OTH_DEBUG', '0')})", flush=True)
