import sys, ctypes as C, torch
sys.path.insert(0, "alphazero-al_amd")
from src.fast_net import glue
L = glue(); L.az_nn_debug.argtypes=[C.c_int]
B = 32768; bf = torch.bfloat16
x = torch.randn(B,42,64,device="cuda").to(bf); y = torch.empty_like(x)
w = torch.randn(64,64,3,3,device="cuda").to(bf).contiguous(memory_format=torch.channels_last)
b = torch.randn(64,device="cuda").to(bf); g = torch.ones(64,device="cuda").to(bf); be = torch.zeros(64,device="cuda").to(bf)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(mode, n=20):
    L.az_nn_debug(mode)
    for _ in range(3): L.az_nn_conv_block(x.data_ptr(),64,w.data_ptr(),b.data_ptr(),g.data_ptr(),be.data_ptr(),1,y.data_ptr(),B,1e-5,s)
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): L.az_nn_conv_block(x.data_ptr(),64,w.data_ptr(),b.data_ptr(),g.data_ptr(),be.data_ptr(),1,y.data_ptr(),B,1e-5,s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for mode,name in ((0,"full"),(1,"no MFMA"),(2,"no store"),(3,"load+norm only")):
    print(name, "%.1f us"%run(mode))
