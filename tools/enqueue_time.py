import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from demucs_cpp_amd import binding as dmx
from demucs_cpp_amd.weights import write_synthetic_model
write_synthetic_model('/tmp/pm4.bin', 4, 0)
m = dmx.Model('/tmp/pm4.bin')
for B in (1, 2, 4, 12):
    ctx = dmx.Context(m, 0, B)
    mix = torch.zeros((B, 343980, 2), device='cuda'); out = torch.zeros((B, 4, 2, 343980), device='cuda')
    torch.cuda.synchronize()
    for _ in range(3):
        ctx.segment_device(mix.data_ptr(), out.data_ptr(), B)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.segment_device(mix.data_ptr(), out.data_ptr(), B)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print(f'B={B}: enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms')
    ctx.close()
