import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import torch, bench
gen, disc = bench.build_models(64, 64, 64, 1, "f16x3", torch.device("cuda"))
gen.train()
with torch.no_grad():
    for i in range(3): gen(bs=1, it=i, data={})
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
        for i in range(4): gen(bs=1, it=i, data={})
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60))
