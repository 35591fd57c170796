"""Training iterations at the reference's shipped configuration (configs/train.yaml: 128^2 crop, 16 + 4 samples, K = 1, D at
128^2) for rocprofv3 --kernel-trace: OI_DBG_IT iterations."""
import os, sys, copy, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd"):
    sys.path.insert(0, p)
import torch
import bench
n = int(os.environ.get("OI_DBG_IT", 10))
a = argparse.Namespace(res=128, samples=16, importance=4, up_steps=1, batch=1, train_steps=n, eager_d_steps=False, precision="f16x3")
dev = torch.device("cuda", 0)
g, d = bench.build_models(128, 16, 4, 1, "f16x3", dev)
g.train()
r = bench.bench_training(a, g, d, dev, 1, torch.cuda.synchronize, False, sub_legs=False)
print(r)
