import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/object-intrinsics_amd", ROOT + "/oracle", ROOT + "/tests"):
    sys.path.insert(0, p)
import torch
from conftest import load_golden
from test_gpu_modules import make_renderer
col_sd = load_golden("weights_col" + "or")
r = make_renderer(col_sd, 16, 16, 1, "f16x3")
for N in (0, 1, 31, 33):
    try:
        ro = torch.tensor([0.0, 0.0, -3.0]).expand(N, 3).contiguous().cuda()
        rd = torch.tensor([0.0, 0.0, 1.0]).expand(N, 3).contiguous().cuda()
        near, far = torch.full((N, 1), 2.0).cuda(), torch.full((N, 1), 4.0).cuda()
        with torch.no_grad():
            out = r.render(ro, rd, near, far, perturb_overwrite=0, cos_anneal_ratio=0.5, w=torch.zeros(1, 64).cuda())
        torch.cuda.synchronize()
        print(N, "ok", tuple(out["color_fine"].shape), bool(torch.isfinite(out["color_fine"]).all()))
    except Exception as ex:
        print(N, "EXC", type(ex).__name__, str(ex)[:200])
