"""A few fused scoring calls of `users` (default 16 384) x 50 000 items x d (default 64) for PMC passes over the filter kernels -- the
bf16-candidate path (k_score_prefilter / k_score_prefilter_wide / k_rescore) or, with a third argument `fp32`, the fp32 MFMA filter:
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d out -- python tools/prefilter_pmc.py [d] [users] [fp32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textgcn_amd import scoring  # noqa: E402

dev = torch.device('cuda:0')
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
b = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
g = torch.Generator().manual_seed(0)
ue = (torch.randn(b, d, generator=g) * 0.1).to(dev)
ie = (torch.randn(50000, d, generator=g) * 0.1).to(dev)
fp32 = len(sys.argv) > 3 and sys.argv[3] == 'fp32'
pack = None if fp32 else scoring.item_pack(ie)
for _ in range(3):
    scoring.score_topk(ue, ie, 40, prefilter=not fp32, item_pack=pack)
torch.cuda.synchronize()
