import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unsupervised_detection_amd import ops
from unsupervised_detection_amd._devel import dbg
g = torch.Generator().manual_seed(0)
n, h, w, cout = 4, 48, 96, 128
for act in ("leaky", "elu"):
    for v in (0, 2, 4):
        for cin in (8, 128):
            x = (torch.rand(n, h, w, cin, generator=g) - 0.5).cuda()
            wt = ((torch.rand(3, 3, cin, cout, generator=g) - 0.5) * (2.0 / (9 * cin)) ** 0.5).cuda()
            b = torch.zeros(cout).cuda()
            dbg.udet_debug_force_conv((1 << 25) + v, 0, 1)
            for _ in range(20):
                ops.conv2d(x, wt, b, 1, 1, act, 0.1, False)
            torch.cuda.synchronize()
