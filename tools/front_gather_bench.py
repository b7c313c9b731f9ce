"""The gather the model step executes, on HBM-resident tables (bench.py `roofline_gather_in_step`): front_fwd_kernel at d = 64 on a
16 M x 64 table and embed_fwd_kernel at d = 256 on the C5 table, one step's tokens per launch.  One JSON line per case; run under
rocprofv3 by tools/collect_profiles.sh for the kernel stats and the HBM-read pass."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

for r in bench.front_gather_roofline(torch.device("cuda", 0)):
    print(json.dumps(r))
