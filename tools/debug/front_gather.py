import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
for r in bench.front_gather_roofline(torch.device("cuda", 0)):
    print(json.dumps(r))
