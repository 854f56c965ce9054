import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
print(bench.config5_md17(dev, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 20, cpu_steps=0))
