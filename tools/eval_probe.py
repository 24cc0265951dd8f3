import time, torch, numpy as np
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import temp_amd
from tests.window_cases import build_window_model
from tests.golden_util import load
dev = torch.device("cuda:0")
z = load("G10_bi_grrgcn_rol")
m = build_window_model(z, dev)
t_list = torch.tensor([int(t) for t in z["t_list"]])
for i in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    ranks, loss = m.evaluate(t_list, val=False)
    torch.cuda.synchronize(); print("evaluate ms", (time.time() - t0) * 1e3, ranks.shape, float((1.0 / ranks.float()).mean()))
