"""Host time to enqueue one trainer step (no device sync inside the loop) next to the device time per step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cinema_amd.trainer import HotPathTrainer, SubjectStore  # noqa: E402

dev = torch.device("cuda:0")
n = 25000
args = bench.make_args(512, "bf16", "mse", n)
torch.manual_seed(42)
store = SubjectStore.synthetic(8, (200,) * 3, 155.0 / 200, [155, 155, 155], 2, 10, seed=42, device=dev)
store.shuffle(torch.Generator(device=dev).manual_seed(1), group_chunk=n)
tr = HotPathTrainer(args, store, dev)
for i in range(10):
    tr.step(i * n, (i + 1) * n)
torch.cuda.synchronize()
K = 50
t0 = time.perf_counter()
for i in range(10, 10 + K):
    tr.step(i * n, (i + 1) * n)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / K:.3f} ms/step, until device idle {1e3 * (t2 - t0) / K:.3f} ms/step")
