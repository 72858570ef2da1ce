"""Is the step host-bound?  Host time to ENQUEUE one step (Python + ctypes + launches) vs. the GPU time per step."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from video_rep_learning_amd.utils import presets
from video_rep_learning_amd.utils.optimizer import construct_optimizer
from video_rep_learning_amd.models import build_model
from video_rep_learning_amd.algos import get_algo
from video_rep_learning_amd.train import DataParallelModel
cfg = presets.baseline_config_2()
dev = torch.device('cuda')
torch.manual_seed(1)
model = build_model(cfg, 0).to(dev)
wrapped = DataParallelModel(model); opt = construct_optimizer(wrapped, cfg); algo = get_algo(cfg); model.train()
b, t, s = 4, 32, 224
videos = torch.randn(b, 2, t, 3, s, s, device=dev)
seq_lens = torch.full((b, 2), 100, dtype=torch.long, device=dev)
steps = torch.sort(torch.randint(0, 100, (b, 2, t)), dim=-1)[0].to(dev)
masks = torch.ones(b, 2, t, device=dev)
def step():
    wrapped.prefetch(videos); opt.zero_grad()
    loss = algo.compute_loss(wrapped, videos, seq_lens, steps, masks)['loss']; loss.backward(); opt.step(max_norm=10.0)
wrapped.prefetch(videos)
for _ in range(5): step()
torch.cuda.synchronize()
# host-only cost: enqueue 3 steps right after a sync (queues empty, nothing blocks), time the Python side
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
print('host enqueue time per step (queues empty): %s ms' % ' '.join('%.2f' % (x * 1e3) for x in ts))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('20 steps: host loop %.2f ms/step, wall incl. drain %.2f ms/step' % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
