#!/usr/bin/env python3
"""Is the inference step ever HOST-bound?  Per step: the host time to ENQUEUE it (no synchronisation) against the GPU time between two events
around it, and the GPU's idle share inside the step (sum of kernel times from the conv roofline events is not available here; idle = step -
back-to-back GPU time of the same launches measured with the queue kept 3 steps deep).   python tools/debug/host_enqueue_probe.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from casapose_amd.pose_models.tfkeras import Classifiers
from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted

dev = torch.device("cuda:0")
B, H, W, seg_dim, kp = 16, 480, 640, 9, 9
net = Classifiers.get("casapose_c_gcu5")(ver_dim=27, seg_dim=seg_dim, input_shape=(H, W, 3), input_segmentation_shape=None, weights=None, base_model="resnet18", device=dev, seed=1237)
img = (2.0 * torch.rand(B, H, W, 3, generator=torch.Generator(device="cpu").manual_seed(1)) - 1.0).to(dev)
voter = CoordLSVotingWeighted(name="coords_ls_voting", num_classes=seg_dim, num_points=kp, filter_estimates=True)


def step():
    out = net([img], training=False)
    s, d, c = torch.split(out, [seg_dim, 2 * kp, kp], dim=3)
    return voter([s, d, c])


for _ in range(5):
    step()
torch.cuda.synchronize()
# (1) host enqueue time per step with an empty queue in front of every step (synchronise first): pure host cost of one step's launches
host = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
# (2) free-running: 20 steps, wall time per step
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
free = (time.perf_counter() - t0) / 20
# (3) GPU time of one step when its launches are already queued behind a long-running kernel (host fully ahead): events around the step
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
big = torch.empty(1 << 28, device=dev)
gpu = []
for _ in range(5):
    torch.cuda.synchronize()
    for _ in range(40):
        big.mul_(1.0001)       # ~40 x 0.45 ms of queued work: the host enqueues the whole step while the GPU is still busy with these
    e0.record()
    step()
    e1.record()
    torch.cuda.synchronize()
    gpu.append(e0.elapsed_time(e1))
print("host time to enqueue one step (empty queue): %.2f ms (min %.2f)" % (1e3 * np.mean(host), 1e3 * np.min(host)))
print("free-running step: %.3f ms" % (1e3 * free))
print("GPU time of one step with the host far ahead: %.3f ms (min %.3f)" % (np.mean(gpu), np.min(gpu)))
print("logical CPUs %d, cpu.max %s" % (os.cpu_count(), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?"))
