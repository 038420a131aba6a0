#!/usr/bin/env python3
"""Headline benchmark: images/s of the CASAPose hot path on MI355X.

Workload at N=1 (BASELINE.json configs[1]): 8-object LMO inference, casapose_c_gcu5,
batch 16, 480x640, fp32 -- one step = network forward (encoder + both decoders, estimated
mask) + confidence-weighted LS keypoint voting, inputs already resident in HBM.  PnP stays
on the host in the reference and is outside the timed GPU path.  For N>1 every rank runs
an independent replica on its own batch (inference shards by image, no data-path
collective): weak scaling.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMD x 64 FLOP/clk x 2.4 GHz
TILE_NAMES = {1: "conv_f32_kernel<2,2,2,2> (128x128)", 2: "conv_f32_kernel<2,2,1,2> (64x128)", 3: "conv_f32_kernel<2,2,2,1> (128x64)",
              4: "conv_f32_kernel<4,1,1,1> (128x32)", 5: "conv_f32_kernel<2,2,1,1> (64x64)", 6: "conv_f32_kernel<4,1,2,1> (256x32)",
              7: "conv_halo_kernel (LDS-resident 4-row halo tile, cout <= 64)",
              8: "conv_stem_kernel (7x7/s2 stem, LDS-resident input halo)",
              100: "wino_gemm_kernel (persistent 64x64 grouped GEMM of the Winograd planes)",
              103: "wino_gemm_split_kernel (bf16 / fp16 pipe: the Winograd planes' GEMM as exact three-way bf16 splits or fp16 two-way splits; FLOPs counted as executed 2-byte FLOPs)",
              203: "conv_hsplit_kernel<3> (bf16 pipe, exact three-way split: six bf16 products per fp32 product; FLOPs counted as executed bf16 FLOPs)",
              218: "conv_hsplit_kernel<2> (fp16 pipe, two-way split: three fp16 products per fp32 product; FLOPs counted as executed fp16 FLOPs)",
              201: "conv_hsplit_kernel<1> (bf16 pipe, operands rounded to bf16)",
              301: "conv_bf16d_kernel (bf16 pipe, direct 3x3 of the deep layers, operands rounded to bf16)",
              208: "conv_stem_split_kernel (bf16 pipe, the 7x7/s2 stem as exact three-way splits or bf16 operands)"}
VALU_LANE_OPS_PEAK = 256 * 4 * 32 * 2.4e9   # fp32 vector lane-operations per second (an FMA counted once): 78.6e12
RANSAC_VALU_PER_HYPOTHESIS = 120               # v_* instructions per hypothesis (nine tests) in vote_kernel<64>'s loop, counted in the ISA (tools/debug/isa_stats.py family)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16; v_mfma_f32_32x32x16_f16 has the same rate)
# executed 2-byte products per fp32 product by `planes` code: exact bf16 split, hi + mid bf16 planes, fp16 two-way split (CP_PLANES_F16X2), bf16 operands
PRODUCTS = {3: 6.0, 2: 3.0, 0x12: 3.0, 1: 1.0}


def _log(msg):
    """progress on stderr (stdout carries exactly one JSON line)"""
    print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def _cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _median_rate(fn, images, warmup=3, timed=10, budget_s=12.0):
    """images/s of fn(), strictly time-boxed: the first (warm-up) call is clocked; if it is cheap, up to `warmup` - 1 more untimed
    calls follow; then as many timed calls as fit the rest of `budget_s` (at least 1, at most `timed`) -- the median of those."""
    t0 = time.perf_counter()
    fn()
    first = time.perf_counter() - t0
    for _ in range(warmup - 1):
        if (time.perf_counter() - t0) + first > budget_s / 3:
            break
        fn()
    left = budget_s - (time.perf_counter() - t0)
    n = max(1, min(timed, int(left / max(first, 1e-6))))
    ts = []
    for _ in range(n):
        t1 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t1)
    ts.sort()
    return images / ts[len(ts) // 2], len(ts)


def cpu_quota():
    """CPUs of run time the cgroup grants this process tree per period (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`), or None
    without a limit.  The GPU box shows 256 logical CPUs and grants 16 (`1600000 100000`): 128 busy threads there are throttled to an eighth."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / per if q > 0 else None
    except Exception:
        return None


def cpu_worker_shape():
    """(worker processes, threads per worker) of the CPU baseline; CASAPOSE_CPU_WORKERS / CASAPOSE_CPU_WORKER_THREADS override.
    Without a CPU quota: 8 threads per worker, one worker per 8 PHYSICAL cores, at most 16 (measured on the 2 x 64-core host when nothing
    throttled the box, profiles/probes/r05_cpu_workers_probe.txt).  Under a cgroup quota of Q CPUs (the GPU box: Q = 16 of 256 logical CPUs) the
    throttle, not the core count, is the resource: 4 threads per worker and Q / 2 workers -- twice the quota in threads, because threads waiting
    passively in a pool cost no quota -- measured 10.5-10.9 images/s there against 8.5-9.9 for 4 x 8 and 8.8 for 16 x 8
    (profiles/probes/r05_cpu_quota_probe.txt)."""
    cores, quota = os.cpu_count() or 1, cpu_quota()
    env_t, env_n = int(os.environ.get("CASAPOSE_CPU_WORKER_THREADS", "0")), int(os.environ.get("CASAPOSE_CPU_WORKERS", "0"))
    if quota is not None and quota < cores / 2:
        threads = env_t or 4
        workers = env_n or max(1, int(round(2.0 * quota / threads)))
    else:
        threads = env_t or 8
        workers = env_n or max(1, min(16, cores // (2 * threads)))   # half the logical CPUs: one thread per core under SMT 2
    return workers, threads


CPU_WORKERS, CPU_WORKER_THREADS = cpu_worker_shape()


def cpu_worker(args):
    """One CPU-baseline worker process (`bench.py --cpu-worker i/N`): pins itself to its own block of CPU_WORKER_THREADS logical CPUs, builds the CPU
    restatement of the inference graph (oracle/torch_train_ref.forward_infer_fast + component filter + LS voter: the same program as the
    in-process leg), warms up, then WAITS for a line on stdin -- the parent sends it when the GPU legs are over -- and runs whole images
    (forward + filter + voting, bs 1, its own seed) for --cpu-seconds; prints {"images": n, "seconds": t}.  Never touches a GPU."""
    idx, n = [int(v) for v in args.cpu_worker.split("/")]
    cores = os.cpu_count()
    first = (idx * CPU_WORKER_THREADS) % cores
    try:   # BEFORE torch / OpenMP exist in this process: their threads inherit the mask (set afterwards, 8 x 32 floating threads took 24 s per image)
        os.sched_setaffinity(0, set(range(first, min(first + CPU_WORKER_THREADS, cores))))
    except OSError:
        pass
    import numpy as np
    import torch
    from scipy import ndimage

    torch.set_num_threads(CPU_WORKER_THREADS)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import casapose_oracle as O
    import torch_train_ref as R

    seg_dim, ver_dim, h, w = 9, 27, args.height, args.width
    q = R.prepare_inference(R.to_torch(O.init_params(seg_dim, ver_dim, seed=1237, dtype=np.float32), dtype=torch.float32, requires_grad=False))
    gen = torch.Generator().manual_seed(1237 + idx)
    kp = ver_dim // 3

    def one():
        img = 2.0 * torch.rand(1, h, w, 3, generator=gen) - 1.0
        with torch.no_grad():
            out = R.forward_infer_fast(q, img)
            lab = out[..., :seg_dim].argmax(-1).numpy()
            keep = np.zeros_like(lab)
            for o in range(1, seg_dim):
                cc, ncomp = ndimage.label(lab[0] == o)
                if ncomp:
                    sizes = np.bincount(cc.ravel())[1:]
                    if sizes.max() >= 50:
                        keep[0][cc == 1 + int(sizes.argmax())] = o
            R.ls_voting_fast(torch.from_numpy(keep), out[..., seg_dim:seg_dim + 2 * kp], out[..., seg_dim + 2 * kp:], seg_dim - 1)

    print("ready", flush=True)   # network built; nothing has run yet: the host's cores stay free for the parent's GPU legs
    sys.stdin.readline()
    one()
    one()
    t0 = time.perf_counter()
    images = 0
    while time.perf_counter() - t0 < args.cpu_seconds:
        one()
        images += 1
    print(json.dumps({"images": images, "seconds": time.perf_counter() - t0, "worker": idx, "of": n, "first_cpu": first}), flush=True)


def spawn_cpu_workers(args):
    """Start the CPU-baseline workers BEFORE this process touches the GPU (a process that has initialised the GPU must not be the one that
    execs): cpu_worker_shape() processes, each on its own block of CPU_WORKER_THREADS logical CPUs.  They build their network and wait; collect_cpu_workers()
    releases them together once the GPU legs are done, so they never compete with the timed GPU region for host cores."""
    import subprocess

    cores = os.cpu_count() or 1
    n = CPU_WORKERS
    env = dict(os.environ, OMP_NUM_THREADS=str(CPU_WORKER_THREADS), MKL_NUM_THREADS=str(CPU_WORKER_THREADS), OMP_WAIT_POLICY="PASSIVE", KMP_BLOCKTIME="0",
               CASAPOSE_CPU_WORKERS=str(CPU_WORKERS), CASAPOSE_CPU_WORKER_THREADS=str(CPU_WORKER_THREADS),
               HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    procs = []
    for i in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "%d/%d" % (i, n), "--cpu-seconds", str(args.cpu_seconds),
                                       "--height", str(args.height), "--width", str(args.width)],
                                      stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env))
    return procs


def wait_cpu_workers(procs):
    """Block until every worker has built its network (or 120 s): called by the parent BEFORE its GPU warm-up, so that no worker is still importing
    or building while the GPU legs are timed.  Returns the workers that are up; the others are killed."""
    import select

    ready = []
    t0 = time.perf_counter()
    for p in procs or []:   # a worker that is not up after 120 s (first import of torch on a cold box) is left out
        try:
            r, _, _ = select.select([p.stdout], [], [], max(1.0, 120.0 - (time.perf_counter() - t0)))
            if r and p.stdout.readline().strip() == "ready":
                ready.append(p)
                continue
        except Exception:
            pass
        p.kill()
    return ready


def collect_cpu_workers(ready, budget_s):
    """Release the waiting workers at once and add up what they did: aggregate images/s = all images / the longest worker's time."""
    for p in ready:
        try:
            p.stdin.write("go\n")
            p.stdin.flush()
        except Exception:
            pass
    res = []
    for p in ready:
        try:
            out, _ = p.communicate(timeout=budget_s + 60)
            res.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:
            p.kill()
    if not res:
        return None
    images, longest = sum(r["images"] for r in res), max(r["seconds"] for r in res)
    return {"images_per_s": round(images / longest, 3), "processes": len(res), "threads_per_process": CPU_WORKER_THREADS, "images": images,
            "seconds": round(longest, 2), "per_process_images": [r["images"] for r in res]}


def cpu_baseline(h, w, seg_dim, ver_dim, batch, accuracy=None, workers=None, worker_seconds=12.0):
    """CPU restatement (PyTorch-CPU, oneDNN), NOT TensorFlow: the reference's TF-CPU path cannot run here, so the
    number beside the GPU is the oracle's torch restatement of the same inference graph (oracle/torch_train_ref.forward_train with
    training=False and the estimated mask) in fp32 on all host cores, as BASELINE.md 3 / SURVEY 8(d) prescribe: warm-up, then the
    median of the timed iterations, at bs 1 and at the bench batch; legs: forward, forward + component filter + LS voter.  Every leg
    is time-boxed (about 20 s in all, whatever the host).  Round 4: the timed graph is `forward_infer_fast` -- the same network written the way
    one would run it on a CPU (channels-last throughout, normalisation folded to one fused multiply-add, the partial convolution as one 1x1
    convolution to 9*Cout tap planes + nine masked accumulations); tests/test_train_oracle.py holds it equal to the plain restatement.
    A reported baseline, not the optimisation target."""
    import numpy as np
    import torch
    from scipy import ndimage

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import casapose_oracle as O
    import torch_train_ref as R

    cores = os.cpu_count()
    p = R.to_torch(O.init_params(seg_dim, ver_dim, seed=1237, dtype=np.float32), dtype=torch.float32, requires_grad=False)
    gen = torch.Generator().manual_seed(1237)
    objects, kp = seg_dim - 1, (ver_dim // 3)

    q = R.prepare_inference(p)   # folded normalisation tables, channels-last kernels, tap-major partial-convolution kernels (oracle/torch_train_ref.py)

    def forward(img):
        with torch.no_grad():   # the inference graph as a CPU program: channels-last end to end, fused affine + activation, masked tap accumulation
            return R.forward_infer_fast(q, img)

    def vote(out):
        lab = out[..., :seg_dim].argmax(-1).numpy()
        keep = np.zeros_like(lab)
        for bi in range(lab.shape[0]):  # largest 4-connected component of >= 50 px per object (voting_layers_2d.py:43-79)
            for o in range(1, seg_dim):
                cc, n = ndimage.label(lab[bi] == o)
                if n:
                    sizes = np.bincount(cc.ravel())[1:]
                    if sizes.max() >= 50:
                        keep[bi][cc == 1 + int(sizes.argmax())] = o
        with torch.no_grad():
            return R.ls_voting_fast(torch.from_numpy(keep), out[..., seg_dim:seg_dim + 2 * kp], out[..., seg_dim + 2 * kp:], objects)

    # threads: all host cores unless that is slower than 32 threads on this graph (a 2-socket, 256-thread host loses time in the thread
    # pool on the small layers); one bs-1 forward each decides, and `cores` below reports the count actually used
    probe = 2.0 * torch.rand(1, h, w, 3, generator=gen) - 1.0
    timing = {}
    quota = cpu_quota()
    counts = {min(cores, 32), cores} if quota is None or quota >= cores / 2 else {max(1, min(cores, 32, int(quota)))}   # under a cgroup quota: its CPUs, no more
    for n in sorted(counts):  # the safe count first; then ONE forward with every core, kept only if it is faster
        if timing and min(timing.values()) < 2.0:
            # a 256-thread host took 74 s for ONE image in the all-cores probe (thread-pool overhead on the small layers) against 0.86 s
            # with 32 threads: when the capped count already runs an image in under 2 s the all-cores probe is skipped
            _log("cpu baseline: %d-thread probe skipped (%.2f s per image with %d threads)" % (n, min(timing.values()), min(timing, key=timing.get)))
            break
        torch.set_num_threads(n)
        if not timing:
            forward(probe)
        t0 = time.perf_counter()
        forward(probe)
        timing[n] = time.perf_counter() - t0
        _log("cpu baseline: %d threads -> %.2f s per image" % (n, timing[n]))
    threads = min(timing, key=timing.get)
    torch.set_num_threads(threads)
    legs = {}
    for bs, budget in ((1, 4.0), (batch, 6.0)):
        if bs > 1:  # >= 5 timed iterations per leg inside its budget: shrink the batch of the second leg to what runs in ~1 s, and say so in its key
            rate1 = legs["bs1"]["forward_images_per_s"]
            bs = max(2, min(bs, int(1.0 * rate1)))
        img = 2.0 * torch.rand(bs, h, w, 3, generator=gen) - 1.0
        fwd, n1 = _median_rate(lambda: forward(img), bs, budget_s=budget)
        both, n2 = _median_rate(lambda: vote(forward(img)), bs, warmup=1, budget_s=budget)
        _log("cpu baseline: bs %d forward %.3f images/s, with voting %.3f" % (bs, fwd, both))
        legs["bs%d" % bs] = {"forward_images_per_s": round(fwd, 3), "forward_plus_filter_plus_ls_images_per_s": round(both, 3), "timed_iterations": [n1, n2]}
    best = max(v["forward_plus_filter_plus_ls_images_per_s"] for v in legs.values())
    acc = None
    if accuracy and accuracy["logits"]:
        # the checker's other job in this leg: ONE image of the bench batch through the same network (the bench's parameters) in fp64 on the CPU;
        # the segmentation logits of every arithmetic the GPU ran in this line against it (max |difference| / max |logit|).  Logits only: the
        # vector fields are conditioned on the arg-max label map, which random weights put on ties.
        t0 = time.perf_counter()
        p64 = R.to_torch({k: np.asarray(v) for k, v in accuracy["params"].items()}, dtype=torch.float64, requires_grad=False)
        q64 = R.prepare_inference(p64)
        img64 = torch.from_numpy(accuracy["image"]).double()
        with torch.no_grad():
            ref = R.forward_infer_fast(q64, img64)[..., :seg_dim].numpy()
        den = float(np.abs(ref).max())
        # ... and the VECTOR FIELD (round 5): decoder 2 is conditioned on the hard label map, which random weights put on arg-max ties, so each mode's
        # field is compared with an fp64 evaluation conditioned on that mode's OWN label map (the one its forward computed on the device)
        vec, vref = {}, {}
        for m, lab in accuracy.get("labels", {}).items():
            key = lab.tobytes()
            if key not in vref:
                with torch.no_grad():
                    vref[key] = R.forward_infer_fast(q64, img64, labels=torch.from_numpy(lab.astype(np.int64)))[..., seg_dim:].numpy()
            vec[m] = float("%.3g" % (float(np.abs(accuracy["vertex"][m].astype(np.float64) - vref[key]).max()) / float(np.abs(vref[key]).max())))
        acc = {"what": "max |segmentation logit - fp64 logit| / max |fp64 logit| for image 0 of the bench batch, per convolution arithmetic run in this line "
                       "(fp64 = the CPU restatement in double precision with the bench's parameters); vector_field_per_conv_mode: the same for the 27 "
                       "vector-field channels, the fp64 evaluation conditioned on the label map the device computed in that mode",
               "per_conv_mode": {m: float("%.3g" % (float(np.abs(v.astype(np.float64) - ref).max()) / den)) for m, v in accuracy["logits"].items()},
               "vector_field_per_conv_mode": vec, "label_maps_compared": len(vref),
               "seconds": round(time.perf_counter() - t0, 1)}
        _log("cpu baseline: fp64 accuracy check %s, vector field %s" % (acc["per_conv_mode"], vec))
    multi = collect_cpu_workers(workers, worker_seconds) if workers else None
    if multi and multi["images_per_s"] > best:
        # the way one would actually run this on the box's host: several worker processes, each on its own block of cores, on disjoint images
        _log("cpu baseline: %d processes x %d threads -> %.2f images/s" % (multi["processes"], multi["threads_per_process"], multi["images_per_s"]))
        return {"value": multi["images_per_s"], "accuracy_vs_fp64": acc, "unit": "images/s", "cores": multi["processes"] * multi["threads_per_process"],
                "shape": "%d processes x %d threads, each pinned to its own block of logical CPUs, disjoint images" % (multi["processes"], multi["threads_per_process"]),
                "host_cores": cores, "cpu_quota": cpu_quota(), "kind": "port", "what": "CPU restatement (PyTorch-CPU fp32, oneDNN), not TensorFlow", "cpu": _cpu_model_name(),
                "sample": "%dx%d: every worker runs whole images (forward + component filter + LS voting, bs 1) for %.0f s after two warm-up images; value = all "
                          "images / the longest worker's time" % (h, w, worker_seconds),
                "workers": multi, "single_process": {"value": round(best, 3), "threads": threads, "legs": legs,
                                                     "thread_probe_s_per_image": {str(k): round(v, 3) for k, v in timing.items()}}}
    return {"value": round(best, 3), "accuracy_vs_fp64": acc, "unit": "images/s", "cores": threads, "host_cores": cores, "cpu_quota": cpu_quota(), "kind": "port",
            "what": "CPU restatement (PyTorch-CPU fp32, oneDNN, %d threads), not TensorFlow" % threads, "cpu": _cpu_model_name(),
            "thread_probe_s_per_image": {str(k): round(v, 3) for k, v in timing.items()}, "workers": multi,
            "sample": "%dx%d, %s: warm-up + median of <= 10 timed iterations per leg (time-boxed); value = best forward + "
                      "component filter + LS voting rate" % (h, w, " and ".join(sorted(legs))), "legs": legs}


TILE_PMC_PREFIX = {1: "conv_f32_kernel<2, 2, 2, 2,", 2: "conv_f32_kernel<2, 2, 1, 2,", 3: "conv_f32_kernel<2, 2, 2, 1,", 4: "conv_f32_kernel<4, 1, 1, 1,",
                   5: "conv_f32_kernel<2, 2, 1, 1,", 6: "conv_f32_kernel<4, 1, 2, 1,", 7: "conv_halo_kernel<", 8: "conv_stem_kernel", 100: "wino_gemm_kernel",
                   203: "conv_hsplit_kernel<", 218: "conv_hsplit_kernel<", 201: "conv_hsplit_kernel<", 103: "wino_gemm_split_kernel", 301: "conv_bf16d_kernel<", 208: "conv_stem_split_kernel<"}


def binary_stamp():
    """What identifies the code a measurement was taken on: sha256 over the kernel sources + C ABI header + the host files that choose
    launches (stable across rebuilds and machines, unlike the .so's bytes) and the sha256 of the loaded libcasapose_hip.so itself.  Every
    GPU-side profiling script stores it beside its output (`python3 bench.py --stamp`); tools/make_summary.py refuses to mix stamps, and the
    bench line only quotes PMC traffic from a committed profile whose source stamp equals the running tree's."""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "casapose_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "casapose_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.join(ROOT, "casapose_amd", "csrc", "Makefile")]
                   + [os.path.join(ROOT, "casapose_amd", n) for n in ("engine.py", "train_engine.py", "_lib.py", "ops.py")] + [os.path.join(ROOT, "bench.py")])
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode() + b"\0" + open(f, "rb").read() + b"\0")
    so = os.path.join(ROOT, "casapose_amd", "libcasapose_hip.so")
    so_sha = hashlib.sha256(open(so, "rb").read()).hexdigest() if os.path.exists(so) else None
    return {"src_sha256": h.hexdigest()[:16], "so_sha256": so_sha[:16] if so_sha else None, "files": len(files)}


def _committed_profile(name):
    """newest profiles/r0N_<name> (by round number) and the stamp recorded for it in profiles/r0N_STAMP.json (None if unstamped)"""
    import glob
    import re

    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + name)), key=lambda q: int(re.search(r"r(\d+)_", os.path.basename(q)).group(1)))
    if not cands:
        return None, None
    path = cands[-1]
    tag = os.path.basename(path).split("_")[0]
    sp = os.path.join(ROOT, "profiles", tag + "_STAMP.json")
    stamp = json.load(open(sp)).get("stamp") if os.path.exists(sp) else None
    return path, stamp


def measured_traffic(tile):
    """(average HBM bytes per launch of the instantiation family `tile` from the newest committed PMC passes, note).  The figure is only
    quoted when that profile was taken on THIS source tree (stamp match); otherwise None and the note says which profile was refused."""
    path, stamp = _committed_profile("pmc_traffic.json")
    if path is None or tile not in TILE_PMC_PREFIX:
        return None, "no committed PMC profile"
    rel = os.path.relpath(path, ROOT)
    if stamp is None or stamp.get("src_sha256") != binary_stamp()["src_sha256"]:
        return None, "%s was measured on another source tree (stamp %s, this tree %s): not quoted" % (rel, (stamp or {}).get("src_sha256"), binary_stamp()["src_sha256"])
    tab = json.load(open(path))
    tot = n = 0.0
    for name, v in tab.items():
        if TILE_PMC_PREFIX[tile] in name:
            tot += v["hbm_bytes_per_launch"] * v["launches"]
            n += v["launches"]
    return (round(tot / n) if n else None), "%s (same source stamp %s as this run)" % (rel, stamp["src_sha256"])


def _sustained_mfma(dev):
    """What the matrix pipes SUSTAIN on this box under the power limit: a bare MFMA stream on every SIMD (csrc/capi.hip: cp_mfma_probe), timed
    with HIP events -- reported beside the datasheet peaks the roofline fractions are priced against, never instead of them."""
    import ctypes as C

    import torch

    from casapose_amd import _lib
    lib = _lib.load()
    ws = torch.empty(lib.cp_mfma_probe_workspace_bytes(), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = {}
    for which, name, iters in ((0, "f32_mfma_tflops", 4000), (1, "bf16_mfma_tflops", 8000)):
        fl = C.c_double(0.0)
        _lib.check(lib.cp_mfma_probe(which, iters // 10, ws.data_ptr(), C.byref(fl), stream), "cp_mfma_probe")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            _lib.check(lib.cp_mfma_probe(which, iters, ws.data_ptr(), C.byref(fl), stream), "cp_mfma_probe")
        e1.record()
        e1.synchronize()
        out[name] = round(3.0 * fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
    return out


def _dtype_note_train():
    """the training plan's defaults put two kernel groups on the bf16 matrix pipe with fp32-EQUIVALENT arithmetic (exact three-way splits, six
    products, fp32 accumulate): the Winograd GEMMs (CASAPOSE_WINO_GEMM) and the forward / data-gradient of the shallow 3x3 layers
    and the forward / data gradient / weight gradient of the plain 3x3 layers that are not on the Winograd path (CASAPOSE_CONV_MODE);
    CASAPOSE_CONV_MODE=bf16 rounds the latter's operands to bf16 (BASELINE configs[2])."""
    parts = []
    if os.environ.get("CASAPOSE_WINO_GEMM", "split") == "split":
        parts.append("Winograd GEMMs")
    conv = os.environ.get("CASAPOSE_CONV_MODE", "split")
    if conv == "split":
        parts.append("forward / data gradient / weight gradient of the 3x3 layers off the Winograd path")
    note = "f32"
    if parts:
        note += " (%s as exact 3-way bf16 splits on v_mfma_f32_32x32x16_bf16: six products, fp32 accumulate, fp32-equivalent; all other kernels fp32 MFMA)" % " and ".join(parts)
    if conv == "split" and os.environ.get("CASAPOSE_TRAIN_FWD", "f16x2") == "f16x2":
        note += ("; FORWARD launches of those groups as fp16 two-way splits on v_mfma_f32_32x32x16_f16 (three exact products per fp32 product, fp32-level error: "
                 "tests/test_gpu_f16x2.py; the operands' range watched by a device-side monitor, train_engine.TrainPlan._poll_f16x2)")
        if os.environ.get("CASAPOSE_TRAIN_BWD", "f16x2") == "f16x2":
            note += ("; BACKWARD launches of the 3x3 layers (data and weight gradients, Winograd and direct) in the same fp16 two-way split from the second "
                     "step on -- gradients brought into fp16's band by powers of two (one on the loss, one per Winograd GEMM; exact), chosen from "
                     "device-side maxima without a synchronisation; the first step and any GEMM outside the band run the exact bf16 split "
                     "(`f16x2_backward` in this line, train_engine.train_bwd_f16x2)")
        else:
            note += ", backward launches as exact bf16 splits"
    if conv == "bf16":
        note = "bf16 operands / f32 accumulate in the forward / data gradient / weight gradient of the 3x3 layers off the Winograd path; " + note + " elsewhere"
    return note


def _dtype_note():
    """The arithmetic the convolutions compute in.  Default since round 3 (conv mode "split"): fp32 values, every product formed EXACTLY on the
    bf16 matrix pipe -- each fp32 operand is split into three bf16 terms (8+8+8 significand bits), six bf16 x bf16 products (each exact in fp32)
    are accumulated in fp32; the three dropped terms are <= 2^-24 of the product, the rounding an fp32 multiply makes itself, and the measured
    error against fp64 is at or below the fp32 MFMA's (tests/test_gpu_hsplit.py, test_gpu_conv.py).  Tensors stay fp32 in HBM.  Layers the split
    kernels do not cover (strided 3x3, a few 1x1) run on v_mfma_f32_32x32x2_f32.  CASAPOSE_INFER_CONV_MODE=f32 puts every layer
    there; =bf16 rounds operands to bf16 (not fp32-equivalent)."""
    from casapose_amd import engine

    mode = os.environ.get("CASAPOSE_INFER_CONV_MODE", engine.DEFAULT_INFER_CONV_MODE)
    wino = os.environ.get("CASAPOSE_WINO_GEMM", "")
    if mode == "bf16":
        return "bf16 operands / f32 accumulate in the 3x3 layers off the Winograd path, Winograd GEMMs on hi + mid bf16 planes, f32 elsewhere (NOT fp32-equivalent)"
    if mode == "f16x2":
        return ("f32 (fp32-level: every fp32 operand as an fp16 pair hi = rn(x), lo = rn(x - hi) -- reproduced to within one fp32 ulp, 0.75 x 2^-24 rms -- and the three products hi*hi, hi*lo, lo*hi, "
                "each exact, accumulated in fp32 on v_mfma_f32_32x32x16_f16; weights pre-scaled by a power of two; measured error against fp64 at or below the fp32 "
                "MFMA's (tests/test_gpu_f16x2.py, `accuracy_vs_fp64` in this line); for the 3x3 / stride-1 layers, the 7x7 stem%s; strided 3x3 / 1x1 layers on "
                "v_mfma_f32_32x32x2_f32; tensors fp32 in HBM)" % ("" if wino == "f32" else " and the Winograd GEMMs"))
    if mode == "split":
        return ("f32 (fp32-equivalent: exact 3-way bf16 splits on v_mfma_f32_32x32x16_bf16, six exact products per fp32 product, fp32 accumulate, for the 3x3 / "
                "stride-1 layers, the 7x7 stem%s; strided 3x3 / 1x1 layers on v_mfma_f32_32x32x2_f32; tensors fp32 in HBM)" % ("" if wino == "f32" else " and the Winograd GEMMs"))
    return "f32 (v_mfma_f32_32x32x2_f32 in every convolution%s)" % ("; Winograd GEMMs as exact 3-way bf16 splits" if wino == "split" else "")


def bench_train(args):
    """Secondary workload (BASELINE.json configs[2]/[3]): one data-parallel TRAINING step of casapose_c_gcu5 --
    forward with batch statistics (SyncBN all-reduced across ranks), mask/vertex/proxy/keypoint losses, hand-written
    backward, SUM all-reduce of the flat gradient over RCCL, Adam, weight re-pack -- at 448x448, K=9, fp32."""
    import torch
    import torch.distributed as dist

    from casapose_amd import parallel

    rank, local, world = parallel.init_from_env("nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    result = train_leg(args.batch, args.height, args.width, args.steps, args.warmup, dev, rank, world)
    if rank == 0:
        print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


def train_leg(B, H, W, steps, warmup, dev, rank, world):
    """`steps` timed training steps (after `warmup`) on this rank's synthetic batch; returns the result dict of `--mode train`.  Also run by
    the DEFAULT bench line (3 steps at bs 32, 448x448, key `training_leg`) so that the driver's clock covers the training path too."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from casapose_amd import parallel
    from casapose_amd.pose_models.tfkeras import Classifiers
    from casapose_amd.train_engine import crop_to_image_affine

    class _A:
        pass

    args = _A()
    args.steps, args.warmup = steps, warmup
    seg_dim, ver_dim, kp = 9, 27, 9
    net = Classifiers.get("casapose_c_gcu5")(ver_dim=ver_dim, seg_dim=seg_dim, input_shape=(H, W, 3), input_segmentation_shape=(H, W, seg_dim),
                                             weights=None, base_model="resnet18", device=dev, seed=1237)
    group = dist.group.WORLD if dist.is_initialized() else None   # world 1 only with CASAPOSE_DIST_FORCE=1 (the RCCL path on one GPU)
    plan, _ = net.training_plan(B, H, W, group, world)
    rng = np.random.default_rng(1237 + rank)
    gen = torch.Generator(device="cpu").manual_seed(1237 + rank)
    img = (2.0 * torch.rand(B, H, W, 3, generator=gen) - 1.0).to(dev)
    # ground truth: 8 axis-aligned ellipses, one per class, keypoints inside their bounding boxes
    lab = np.zeros((B, H, W), np.uint8)
    kpts = np.zeros((B, seg_dim - 1, kp, 2), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for o in range(seg_dim - 1):
        cy, cx = H * (0.25 + 0.5 * (o // 4)), W * (0.125 + 0.25 * (o % 4))
        ry, rx = H * 0.11, W * 0.09
        lab[:, ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = o + 1
        kpts[:, o, 0] = (cy, cx)
        kpts[:, o, 1:, 0] = rng.uniform(cy - ry, cy + ry, (B, kp - 1))
        kpts[:, o, 1:, 1] = rng.uniform(cx - rx, cx + rx, (B, kp - 1))
    labd = torch.from_numpy(lab).to(dev)
    kd = torch.from_numpy(kpts).to(dev)
    offsets = np.tile(np.array([[16.0, 96.0, 0, 0, 0, 0, 0, 1, 640, 480]]), (B, 1))
    aff = torch.from_numpy(crop_to_image_affine(offsets)).to(dev)
    gt_xy = torch.from_numpy((kpts[..., ::-1] + np.array([96.0, 16.0], np.float32) + rng.normal(0, 2.0, kpts.shape)).astype(np.float32)).to(dev).contiguous()
    wts = (1.0, 0.5, 0.015)
    kp_args = dict(labels_gt=labd, gt_xy=gt_xy, affine=aff, kp_w=0.007, max_pixel_error=12.5, min_num=50, confidence_regularization=True, vote_with_gt=True)

    def step():
        return plan.train_step(img, labd, labd, kd, 1e-3, cond_labels=labd, weights=wts, filter_with_segmentation=True, kp_args=kp_args)

    plan.start_comm_log()   # one logged step: the structure of the step's exchanges (identical for every world size; python list appends only)
    for _ in range(args.warmup):
        step()
    comm_structure = plan.comm_structure()
    plan.comm_log = None
    multi = group is not None and (world > 1 or parallel.force_collectives())
    if multi:
        plan.start_comm_timing()   # events around every collective the compute stream waits for (no host synchronisation inside the timed loop)
    parallel.barrier_sync(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sums, kpl = step()
    parallel.barrier_sync(dev)
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    assert torch.isfinite(sums).all() and torch.isfinite(kpl)
    comm = plan.comm_report() if multi else None
    fwd_flops = sum(2.0 * c.desc.batch * c.desc.out_h * c.desc.out_w * c.k * c.k * sum(s[1] for s in c.sources) * c.cout for c in plan.convs)
    ex = {"f32": 0.0, "bf16": 0.0}
    for op in plan.ops:
        if hasattr(op, "executed_flops"):
            for pipe, fl in op.executed_flops().items():
                ex[pipe] += fl
    result = {
        "metric": "training images/sec at 448x448, 8-object (casapose_c_gcu5 forward + losses + backward + Adam)",
        "value": round(world * B * args.steps / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": _dtype_note_train(),
        "data": "synthetic (seed 1237: uniform images, 8 elliptical objects, he_uniform weights)",
        # N > 1 (or CASAPOSE_DIST_FORCE=1): milliseconds per step rank 0's compute stream spent inside / waiting for collectives -- the 58 blocking
        # SyncBN table all-reduces + the wait for the four gradient buckets -- and what the same buckets cost back to back on an idle GPU
        "comm_exposed_ms": comm["exposed_ms"] if comm else None, "comm": comm,
        # what a data-parallel step exchanges, counted on this run's own launches (reported at world 1 too, where nothing is sent): blocking SyncBN
        # table all-reduces and their payload, gradient buckets, and how many backward ops are launched after each bucket's asynchronous all-reduce
        "comm_structure": comm_structure,
        "config": {"workload": "config_8.ini training step: casapose_c_gcu5, K=9, ver_dim=27, bs=%d per GPU, %dx%d, fp32, GT-mask conditioning, "
                               "mask+vertex+proxy+keypoint losses, SyncBN, Adam" % (B, H, W),
                   "images_per_gpu_per_step": B, "global_batch": B * world, "parallelism": "dp%d (RCCL all-reduce of BN statistics + flat gradient)" % world},
        # EXECUTED FLOPs of the step's convolution launches by matrix pipe (Winograd layers count their grouped GEMMs, exact three-way splits six
        # bf16 products per fp32 product); `frac` = the time the two pipes would need at their peaks / the measured step time -- the step also
        # holds the normalisation, resampling, loss and optimizer passes (HBM-bound), so this is a lower bound on how busy the matrix pipes are
        "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_F32_MFMA_TFLOPS, "peak_bf16": PEAK_BF16_MFMA_TFLOPS,
                     "executed_f32_gflop_per_step": round(ex["f32"] / 1e9, 1), "executed_bf16_gflop_per_step": round(ex["bf16"] / 1e9, 1),
                     "achieved": round((ex["f32"] + ex["bf16"]) * args.steps / dt / 1e12, 3),
                     "frac": round((ex["f32"] / PEAK_F32_MFMA_TFLOPS + ex["bf16"] / PEAK_BF16_MFMA_TFLOPS) / 1e12 / (dt / args.steps), 4),
                     "frac_definition": "(f32 FLOPs / 157.3 TF + bf16 FLOPs / 2500 TF) / step time",
                     "direct_equivalent_tflops": round(3.0 * fwd_flops * args.steps / dt / 1e12, 3), "traffic": None,
                     "kernel": "all convolution launches of the step (forward, data gradient, weight gradient; conv_f32 / conv_halo / conv_hsplit / wino_gemm(_split) / conv_wgrad(_split))"},
        "losses": {"mask": float(sums[0]), "vertex": float(sums[1]), "proxy": float(sums[2]), "keypoint": float(kpl)},
        "f16x2_backward": _f16x2_backward_state(plan),
    }
    return result


def _f16x2_backward_state(plan):
    """what the backward of the timed steps ran on: the power of two on the loss, the backward GEMMs on fp16 pairs / on the exact split"""
    slots = plan._bwd_slots() if hasattr(plan, "_bwd_slots") else []
    if not slots:
        return None
    direct = [f for _, f, e in slots if e == "direct"]
    wino = [f for _, f, e in slots if e != "direct"]
    exps = [f["e"] for f in wino if f["e"] is not None]
    return {"loss_exponent": plan.loss_exp, "direct_layers_on_fp16_pairs": sum(1 for f in direct if f["on"]), "direct_layers": len(direct),
            "winograd_gemms_on_fp16_pairs": len(exps), "winograd_gemms": len(wino), "winograd_exponents": [min(exps), max(exps)] if exps else None,
            "returned_to_exact_split": list(plan.f16x2_demoted), "range_readings": plan.f16x2_checks}


def bench_vote(args):
    """Voting stage alone on the SURVEY 8(d) voting inputs (exact unit field towards random keypoints + N(0, 0.05 rad) angular noise,
    confidences N(0,1), 8 elliptical masks): the LS voter with the component filter (HBM-bound: H*W*36*4 = 44.24 MB per image read
    once) and the RANSAC voter (VALU / ballot-bound: reported as cosine tests per second)."""
    import numpy as np
    import torch

    from casapose_amd import parallel
    from casapose_amd.pose_estimation.ransac_voting import ransac_voting_layer_all_masks
    from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted

    rank, local, world = parallel.init_from_env("nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B, H, W, K, kp = args.batch, args.height, args.width, 9, 9
    g = torch.Generator(device="cpu").manual_seed(1237 + rank)
    lab = torch.zeros(B, H, W, dtype=torch.long)
    kpts = torch.zeros(B, K - 1, kp, 2)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32) + 0.5, torch.arange(W, dtype=torch.float32) + 0.5, indexing="ij")
    direct = torch.zeros(B, H, W, kp, 2)
    for o in range(K - 1):
        cy, cx = H * (0.25 + 0.5 * (o // 4)), W * (0.125 + 0.25 * (o % 4))
        ry, rx = H * 0.2, W * 0.1
        m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        lab[:, m] = o + 1
        kpts[:, o, :, 0] = cy + ry * (2 * torch.rand(B, kp, generator=g) - 1)
        kpts[:, o, :, 1] = cx + rx * (2 * torch.rand(B, kp, generator=g) - 1)
        d = kpts[:, o][:, None, :, :] - torch.stack([yy[m], xx[m]], -1)[None, :, None, :]           # [B, px, kp, 2] (dy, dx)
        ang = torch.atan2(d[..., 0], d[..., 1]) + 0.05 * torch.randn(d.shape[:-1], generator=g)
        direct[:, m] = torch.stack([torch.sin(ang), torch.cos(ang)], -1)
    seg = (10.0 * torch.nn.functional.one_hot(lab, K).float()).to(dev)
    direct = direct.reshape(B, H, W, 2 * kp).to(dev)
    conf = torch.randn(B, H, W, kp, generator=g).to(dev)
    rec = torch.cat([seg, direct, conf], 3).contiguous()   # the record layout the forward writes: [9 | 18 | 9]
    voter = CoordLSVotingWeighted(name="coords_ls_voting", num_classes=K, num_points=kp, filter_estimates=True)
    mask = torch.nn.functional.one_hot(lab, K)[..., 1:].float().to(dev)
    vertex = direct.reshape(B, H, W, kp, 2)
    draws = torch.randint(0, 2**31 - 1, (20, B, K - 1, 512, kp, 2), device=dev, dtype=torch.int32, generator=torch.Generator(device=dev).manual_seed(7))

    def ls_step():
        return voter([rec[..., :K], rec[..., K:K + 2 * kp], rec[..., K + 2 * kp:]])

    for _ in range(args.warmup):
        coords = ls_step()
    parallel.barrier_sync(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        coords = ls_step()
    parallel.barrier_sync(dev)
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    err = float((coords.cpu() - kpts).abs().max())
    # the accumulation kernel alone, HIP events on the launch stream
    from casapose_amd import ops
    labels8 = lab.to(torch.uint8).to(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.ls_vote(rec, 0, K, K + 2 * kp, K - 1, kp, labels=labels8)
    e0.record()
    for _ in range(args.steps):
        ops.ls_vote(rec, 0, K, K + 2 * kp, K - 1, kp, labels=labels8)
    e1.record()
    e1.synchronize()
    ls_us = 1e3 * e0.elapsed_time(e1) / args.steps
    alg_bytes = B * H * W * 36 * 4
    # RANSAC voter
    out, rounds = ransac_voting_layer_all_masks(mask, vertex, 512, draws=draws, return_rounds=True)
    e0.record()
    for _ in range(args.steps):
        out, rounds = ransac_voting_layer_all_masks(mask, vertex, 512, draws=draws, return_rounds=True)
    e1.record()
    e1.synchronize()
    rs_ms = e0.elapsed_time(e1) / args.steps
    tn = torch.clamp(mask.sum((1, 2)), max=30000).double()                                     # pixels per (image, object)
    tests = float((rounds.double() * 512 * kp * tn).sum())
    rerr = float((out.cpu().flip(-1) - kpts).abs().max())
    result = {
        "metric": "keypoint-voting images/sec at 640x480, 8 objects (component filter + LS voter)", "value": round(world * B * args.steps / dt, 3),
        "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (fp64 accumulation)",
        "data": "synthetic (SURVEY 8d voting inputs: exact field + N(0, 0.05 rad) noise, conf N(0,1), 8 ellipses)",
        "config": {"workload": "voting stage: bs=%d per GPU, %dx%d, 8 objects, 9 keypoints, record [9|18|9] fp32" % (B, H, W), "images_per_gpu_per_step": B,
                   "parallelism": "replicas x%d (no collective)" % world},
        "roofline": {"bound": "hbm", "achieved": round(alg_bytes / ls_us / 1e3, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(alg_bytes / ls_us / 1e3 / 8000.0, 4),
                     "traffic": None, "kernel": "ls_accumulate36_kernel (+ ls_solve_kernel)", "avg_launch_us": round(ls_us, 2),
                     "algorithmic_bytes_per_launch": alg_bytes},
        "ls_max_keypoint_error_px": round(err, 3),
        "ransac": {"ms_per_call": round(rs_ms, 3), "images_per_s": round(B / rs_ms * 1e3, 1), "cosine_tests_per_s": round(tests / rs_ms * 1e3, 1),
                   "rounds_mean": round(float(rounds.double().mean()), 2), "max_keypoint_error_px": round(rerr, 3),
                   "note": "512 hypotheses x 9 keypoints x object pixels per round; fp32 VALU + wave ballots, no MFMA",
                   # the bound (VALU / ballot): vector instructions per cosine test counted in the generated ISA of vote_kernel<64>'s
                   # hypothesis loop (common path: RANSAC_VALU_PER_HYPOTHESIS for the nine tests of one hypothesis, a wave instruction = 64 tests)
                   # x tests/s against the part's fp32 VALU issue rate (256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz = 78.6e12 lane-operations/s =
                   # the 157.3 TFLOP/s vector peak counted as FMAs); whole-call time, so hypothesis generation / refinement / compaction count against it
                   "valu": {"instructions_per_test": round(RANSAC_VALU_PER_HYPOTHESIS / 9.0, 2), "lane_ops_per_s": round(tests / rs_ms * 1e3 * RANSAC_VALU_PER_HYPOTHESIS / 9.0, 1),
                            "peak_lane_ops_per_s": VALU_LANE_OPS_PEAK, "frac": round(tests / rs_ms * 1e3 * RANSAC_VALU_PER_HYPOTHESIS / 9.0 / VALU_LANE_OPS_PEAK, 4),
                            "counted_in": "hipcc --cuda-device-only -S csrc/ransac_vote.hip: .LBB of the `#pragma unroll 1` hypothesis loop, v_* instructions on the path without borderline lanes"},
                   # direction-field bytes the votes need once per round (tn pixels x 9 keypoints x 2 floats) over the call time: far below HBM rates -- the
                   # voter is not bandwidth-bound (every block re-reads its pixel records from L2 for its 64 hypotheses)
                   "field_read_GBps_algorithmic": round(float((rounds.double() * tn * kp * 2 * 4).sum()) / rs_ms / 1e6, 2)},
    }
    if rank == 0:
        print(json.dumps(result))
    if world > 1 or parallel.force_collectives():
        import torch.distributed as dist
        dist.destroy_process_group()


def launch_ranks(args):
    """`bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): start N fresh rank processes, one per GPU,
    before anything here touches the GPU, and return their worst exit code.  Under torch.distributed.run (the driver's way) the
    ranks already exist: only check that --gpus matches WORLD_SIZE.  Returns None when this process is itself a rank."""
    import subprocess

    world = os.environ.get("WORLD_SIZE")
    if world is not None:
        if int(world) != args.gpus:
            print("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, world), file=sys.stderr)
            return 2
        return None
    if args.gpus <= 1:
        return None
    import socket

    import torch  # device_count() does not initialise the GPU runtime

    ndev = torch.cuda.device_count()
    shared = os.environ.get("CASAPOSE_DIST_BACKEND", "nccl") == "gloo"  # tests: several ranks on one GPU over gloo
    if args.gpus > ndev and not (shared and ndev > 0):
        print("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, ndev), file=sys.stderr)
        return 2
    with socket.socket() as sk:  # a free rendezvous port on the loop-back interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % ndev), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    return max(abs(p.wait()) for p in procs)


TRAIN_LEG_FAILED_EXIT = 3   # exit status when the training leg hung (watchdog) or failed with other ranks possibly waiting in a collective


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["infer", "train", "vote"], default="infer",
                    help="infer = the headline metric (default); train = one DP training step; vote = the voting stage alone (LS + RANSAC)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-worker", default=None, help="internal: run as CPU-baseline worker i/N (started by the parent before it touches the GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="timed seconds of every CPU-baseline worker process")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-bs32", action="store_true", help="skip the bs-32 roofline annotation (`roofline_bs32`: the north-star operating point beside the bs-16 headline)")
    ap.add_argument("--no-optin", action="store_true", help="skip the extra line on the fp32 MFMA (conv_mode f32) reported beside the headline")
    ap.add_argument("--train-leg-timeout", type=int, default=300, help="seconds after which a hanging training leg is abandoned (the headline line is printed without it)")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the 3-step training leg (BASELINE configs[2]) the default line reports under `training_leg`")
    ap.add_argument("--stamp", action="store_true", help="print the source / binary stamp (JSON) and exit; no GPU is touched")
    args = ap.parse_args()
    if args.stamp:
        print(json.dumps(binary_stamp()))
        return
    if args.cpu_worker:
        return cpu_worker(args)
    launched = launch_ranks(args)
    if launched is not None:
        sys.exit(launched)
    if args.mode == "train":
        if args.batch == 16 and args.height == 480 and args.width == 640:  # training defaults (config_8.ini:18, BASELINE configs[2])
            args.batch, args.height, args.width = 32, 448, 448
        return bench_train(args)
    if args.mode == "vote":
        return bench_vote(args)

    # CPU-baseline workers: started now, before anything here initialises the GPU; they wait until the GPU legs are over (rank 0 at N = 1 only)
    cpu_workers = None
    if int(os.environ.get("RANK", "0")) == 0 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        try:
            cpu_workers = spawn_cpu_workers(args)
        except Exception as exc:
            _log("cpu baseline workers not started: %s" % exc)

    import numpy as np
    import torch
    import torch.distributed as dist

    from casapose_amd import _lib
    from casapose_amd.pose_estimation.voting_layers_2d import CoordLSVotingWeighted
    from casapose_amd.pose_models.tfkeras import Classifiers

    from casapose_amd import parallel

    rank, local, world = parallel.init_from_env("nccl")  # RCCL over xGMI on ROCm
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    B, H, W = args.batch, args.height, args.width
    seg_dim, ver_dim, kp = 9, 27, 9
    net = Classifiers.get("casapose_c_gcu5")(ver_dim=ver_dim, seg_dim=seg_dim, input_shape=(H, W, 3), weights=None,
                                             base_model="resnet18", device=dev, seed=1237)
    # randomised normalisation statistics so no term of the folded affines is trivial
    rng = np.random.default_rng(1237)
    params = net.get_parameters()
    for k, v in params.items():
        if k.endswith(".gamma") or k.endswith(".moving_variance"):
            params[k] = rng.uniform(0.5, 1.5, v.shape).astype(np.float32)
        elif k.endswith(".beta") or k.endswith(".moving_mean"):
            params[k] = (0.1 * rng.standard_normal(v.shape)).astype(np.float32)
    net.set_parameters(params)
    gen = torch.Generator(device="cpu").manual_seed(1237 + rank)
    img = (2.0 * torch.rand(B, H, W, 3, generator=gen) - 1.0).to(dev)
    voter = CoordLSVotingWeighted(name="coords_ls_voting", num_classes=seg_dim, num_points=kp, filter_estimates=True)

    def step():  # the reference's own call sequence (test_casapose.py:299-312): model call, split, voting layer
        out = net([img], training=False)
        s, d, c = torch.split(out, [seg_dim, 2 * kp, kp], dim=3)
        return voter([s, d, c])

    if cpu_workers:
        cpu_workers = wait_cpu_workers(cpu_workers)   # all built and idle (blocked on stdin) before anything here is timed
        _log("%d CPU-baseline workers ready" % len(cpu_workers))
    _log("model built, %d warm-up steps" % args.warmup)
    for _ in range(args.warmup):
        step()
    _log("warm-up done")

    def barrier():
        parallel.barrier_sync(dev)

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kpts = step()
    barrier()
    dt = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt, dev)
    assert torch.isfinite(kpts).all()
    _log("timed region done: %.3f ms/step" % (1e3 * dt / args.steps))

    result = {
        "metric": "images/sec at 640x480, 8-object LMO (casapose_c_gcu5 forward + component filter + LS keypoint voting)",
        "value": round(world * B * args.steps / dt, 3),
        "unit": "images/s",
        "n_gpus": world,
        "gpus_requested": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": _dtype_note(),
        "data": "synthetic (seed 1237: uniform [-1,1) images, he_uniform weights, randomised BN/CLADE statistics)",
        "config": {"workload": "config_8.ini inference: casapose_c_gcu5, K=9 classes, ver_dim=27, bs=%d per GPU, %dx%d, fp32, estimated-mask conditioning, connected-component filter + LS voting" % (B, H, W),
                   "conv_mode": net._net.conv_mode,
                   "images_per_gpu_per_step": B, "parallelism": "replicas x%d (no collective)" % world},
    }

    def conv_roofline(net_, B=B):
        """Per-launch durations of the convolution kernels of net_'s plan (HIP events on the launch stream), grouped by kernel family and
        priced per MATRIX PIPE: a family on the bf16 pipe (exact three-way splits: six executed bf16 FLOPs per fp32 FLOP; the hi + mid
        Winograd GEMM: three) against the dense bf16 peak, a family on the fp32 MFMA against the fp32 peak.  `frac` = the time-weighted
        mean of (family rate / its pipe's peak) over ALL convolution time incl. the Winograd transform passes (which execute no FLOPs)."""
        plan = net_._net.plan(B, H, W)
        lib = _lib.load()
        stream = torch.cuda.current_stream(dev).cuda_stream
        reps = max(3, min(args.steps, 10))
        per_tile = {}

        def timed(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / reps

        wino = {"layers": 0, "ms": 0.0, "gemm_ms": 0.0, "replaced_flops": 0.0}
        direct_flops = 0.0
        for conv in plan.convs:
            pipe = getattr(conv, "split_mode", 0)   # 3 / 1: this layer runs on the bf16 matrix pipe, csrc/conv_hsplit.hip
            deep = getattr(conv, "deep_bf16", False)   # bf16 conv mode: csrc/conv_bf16d.hip
            if deep:
                pipe = 1
            if hasattr(conv, "gemm_flops") and getattr(conv, "Us", None) is not None:
                pipe = 3                                # Winograd GEMM on the bf16 pipe (exact three-way split, or hi + mid planes in the bf16 mode)
            gemm1x1 = getattr(conv, "_gemm", None)   # 1x1 / stride-1 layer on the bf16-pipe GEMM
            if gemm1x1 is not None:
                pipe = 3
            stem_split = getattr(conv, "stem_split", 0)   # conv0 on the bf16 matrix pipe, csrc/conv_stem_split.hip
            if stem_split:
                pipe = stem_split
            tile = (103 if pipe else 100) if (hasattr(conv, "gemm_flops") or gemm1x1 is not None) else (301 if deep else (208 if stem_split else (200 + pipe if pipe else lib.cp_conv_selected_tile(conv.desc))))
            d_ = conv.desc
            t = per_tile.setdefault(tile, {"ms": 0.0, "flops": 0.0, "launches": 0, "bytes": 0.0, "peak": PEAK_BF16_MFMA_TFLOPS if pipe else PEAK_F32_MFMA_TFLOPS})
            direct_flops += conv.flops
            if hasattr(conv, "gemm_flops"):
                # Winograd layer: the grouped GEMM is an MFMA kernel of its own family, accounted with the FLOPs it EXECUTES;
                # the two transform passes are streaming kernels and are reported separately
                gemm_ms = timed(lambda: conv.run_gemm(stream))
                whole_ms = timed(lambda: conv.run(stream))
                t["ms"] += gemm_ms
                t["flops"] += conv.gemm_flops * (PRODUCTS[getattr(conv, "planes", 3)] if pipe == 3 else 1.0)
                t["launches"] += 1
                t["bytes"] += 4.0 * (36.0 * conv.Tp * (conv.ktot + conv.cout) + conv.U.numel())
                wino["layers"] += 1
                wino["ms"] += whole_ms
                wino["gemm_ms"] += gemm_ms
                wino["replaced_flops"] += conv.flops
                continue
            if gemm1x1 is not None:
                t["ms"] += timed(lambda: conv.run(stream))
                t["flops"] += conv.flops * PRODUCTS[gemm1x1["planes"]]
                t["launches"] += 1
                continue
            t["ms"] += timed(lambda: conv.run(stream))
            t["flops"] += conv.flops * PRODUCTS.get(pipe, 1.0)   # the exact split executes six bf16 products per fp32 product, the fp16 split three
            t["launches"] += 1
            # algorithmic HBM bytes: every operand once (sources at their stored resolution, packed weights, outputs)
            byt = 4.0 * (conv.wp.numel() if conv.wp is not None else 0)
            for si in range(d_.num_sources):
                sc = d_.src[si]
                byt += 4.0 * d_.batch * d_.in_h * d_.in_w * sc.channels / (1 if sc.mode == 0 else 4)
            byt += 4.0 * d_.batch * d_.out_h * d_.out_w * d_.cout * ((1 if d_.out_raw else 0) + (1 if d_.out_act else 0))
            byt += 4.0 * d_.batch * d_.out_h * d_.out_w * d_.head_cout if d_.head_out else 0.0
            t["bytes"] += byt
        dom = max(per_tile, key=lambda k: per_tile[k]["ms"])
        d = per_tile[dom]
        transform_ms = wino["ms"] - wino["gemm_ms"]
        conv_ms = sum(t["ms"] for t in per_tile.values()) + transform_ms   # MFMA launches + Winograd transform passes
        pipes = {}
        for name, peak in (("f32", PEAK_F32_MFMA_TFLOPS), ("bf16", PEAK_BF16_MFMA_TFLOPS)):
            fam = [t for t in per_tile.values() if t["peak"] == peak]
            ms, fl = sum(t["ms"] for t in fam), sum(t["flops"] for t in fam)
            if ms > 0:
                pipes[name] = {"ms_per_step": round(ms, 3), "executed_gflop_per_step": round(fl / 1e9, 1), "tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                               "peak": peak, "frac_of_its_peak": round(fl / (ms * 1e-3) / 1e12 / peak, 4)}
        # time-weighted mean of each family's fraction of ITS pipe's peak over all convolution time (transform passes count as zero)
        weighted = sum(t["ms"] * (t["flops"] / (t["ms"] * 1e-3) / 1e12 / t["peak"]) for t in per_tile.values()) / conv_ms
        main_pipe = max(pipes, key=lambda k: pipes[k]["ms_per_step"])
        peak = pipes[main_pipe]["peak"]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        traffic, traffic_src = measured_traffic(dom)
        return {
            "bound": "mfma", "unit": "TFLOP/s", "peak": peak, "achieved": round(weighted * peak, 3), "frac": round(weighted, 4),
            "frac_definition": "time-weighted mean over all convolution time (incl. the Winograd transform passes, which execute no FLOPs) of each kernel "
                               "family's EXECUTED FLOP rate / the dense peak of the matrix pipe it runs on (fp32 MFMA %.1f, bf16 / fp16 MFMA %.0f TFLOP/s; an exact "
                               "three-way bf16 split executes six 2-byte FLOPs per fp32 FLOP, the fp16 two-way split three); `achieved` = frac x `peak` of the pipe that holds most of the time (%s)"
                               % (PEAK_F32_MFMA_TFLOPS, PEAK_BF16_MFMA_TFLOPS, main_pipe),
            "kernel": "all convolution launches of the forward (conv_hsplit / wino_gemm_split / conv_halo / wino_gemm / conv_f32 / conv_stem kernels + Winograd transform passes)",
            "per_pipe": pipes, "winograd_transform_ms_per_step": round(transform_ms, 3),
            "all_conv_ms_per_step": round(conv_ms, 3), "launches_per_step": len(plan.convs),
            "direct_equivalent_gflop_per_step": round(direct_flops / 1e9, 2), "direct_equivalent_tflops": round(direct_flops / (conv_ms * 1e-3) / 1e12, 2),
            # the OTHER reading of the same measurement (round-3 verdict): `frac` above prices EXECUTED FLOPs (six bf16 products per fp32 product on the
            # split layers, the Winograd GEMMs as run); `useful_*` prices the convolutions the network DEFINES (2*M*k*k*Cin*Cout each, 177 GFLOP per
            # image) over the same convolution time, against the rate at which the bf16 pipe could deliver fp32-equivalent products at best
            # (dense bf16 peak / 6 products)
            "useful_tflops": round(direct_flops / (conv_ms * 1e-3) / 1e12, 2),
            # (dense 2-byte peak / products per fp32 product of this conv mode: 6 for the exact bf16 split, 3 for the fp16 two-way split)
            "useful_frac_of_fp32_equiv_peak": round(direct_flops / (conv_ms * 1e-3) / 1e12 / (PEAK_BF16_MFMA_TFLOPS / PRODUCTS.get(net_._net.conv_planes, 6.0)), 4),
            "fp32_equiv_peak": round(PEAK_BF16_MFMA_TFLOPS / PRODUCTS.get(net_._net.conv_planes, 6.0), 1),
            "products_per_fp32_product": PRODUCTS.get(net_._net.conv_planes, 1.0),
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_unit": "bytes per launch of the dominant family, (2*FETCH_SIZE + WRITE_SIZE)*1024 from the COMMITTED profile profiles/r0N_pmc_traffic.json "
                            "(separate rocprofv3 --pmc passes of this command, tools/profile_round.sh), not measured by this run",
            "dominant_family": {"kernel": TILE_NAMES.get(dom, "conv_f32_kernel"), "achieved": round(ach, 3), "frac": round(ach / d["peak"], 4), "peak": d["peak"],
                                "launches_per_step": d["launches"], "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
                                "algorithmic_bytes_per_launch": round(d.get("bytes", 0.0) / d["launches"])},
            "families": {TILE_NAMES.get(k, str(k)).split(" ")[0] + ("" if k != 5 else "<64x64>"): {"ms": round(t["ms"], 3), "tflops": round(t["flops"] / (t["ms"] * 1e-3) / 1e12, 2), "launches": t["launches"],
                                                                                                  "peak": t["peak"]}
                         for k, t in sorted(per_tile.items())},
            "winograd": {"layers": wino["layers"], "ms_per_step": round(wino["ms"], 3), "gemm_ms": round(wino["gemm_ms"], 3),
                         "transform_ms": round(transform_ms, 3), "replaced_direct_gflop": round(wino["replaced_flops"] / 1e9, 2)},
        }

    if rank == 0 and not args.no_roofline:
        result["roofline"] = conv_roofline(net)
        try:
            sus = _sustained_mfma(dev)
            result["roofline"]["sustained_on_this_box"] = dict(sus, what="bare MFMA stream on every SIMD (cp_mfma_probe), HIP-event timed in this run; the "
                                                               "datasheet peaks above stay the denominators of `frac`")
        except Exception as exc:  # the probe is an annotation: never fail the bench line over it
            result["roofline"]["sustained_on_this_box"] = {"error": str(exc)}
    _log("roofline section done")
    accuracy = {"params": params, "image": img[:1].cpu().numpy(), "logits": {}, "vertex": {}, "labels": {}} if rank == 0 else None

    def one_image(net_, mode):   # image 0 alone: logits, vector field and the label map the device conditioned decoder 2 on
        o = net_([img[:1]], training=False)
        accuracy["logits"][mode] = o[..., :seg_dim].cpu().numpy()
        accuracy["vertex"][mode] = o[..., seg_dim:].cpu().numpy()
        accuracy["labels"][mode] = net_._net.plan(1, H, W).labels[0].cpu().numpy()

    if rank == 0:
        # the f16x2 range guard's findings on the timed plan (engine.ForwardPlan._calibrate): layers rescaled or moved to the exact split
        rep = dict(net._net.plan(B, H, W).f16x2_report)
        one_image(net, net._net.conv_mode)
        result["config"]["f16x2_guard"] = {"enabled": bool(net._net.f16x2_guard), "layers_checked": len(rep),
                                           "not_plain_f16x2": {n: "%s (max %.3g)" % (r[1], r[0]) for n, r in rep.items() if r[1] != "f16x2"}}
    if rank == 0 and world == 1 and not args.no_roofline and B < 32 and not args.no_bs32:
        # BASELINE.json's north-star target is quoted at bs 32 (">= 70 % MFMA roofline on the encoder-decoder forward at bs = 32"): the same network and
        # the same roofline accounting at bs 32, timed in THIS run (round-5 verdict, item 7) -- a few steps of forward + component filter + LS voting
        # under the host clock, then the per-launch HIP-event table.  An annotation: `value` above stays the bs-16 workload of configs[1].
        try:
            img32 = (2.0 * torch.rand(32, H, W, 3, generator=torch.Generator(device="cpu").manual_seed(4321)) - 1.0).to(dev)

            def step32():
                out = net([img32], training=False)
                s_, d_, c_ = torch.split(out, [seg_dim, 2 * kp, kp], dim=3)
                return voter([s_, d_, c_])

            for _ in range(3):
                step32()
            torch.cuda.synchronize(dev)
            t32 = time.perf_counter()
            n32 = max(3, min(args.steps, 8))
            for _ in range(n32):
                step32()
            torch.cuda.synchronize(dev)
            dt32 = time.perf_counter() - t32
            r32 = conv_roofline(net, 32)
            result["roofline_bs32"] = {
                "images_per_s": round(32 * n32 / dt32, 1), "ms_per_step": round(1e3 * dt32 / n32, 3), "steps": n32,
                "frac": r32["frac"], "useful_frac_of_fp32_equiv_peak": r32["useful_frac_of_fp32_equiv_peak"], "useful_tflops": r32["useful_tflops"],
                "all_conv_ms_per_step": r32["all_conv_ms_per_step"], "winograd_transform_ms_per_step": r32["winograd_transform_ms_per_step"],
                "dominant_family": r32["dominant_family"], "families": r32["families"],
                "what": "the forward of the headline network at bs 32 (north-star operating point), same accounting as `roofline`: frac = executed 2-byte FLOPs / "
                        "dense peak, time-weighted over all convolution time; useful_frac = the 177 GFLOP / image the network defines against peak / %g products "
                        "per fp32 product" % PRODUCTS.get(net._net.conv_planes, 1.0)}
            del img32
            _log("bs-32 roofline done: %.3f ms/step" % (1e3 * dt32 / n32))
        except Exception as exc:   # an annotation: never fail the headline over it
            result["roofline_bs32"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_optin and net._net.conv_mode in ("split", "f16x2"):
        # the same workload in the other fp32 arithmetics, measured in the same run and reported BESIDE the headline, each with its own roofline:
        # conv_mode="f32" (v_mfma_f32_32x32x2_f32 in every convolution, the round-1/2 headline) and -- when the headline is the fp16 two-way split --
        # conv_mode="split" (exact three-way bf16 splits, the round-3 / early round-4 headline).  Their logits for one image go to the CPU leg,
        # which compares every mode with an fp64 evaluation of the same network (`accuracy_vs_fp64`).
        ref_logits = net([img], training=False)[..., :seg_dim].clone()
        headline_mode = net._net.conv_mode
        net = None
        torch.cuda.empty_cache()
        for other, key, what in (("f32", "exact_fp32_mfma", "CASAPOSE_INFER_CONV_MODE=f32: every convolution on v_mfma_f32_32x32x2_f32 (the headline of rounds 1 and 2); NOT this round's headline"),
                                 ("split", "exact_bf16_split", "CASAPOSE_INFER_CONV_MODE=split: exact three-way bf16 splits, six products per fp32 product (the headline of round 3); NOT this round's headline")):
            if other == headline_mode:
                continue
            net2 = Classifiers.get("casapose_c_gcu5")(ver_dim=ver_dim, seg_dim=seg_dim, input_shape=(H, W, 3), weights=None, base_model="resnet18", device=dev,
                                                      seed=1237, conv_mode=other)
            net2.set_parameters(params)

            def step2():
                out = net2([img], training=False)
                s_, d_, c_ = torch.split(out, [seg_dim, 2 * kp, kp], dim=3)
                return voter([s_, d_, c_])

            for _ in range(args.warmup):
                step2()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step2()
            torch.cuda.synchronize(dev)
            dt2 = time.perf_counter() - t1
            one_image(net2, other)
            result[key] = {
                "value": round(B * args.steps / dt2, 3), "unit": "images/s", "ms_per_step": round(1e3 * dt2 / args.steps, 4),
                # (random-weight label maps sit on ties, so keypoints are not comparable between two runs; the logits are)
                "max_logit_difference_vs_headline_rel": float("%.3g" % float((net2([img], training=False)[..., :seg_dim] - ref_logits).abs().max() / ref_logits.abs().max())),
                "what": what,
                "roofline": conv_roofline(net2) if not args.no_roofline else None}
            _log("%s line done: %.3f ms/step" % (key, 1e3 * dt2 / args.steps))
            net2 = None
            torch.cuda.empty_cache()
        del ref_logits
    if not args.no_train_leg:
        # BASELINE configs[2] / [3] beside the headline, under the driver's clock: 3 training steps (1 warm-up) at the --mode train defaults.
        # With N > 1 ranks this is the DATA-PARALLEL step -- every rank trains on its own 32 images, SyncBN tables and the four gradient buckets
        # go through RCCL -- so the driver's 1 / 2 / 4 / 8-GPU runs carry a training scaling figure next to the replica-parallel headline.
        # It annotates the headline and must never cost it: an exception is reported in place, and a watchdog ends a leg that hangs (a rank
        # that failed while the others wait in a collective) -- rank 0 then prints the line without it.
        import threading

        net = net2 = plan = None   # noqa: F841  (drop the inference plans' buffers before the training plan allocates its own)
        torch.cuda.empty_cache()

        def give_up():
            if rank == 0:
                err = {"error": "timeout: the training legs did not finish within %d s" % args.train_leg_timeout}
                result["training_leg_bf16_convs" if "training_leg" in result else "training_leg"] = err
                result["binary"] = binary_stamp()
                print(json.dumps(result), flush=True)
            # a process that has touched the GPU and is stuck in a launch or a collective cannot be unwound: leave through _exit, and with a
            # NON-ZERO code -- the line above carries the headline and the error annotation, the exit status says that the run did not finish
            os._exit(TRAIN_LEG_FAILED_EXIT)

        dog = threading.Timer(args.train_leg_timeout, give_up)
        dog.daemon = True
        dog.start()
        keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "losses", "comm_exposed_ms", "comm", "comm_structure",
                "f16x2_backward")
        try:
            leg = train_leg(32, 448, 448, 3, 1, dev, rank, world)
            result["training_leg"] = {k: leg[k] for k in keys}
            _log("training leg done: %.2f ms/step" % leg["ms_per_step"])
            if "CASAPOSE_TRAIN_FWD" not in os.environ and "CASAPOSE_TRAIN_BWD" not in os.environ and "CASAPOSE_CONV_MODE" not in os.environ:
                # the same steps with the EXACT three-way bf16 split everywhere (the default of rounds 2-5; round 6 runs forward and backward in the
                # fp16 two-way split -- fp32-level, range-monitored)
                os.environ["CASAPOSE_TRAIN_FWD"] = os.environ["CASAPOSE_TRAIN_BWD"] = "split"
                try:
                    torch.cuda.empty_cache()
                    leg = train_leg(32, 448, 448, 3, 1, dev, rank, world)
                    result["training_leg_exact_split"] = {k: leg[k] for k in keys}
                    _log("exact-split training leg done: %.2f ms/step" % leg["ms_per_step"])
                finally:
                    del os.environ["CASAPOSE_TRAIN_FWD"], os.environ["CASAPOSE_TRAIN_BWD"]
            if "CASAPOSE_CONV_MODE" not in os.environ:
                # BASELINE configs[2] AS NAMED ("bs=32 bf16 convs"): the same three steps with the operands of the convolutions rounded to bf16
                # (fp32 accumulate; gates 3e-2 on outputs + the convergence test, tests/test_gpu_train.py) beside the fp32-equivalent default
                os.environ["CASAPOSE_CONV_MODE"] = "bf16"
                try:
                    torch.cuda.empty_cache()
                    leg = train_leg(32, 448, 448, 3, 1, dev, rank, world)
                    result["training_leg_bf16_convs"] = {k: leg[k] for k in keys}
                    _log("bf16-conv training leg done: %.2f ms/step" % leg["ms_per_step"])
                finally:
                    del os.environ["CASAPOSE_CONV_MODE"]
        except Exception as exc:  # an annotation of the headline line: report, never fail it
            result.setdefault("training_leg", {"error": "%s: %s" % (type(exc).__name__, exc)})
            if "error" not in result["training_leg"]:
                result["training_leg_bf16_convs"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            if world > 1:   # the other ranks may be waiting for this one in a collective: they leave through their own watchdogs
                dog.cancel()
                if rank == 0:
                    result["binary"] = binary_stamp()
                    print(json.dumps(result), flush=True)
                os._exit(TRAIN_LEG_FAILED_EXIT)
        dog.cancel()
    result["binary"] = binary_stamp()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(H, W, seg_dim, ver_dim, B, accuracy, workers=cpu_workers, worker_seconds=args.cpu_seconds)
    if rank == 0:
        print(json.dumps(result))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
