#!/usr/bin/env python3
"""Headline benchmark: query x gallery NCC pairs/second on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5]

`--gpus N` (N > 1) invoked plainly starts its own N worker processes (`python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>`) BEFORE this process has made any GPU call, and
exits with their code; under torchrun (RANK / WORLD_SIZE set) the process is a worker.

Workloads (`--config`, BASELINE.json `configs`; config.workload names the one that ran):
  2  (default, the one the metric is quoted on) WVU2019-shaped retrieval with VGG16 conv3_3 features: Q = 100 queries x
     G = 1500 gallery items per GPU, feature stacks [256, 128, 64] float32 (a 512x256 print through VGG16 features[:16]).
  3  ResNet50-layer3-shaped maps [1024, 32, 16] stored as bfloat16, Q = 64 x G = 10 000 per GPU (extractor bypassed).
  4  100 k gallery over 8 GPUs: Q = 64 x G = 12 500 per GPU, conv3_3 maps stored as float16, prepared gallery in chunks.
  5  as 4 with three feature layers per item (conv3_3, conv4_3, conv5_3, float16), score = mean of the three, fused on the
     device; the per-layer prepare + score chains run on separate HIP streams.
Seeded synthetic post-ReLU features generated on the device (no dataset / weights offline; queries = noisy shifted
copies of their match with the hardest mixing weights the exact integer generator allows, signal 1 / noise 126: the
true match then scores ~0.006 against ~0.004 for the runner-up; the parity leg compares FULL rank vectors, not only the true
match's rank).  One step = one pass of the hot
path from HBM-resident features to int32 ranks: prepare queries, prepare the gallery (chunked to the HBM budget), score
all pairs, [N > 1: all-gather the score blocks], rank the true matches.

N > 1: the gallery is sharded (every rank owns G items: weak scaling), queries replicated, one RCCL all-gather of the
[Q, G] float32 score blocks per step.

The JSON line also carries `roofline` for the dominant kernel (duration from HIP events on the launch stream inside the
timed region) and, at N = 1, `cpu_baseline`: the CPU oracle ("port" of the reference scorer) timed on the host cores on
a bounded sample (>= 30 s), run BEFORE this process initialises the GPU (its process pool forks), together with a parity
check of the GPU's scores and of the full rank vectors of the sampled queries against it.

`--emu` (tests only) runs the same step on the CPU emulation of the kernels with a tiny workload and the gloo backend.
"""

from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SEED = 1234
CORES_VISIBLE = None  # cores in the affinity mask, before the cgroup quota / the stated 16-core share cuts them down
SIGNAL, NOISE, MAX_SHIFT = 1, 126, 3  # the hardest queries the exact integer generator makes (signal + noise < 128):
# true-match score ~0.006 against ~0.004 for the runner-up at conv3_3 size (tools/ubench/noise_sweep.py)
PEAK_FP32_TFLOPS = 157.3            # MI355X_MICROARCH.md: FP32 vector = FP32 matrix peak
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_BF16_TFLOPS = 2500.0           # MI355X_MICROARCH.md: dense bf16 MFMA (the 2:1-sparsity figure is twice this)

LAYERS = {"conv3_3": (256, 128, 64), "conv4_3": (512, 64, 32), "conv5_3": (512, 32, 16), "resnet50_layer3": (1024, 32, 16)}
WORKLOADS = {
    2: dict(name="WVU2019-shaped VGG16 conv3_3 NCC", layers=["conv3_3"], storage="float32", q=100, g=1500),
    3: dict(name="ResNet50-layer3-shaped NCC (extractor bypassed)", layers=["resnet50_layer3"], storage="bfloat16", q=64, g=10000,
            noise=60),  # 28 x 12 cropped maps carry 1/18 of the pixels of conv3_3: the hardest setting drowns the match
    4: dict(name="100k-gallery VGG16 conv3_3 NCC, chunked prepared gallery", layers=["conv3_3"], storage="float16", q=64, g=12500),
    5: dict(name="multi-layer (conv3_3 + conv4_3 + conv5_3) fused NCC", layers=["conv3_3", "conv4_3", "conv5_3"],
            storage="float16", q=64, g=12500),
}
EMU_LAYERS = {"conv3_3": (2, 24, 16), "conv4_3": (3, 16, 12), "conv5_3": (2, 12, 10), "resnet50_layer3": (3, 12, 10)}


def fft_pair_flops(plan_fft, ih, iw, r_rows):
    """Nominal flops of ONE pair and channel in the FFT pair kernel (5 N log2 N per complex FFT):
    spectrum product, nw/2 inverse column transforms of length nh, r_rows/2 inverse row transforms of
    length nw (two real rows each), 1/sigma weighting + accumulation."""
    nh, nw = plan_fft
    prod = 6 * (nw // 2 + 1) * nh
    cols = (nw // 2) * 5 * nh * math.log2(nh)
    rows = (r_rows // 2) * 5 * nw * math.log2(nw)
    weight = 2 * ih * iw
    return prod + cols + rows + weight


def mfma_tile_steps(th, tw):
    """(issued, all) tile steps of the matrix-core kernel per channel and 16 queries: a step = 16 positions x two template rows;
    the kernel skips the steps whose fragment of the padded search map is all zeros (ncc_mfma.hip, MCfg::frag_zero)."""
    tp = 3 if tw == 12 else 2
    dy, ntile, ks_n, cy = 16 * tp // tw, th * tw // 16, th // 2, th // 2
    issued = 0
    for t in range(ntile):
        tg, ph = divmod(t, tp)
        ymin, ymax = (16 * ph) // tw, (16 * ph + 15) // tw
        for ks in range(ks_n):
            s_ = dy * tg + 2 * ks
            issued += not (s_ + ymax + 1 < cy or s_ + ymin >= cy + th)
    return issued, ntile * ks_n


def direct_pair_flops(th, tw, ih, iw):
    """Exact-overlap multiply-adds of the direct form (15.94 GFLOP/pair at conv3_3)."""
    def ov(n_t, n_i):
        c = n_t // 2
        return sum(max(0, min(n_i, y - c + n_t) - max(0, y - c)) for y in range(n_i))
    return 2 * ov(th, ih) * ov(tw, iw)


class ClockSampler:
    """Best-effort reader of the GPU's current shader clock and socket power from sysfs (plain file reads in a
    thread, no child process), sampled while the timed steps run: the pair kernel sits at the board's power
    limit, so the clock it actually gets belongs next to the roofline fraction."""

    def __init__(self, pci_bus_id=None, period=0.1):
        import glob
        import threading

        self.samples = []
        self._stop = threading.Event()
        self._thread = None
        cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        # the sysfs node of THIS process's GPU: matched by PCI address (a box may expose its neighbours' cards too)
        match = [c for c in cards if pci_bus_id and os.path.realpath(os.path.dirname(c)).lower().endswith(pci_bus_id.lower())]
        if not match:
            return
        dev = os.path.dirname(match[0])
        self._sclk = os.path.join(dev, "pp_dpm_sclk")
        power = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*", "power1_average")) +
                       glob.glob(os.path.join(dev, "hwmon", "hwmon*", "power1_input")))
        self._power = power[0] if power else None
        self._period = period
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        mhz = None
        for line in open(self._sclk).read().splitlines():
            if line.strip().endswith("*"):
                mhz = float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
        watts = float(open(self._power).read()) / 1e6 if self._power else None
        return mhz, watts

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append(self._read())
            except (OSError, ValueError, IndexError):
                return
            self._stop.wait(self._period)

    def start(self):
        if self._thread:
            self._thread.start()

    def stop(self):
        self._stop.set()
        if self._thread:
            self._thread.join(timeout=2.0)
        clocks = sorted(m for m, _ in self.samples if m)
        watts = sorted(w for _, w in self.samples if w)
        if not clocks:
            return None
        out = {"sclk_mhz_median": clocks[len(clocks) // 2], "samples": len(clocks)}
        if watts:
            out["socket_power_w_median"] = round(watts[len(watts) // 2], 1)
        return out


def usable_cores():
    """Host cores this job may actually use: the scheduler affinity, cut down to the cgroup's CPU quota where one is set
    (a GPU box hands a one-GPU job a share of the host: `nproc` and the affinity mask still show every core)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    how = "affinity"
    global CORES_VISIBLE
    CORES_VISIBLE = n
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = max(1, int(math.ceil(float(quota) / period)))
                if q < n:
                    n, how = q, "cgroup quota"
            break
        except (OSError, ValueError, IndexError):
            continue
    if "SPR_CPU_CORES" in os.environ:
        n, how = int(os.environ["SPR_CPU_CORES"]), "SPR_CPU_CORES"
    elif n > 16 and how == "affinity":
        n, how = 16, "capped at the GPU box's stated 16-core share per GPU (no cgroup quota visible)"
    return n, how


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def storage_round(a: np.ndarray, storage: str) -> np.ndarray:
    """float32 values as they read back from their HBM storage type (the oracle then sees what the kernels see)."""
    from shoeprint_image_retrieval_amd import synth

    if storage == "float16":
        return a.astype(np.float16).astype(np.float32)
    if storage == "bfloat16":
        return synth.from_bfloat16_bits(synth.bfloat16_bits(a))
    return a


def cpu_leg(layer_shape, storage, nq_total, ng_total, target_seconds, cores, sample_q, sample_g, more_layers=()):
    """The CPU oracle's compare_maps (process pool over query chunks, scipy FFT per channel: the reference's
    formulation) on a bounded sample of the SAME workload: queries 0 .. sq-1 against a gallery subset that holds their
    true matches.  Returns pairs/s and the oracle's float32 score block + ranks for the parity check.  Runs before the
    GPU is initialised in this process: the pool forks.  `more_layers` (config 5): the further feature layers of the sampled
    block are scored too, outside the timed region, and the mean over all layers is returned for the parity check."""
    from oracle import ncc_oracle as oracle  # baseline / checker leg only
    from shoeprint_image_retrieval_amd import synth

    c, h, w = layer_shape
    matches = synth.default_matches(nq_total, ng_total)
    sq = sample_q or max(8, min(nq_total, cores))
    cfg = {"comparison": {"n_processes": cores, "rotations": None, "scales": None}}

    def features(q_ids, g_ids, c=c, h=h, w=w):
        gal = [storage_round(synth.gallery_features(SEED, g, c, h, w), storage) for g in g_ids]
        qs = [storage_round(synth.query_features(SEED, q, int(matches[q]), c, h, w, max_shift=MAX_SHIFT, signal=SIGNAL,
                                                 noise=NOISE), storage) for q in q_ids]
        return qs, gal

    sg = sample_g
    if not sg:
        # size the sample from a calibration pass on these very cores (eight pairs per process), not from an assumed rate;
        # the short pass still pays more overhead per pair than the long one (it came out 30 % short): allow for that
        qs, gal = features(range(min(nq_total, cores)), list(range(8)))
        oracle.compare_maps(qs[:1], gal[:1], [0], cfg)  # (pool start-up and imports paid once before the clock runs)
        t0 = time.perf_counter()
        oracle.compare_maps(qs, gal, [0] * len(qs), cfg)
        rate = len(qs) * 8 / (time.perf_counter() - t0)
        sg = int(min(ng_total, max(sq, math.ceil(1.3 * target_seconds * rate / sq))))
        print(f"[bench] CPU calibration: {rate:.1f} pairs/s on {cores} processes -> sample {sq} x {sg}", file=sys.stderr, flush=True)
    g_ids = sorted({int(matches[q]) for q in range(sq)})
    g_ids += [g for g in range(ng_total) if g not in set(g_ids)][: max(0, sg - len(g_ids))]
    g_ids = sorted(g_ids)
    queries, gallery = features(range(sq), g_ids)
    local_match = [g_ids.index(int(matches[q])) for q in range(sq)]
    t0 = time.perf_counter()
    ranks, matrix = oracle.compare_maps(queries, gallery, local_match, cfg, return_matrix=True)
    dt = time.perf_counter() - t0
    matrix = np.asarray(matrix, dtype=np.float32)
    if more_layers:  # the fused score of the multi-layer workload = mean of the per-layer float32 matrices (as on the device)
        total = matrix.astype(np.float64)
        for shape in more_layers:
            qs, gal = features(range(sq), g_ids, *shape)
            total += np.asarray(oracle.compare_maps(qs, gal, local_match, cfg, return_matrix=True)[1], dtype=np.float32)
        matrix = (total / (1 + len(more_layers))).astype(np.float32)
        ranks = [oracle.rank_true_match(matrix[i], local_match[i]) for i in range(sq)]
    return {"pairs_per_s": sq * len(g_ids) / dt, "seconds": dt, "sq": sq, "g_ids": g_ids, "local_match": local_match,
            "ranks": np.asarray(ranks), "matrix": matrix, "layers": 1 + len(more_layers)}


def self_launch(argv, gpus):
    """Plain `bench.py --gpus N`: start the N workers as fresh children (this process has not touched the GPU and never
    will) and exit with their code."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def lib_sha16(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for block in iter(lambda: fh.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()[:16]


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS))
    ap.add_argument("--queries", type=int, default=0)
    ap.add_argument("--gallery-per-gpu", type=int, default=0)
    ap.add_argument("--method", default="auto", choices=["auto", "fft", "direct"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-sample", action="store_true", help="(kept for old command lines: the parity check rides on the CPU leg)")
    ap.add_argument("--noise", type=int, default=None, help="noise amplitude of the synthetic queries (default: the workload's)")
    ap.add_argument("--no-extractor", action="store_true")
    ap.add_argument("--max-prepared-gib", type=float, default=0.0, help="HBM budget of one prepared gallery chunk (default: automatic)")
    ap.add_argument("--no-secondary", action="store_true", help="do not append the config-3 measurement to the default run")
    ap.add_argument("--cpu-seconds", type=float, default=45.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--cpu-sample-queries", type=int, default=0)
    ap.add_argument("--cpu-sample-gallery", type=int, default=0)
    ap.add_argument("--emu", action="store_true", help="tests: CPU emulation of the kernels, tiny workload, gloo")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "RANK" not in os.environ:
        return self_launch(argv, args.gpus)

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    rank_env = int(os.environ.get("RANK", "0"))
    # The default one-GPU run of the headline workload also measures BASELINE config 3 (the bf16 small-map workload of the
    # matrix-core scorer) and appends it as `secondary`: the same step, its own roofline / cpu_baseline / parity_sample.
    configs = [args.config]
    if (args.config == 2 and world_env == 1 and not args.emu and not args.no_secondary and not args.queries
            and not args.gallery_per_gpu and args.method == "auto"):
        configs.append(3)

    # ---- CPU baseline legs first: before torch touches the GPU (fork-safe), rank 0 of a one-GPU job only ---------
    cores, cores_how = usable_cores()
    cpu_legs = {}
    if world_env == 1 and rank_env == 0 and not args.no_cpu_baseline and not args.emu:
        for cfg in configs:
            set_noise(args, cfg)
            wl, layer_shapes, nq, ng_local = workload_sizes(args, cfg, cfg == args.config)
            seconds = args.cpu_seconds if cfg == args.config else min(args.cpu_seconds, 20.0)
            cpu_legs[cfg] = cpu_leg(layer_shapes[0], wl["storage"], nq, ng_local * world_env, seconds, cores,
                                    args.cpu_sample_queries if cfg == args.config else 0,
                                    args.cpu_sample_gallery if cfg == args.config else 0, more_layers=layer_shapes[1:])

    import torch

    from shoeprint_image_retrieval_amd import distributed as sdist
    from shoeprint_image_retrieval_amd.similarity import NccScorer

    rank, world, local = sdist.init_from_env("gloo" if args.emu else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.emu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from emu_util import emu_scorer

        scorer = emu_scorer(args.method)
    else:
        torch.cuda.set_device(local)
        scorer = NccScorer(method=args.method,
                           max_prepared_bytes=int(args.max_prepared_gib * (1 << 30)) if args.max_prepared_gib > 0 else None)
    ctx = dict(rank=rank, world=world, local=local, scorer=scorer, cores=cores, cores_how=cores_how)
    outs = []
    for cfg in configs:
        set_noise(args, cfg)
        outs.append(run_config(args, cfg, ctx, cpu_legs.get(cfg), cfg == args.config))
        if not args.emu:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    out = outs[0]
    if len(outs) > 1:
        out["secondary"] = outs[1:]
    if rank == 0:
        print(json.dumps(out))
    sys.stdout.flush()
    if world > 1:  # leave the group together (rank 0 ran the extra legs above) and tear the backend down cleanly
        import torch.distributed as dist

        sdist.barrier()
        dist.destroy_process_group()
    return 0


def set_noise(args, config_id):
    global NOISE
    NOISE = args.noise if args.noise is not None else WORKLOADS[config_id].get("noise", 126)


def workload_sizes(args, config_id, primary):
    """(workload, layer shapes, queries, gallery items per GPU); --queries / --gallery-per-gpu resize the primary workload only."""
    wl = WORKLOADS[config_id]
    layer_shapes = [(EMU_LAYERS if args.emu else LAYERS)[name] for name in wl["layers"]]
    nq = (args.queries if primary else 0) or (4 if args.emu else wl["q"])
    ng_local = (args.gallery_per_gpu if primary else 0) or (6 if args.emu else wl["g"])
    return wl, layer_shapes, nq, ng_local


def run_config(args, config_id, ctx, cpu, primary):
    """One workload: inputs into HBM, warm-up, the timed steps, and its JSON record (rank 0's is the one printed)."""
    import torch

    from shoeprint_image_retrieval_amd import distributed as sdist
    from shoeprint_image_retrieval_amd import parse_results, synth

    rank, world, local, scorer, cores, cores_how = (ctx[k] for k in ("rank", "world", "local", "scorer", "cores", "cores_how"))
    wl, layer_shapes, nq, ng_local = workload_sizes(args, config_id, primary)
    storage = wl["storage"]
    ng_total = ng_local * world
    dev, lib = scorer.dev, scorer.lib
    g0 = rank * ng_local

    def as_torch(x):
        return torch.from_numpy(x) if isinstance(x, np.ndarray) else x

    def as_dev(x):
        return x.numpy() if args.emu and isinstance(x, torch.Tensor) else x

    # ---- inputs resident in HBM before the timed region -----------------------------------------
    matches = synth.default_matches(nq, ng_total)
    match_dev = dev.to_device(matches)

    def synth_layer(c, h, w):
        """(queries, gallery) of one feature layer in their storage type; generated in float32 slabs."""
        q32 = dev.empty((nq, c, h, w), np.float32)
        lib.check(lib.spr_synth_queries(dev.ptr(q32), 0, nq, dev.ptr(match_dev), c, h, w, SEED, MAX_SHIFT, SIGNAL, NOISE,
                                        dev.stream()))
        q = dev.astype_storage(q32, storage)
        if storage == "float32":
            g = dev.empty((ng_local, c, h, w), np.float32)
            lib.check(lib.spr_synth_gallery(dev.ptr(g), g0, ng_local, c, h, w, SEED, dev.stream()))
            return q, g
        slab = 256
        parts = []
        tmp = dev.empty((min(slab, ng_local), c, h, w), np.float32)
        g = None
        for s0 in range(0, ng_local, slab):
            n = min(slab, ng_local - s0)
            lib.check(lib.spr_synth_gallery(dev.ptr(tmp), g0 + s0, n, c, h, w, SEED, dev.stream()))
            part = dev.astype_storage(dev.narrow0(tmp, 0, n), storage)
            if args.emu:
                parts.append(np.array(part, copy=True))
            else:
                if g is None:
                    g = torch.empty((ng_local, c, h, w), dtype=part.dtype, device=part.device)
                g[s0:s0 + n].copy_(part)
        return q, (np.concatenate(parts) if args.emu else g)

    layers = [synth_layer(*shape) for shape in layer_shapes]
    plans = [scorer.plan(c, (h, w), (h, w), dtype=storage) for (c, h, w) in layer_shapes]
    budget = scorer._budget() // max(1, len(layers))
    chunks = [int(max(1, min(ng_local, budget // max(1, p.gallery_item_bytes), 65535))) for p in plans]
    pgs = [dev.empty_bytes(p.gallery_item_bytes * ch) for p, ch in zip(plans, chunks)]
    layer_scores = [dev.zeros((nq, ng_local), np.float32) for _ in layers]
    scores = layer_scores[0] if len(layers) == 1 else dev.zeros((nq, ng_local), np.float32)
    pair_events, gather_events, gather_ms = [], [], []
    # config 5: one HIP stream per feature layer (prepare + score chains overlap), joined by events before the fusion
    streams = [torch.cuda.Stream() for _ in layers] if (len(layers) > 1 and not args.emu) else None

    def score_layer(k, record):
        (q, g), plan, chunk, pg, out = layers[k], plans[k], chunks[k], pgs[k], layer_scores[k]
        pq = scorer.prepare_queries(plan, q)
        for start in range(0, ng_local, chunk):
            n = min(chunk, ng_local - start)
            scorer.prepare_gallery(plan, dev.narrow0(g, start, n), out=pg)
            if record and k == 0:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            scorer.score_prepared(plan, pq, nq, pg, n, out, ng_local, start)
            if record and k == 0:
                e1.record()
                pair_events.append((e0, e1, nq * n))

    def step(record):
        if streams:
            main_stream = torch.cuda.current_stream()
            for k, s in enumerate(streams):
                s.wait_stream(main_stream)
                with torch.cuda.stream(s):
                    score_layer(k, record)
            for s in streams:
                main_stream.wait_stream(s)
        else:
            for k in range(len(layers)):
                score_layer(k, record and not args.emu)
        if len(layers) > 1:
            w = 1.0 / len(layers)
            for k, ls in enumerate(layer_scores):
                lib.check(lib.spr_scores_fuse(dev.ptr(scores), dev.ptr(ls), nq * ng_local, 0.0 if k == 0 else 1.0, w,
                                              dev.stream()))
        if world > 1 and record:
            if args.emu:
                tg = time.perf_counter()
                full = as_dev(sdist.gather_score_blocks(as_torch(scores), ng_total))
                gather_ms.append((time.perf_counter() - tg) * 1e3)
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                full = sdist.gather_score_blocks(scores, ng_total)
                e1.record()
                gather_events.append((e0, e1))
        else:
            full = as_dev(sdist.gather_score_blocks(as_torch(scores), ng_total))
        return scorer.ranks_device(full, match_dev), full

    def sync():
        if not args.emu:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    sampler = None
    if rank == 0 and not args.emu:
        try:
            pr = torch.cuda.get_device_properties(local)
            sampler = ClockSampler(f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0")
        except (AttributeError, OSError, ValueError):
            sampler = None
    sdist.barrier()
    sync()
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ranks_dev, full = step(True)
    sync()
    sdist.barrier()
    dt = sdist.max_over_ranks(time.perf_counter() - t0, device=None if args.emu else layers[0][1].device)
    observed = sampler.stop() if sampler else None

    pairs_per_step = nq * ng_total
    value = pairs_per_step * args.steps / dt
    ranks = dev.to_host(ranks_dev)

    # ---- N > 1: what the collective saw, and proof that every rank ended up with the same [Q, G] matrix ----------------------
    collective = None
    if world > 1:
        import torch.distributed as dist

        gather_ms += [a.elapsed_time(b) for a, b in gather_events]
        digest = sdist.matrix_digest(as_torch(full))
        digests = sdist.gather_digests(digest, device=None if args.emu else layers[0][1].device)
        if len(set(digests)) != 1:
            raise SystemExit(f"rank {rank}: the gathered score matrices differ between ranks: {digests}")
        collective = {"backend": dist.get_backend(), "world_seen": dist.get_world_size(), "ranks_reporting": len(digests),
                      "allgather_ms": round(sum(gather_ms) / max(1, len(gather_ms)), 4), "allgathers_timed": len(gather_ms),
                      "payload_bytes": nq * ng_local * 4, "gathered_bytes": nq * ng_total * 4,
                      "matrix_digest": f"{digests[0] & 0xFFFFFFFFFFFFFFFF:016x}", "digest_agrees_on_all_ranks": True,
                      "timer": "perf_counter around the gloo call (emulation)" if args.emu else
                               "HIP events on the launch stream around all_gather_into_tensor + the block re-layout"}

    # ---- roofline of the dominant kernel (the first layer's pair kernel), from the events recorded above ----------
    c0, h0, w0 = layer_shapes[0]
    plan0 = plans[0]
    ih, iw = h0 - 4, w0 - 4
    roofline = None
    if pair_events:
        pair_ms = sum(a.elapsed_time(b) for a, b, _ in pair_events)
        pair_pairs = sum(p for _, _, p in pair_events)
        launches = len(pair_events)
        if plan0.method == 4:
            # matrix-core direct form (ncc_mfma.hip): priced against the dense bf16 MFMA peak on the ALGORITHMIC flops of the
            # sliding-window form as SURVEY §8d counts them - multiply-adds on in-map pixels only, 56 % of the dense taps x
            # positions product at this size; the kernel runs the dense product and issues 1.33x it (template rows padded 12 -> 16
            # taps; 2.67x in the hi + lo form): all of that counts as loss in `frac`; the time is that of
            # the score call: pair kernels and, in the exact form, the f32 correction-matrix kernels between them
            taps = ih * iw
            flops_pair = direct_pair_flops(ih, iw, ih, iw) * c0  # multiply-adds on in-map pixels only
            dense = 2.0 * taps * taps * c0                       # the [queries x taps] x [taps x positions] product, zero taps included
            achieved = flops_pair * pair_pairs / (pair_ms * 1e-3) / 1e12
            exact = plans[0].gallery_item_bytes > c0 * 4032 + 4096  # the exact form's prepared items carry a V matrix
            steps, all_steps = mfma_tile_steps(ih, iw)
            issued = 2.0 * (ih * 16) * taps * (1 if exact else 2) * c0 * steps / all_steps  # all-zero fragments are skipped
            bytes_pair = c0 * h0 * w0 * 2
            hbm = bytes_pair * pair_pairs / (pair_ms * 1e-3) / 1e9
            roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None, "kernel": "pair_mfma_kernel" + (" + corr_mfma_kernel (exact form)" if exact else " (hi + lo form)"),
                        "launches": launches, "avg_launch_ms": round(pair_ms / launches, 3),
                        "algorithmic_gflop_per_pair": round(flops_pair / 1e9, 4),
                        "dense_product_gflop_per_pair": round(dense / 1e9, 4),
                        "dense_product_frac_of_peak": round(dense * pair_pairs / (pair_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                        "issued_mfma_gflop_per_pair": round(issued / 1e9, 4),
                        "issued_frac_of_peak": round(issued * pair_pairs / (pair_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                        "kernel_time_share": round(pair_ms * 1e-3 / dt, 3),
                        "hbm": {"algorithmic_bytes_per_pair": bytes_pair, "achieved_gbs": round(hbm, 1), "peak_gbs": PEAK_HBM_GBS,
                                "frac": round(hbm / PEAK_HBM_GBS, 4),
                                "note": "SURVEY 8d prices an UNBATCHED pair at one bf16 gallery tensor; 64 queries share every "
                                        "gallery read here, so HBM is not what bounds this kernel"},
                        "note": "v_mfma_f32_16x16x32_bf16, exact bf16 products, f32 accumulation; peak = dense bf16 2.5 PFLOP/s"}
            if observed:
                roofline["observed"] = dict(observed, nominal_sclk_mhz=2400)
        elif config_id == 3:
            # small maps: the kernel streams prepared spectra from L2 / HBM; SURVEY §8d prices it against HBM with the
            # compulsory bytes of an unbatched pair = one gallery feature tensor (bf16: 1.05 MB)
            bytes_pair = c0 * h0 * w0 * 2
            ach = bytes_pair * pair_pairs / (pair_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None, "kernel": "pair_fft_kernel (one wave per pair)",
                        "launches": launches, "avg_launch_ms": round(pair_ms / launches, 3),
                        "algorithmic_bytes_per_pair": bytes_pair,
                        "note": "SURVEY 8d: one bf16 gallery tensor per pair (queries not batched)"}
        else:
            if plan0.method == 1:
                r_rows = 16 * math.ceil(ih / 16) if plan0.fft_size[0] == 256 else ih
                flops_pair = fft_pair_flops(plan0.fft_size, ih, iw, r_rows) * c0
                kernel = "pair6_kernel" if tuple(plan0.fft_size) == (192, 96) and os.environ.get("SPR_NCC_SIX", "1") != "0" \
                    else "pair_fft_kernel"
            else:
                flops_pair = direct_pair_flops(ih, iw, ih, iw) * c0
                kernel = "pair_direct_kernel"
            achieved = (flops_pair * pair_pairs) / (pair_ms * 1e-3) / 1e12 if pair_ms > 0 else 0.0
            roofline = {
                "bound": "valu_fp32", "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "traffic": None,
                "kernel": kernel, "launches": launches, "avg_launch_ms": round(pair_ms / max(1, launches), 3),
                "algorithmic_gflop_per_pair": round(flops_pair / 1e9, 4),
                "direct_form_gflop_per_pair": round(direct_pair_flops(ih, iw, ih, iw) * c0 / 1e9, 3),
                "kernel_time_share": round(pair_ms * 1e-3 / dt, 3),
                "note": "fp32 FFT butterflies on the vector ALU (no MFMA instructions); peak = FP32 vector peak 157.3 TFLOP/s; "
                        "achieved = nominal 5 N log2 N flops of the FFT form / kernel time",
            }
            if observed:
                # what the chip actually ran at during the timed steps (power-limited: see DESIGN.md §5)
                roofline["observed"] = dict(observed, nominal_sclk_mhz=2400, frac_at_observed_clock=round(
                    achieved / (PEAK_FP32_TFLOPS * observed["sclk_mhz_median"] / 2400.0), 4))
            # HBM-side traffic per launch: not measurable from inside the run (PMC passes need rocprofv3); the committed
            # figure of the counter passes of THIS command is used only while it is keyed to the very library build that is
            # running (sha256 of the .so), otherwise null
            import glob

            sha = lib_sha16(lib.path)
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprof_summary.json")), reverse=True):
                try:
                    key = json.load(open(path)).get("pair kernel traffic per launch", {})
                    if (config_id, nq, ng_local, world) == (2, wl["q"], wl["g"], 1) and key.get("lib_sha16") == sha:
                        roofline["traffic"] = key["hbm_bytes_per_launch"]
                        roofline["traffic_unit"] = "bytes per launch (memory-side L2 requests incl. Infinity-Cache hits)"
                        roofline["traffic_source"] = (f"NOT measured in this run: the committed rocprofv3 --pmc passes of this "
                                                      f"command ({os.path.relpath(path, ROOT)}), used because the running "
                                                      f"library's sha256 {sha} is the build those passes profiled")
                        break
                except (OSError, KeyError, ValueError):
                    continue
            if roofline["traffic"] is None:
                roofline["traffic_source"] = ("none: no committed counter pass is keyed to this library build "
                                              f"(sha256 {sha}); PMC counters cannot be read from inside the run")

    shapes = " + ".join(f"[{c},{h},{w}]" for c, h, w in layer_shapes)
    out = {
        "metric": "query x gallery NCC pairs/sec", "value": round(value, 1), "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if plan0.method == 4 else "f32", "data": "synthetic",
        "config": {"workload": f"config {config_id}: {wl['name']}: Q={nq} x G={ng_total} ({ng_local}/GPU), features {shapes} "
                               f"stored as {storage}, rotations/scales none, queries signal {SIGNAL} / noise {NOISE}",
                   "method": {1: "fft", 2: "direct", 4: "mfma"}[plan0.method], "fft_grid": list(plan0.fft_size),
                   "gallery_chunk": chunks[0], "gallery_chunks_per_step": math.ceil(ng_local / chunks[0]),
                   "storage": storage, "parallelism": f"gallery-shard x{world}",
                   "streams": len(streams) if streams else 1},
        "rank1": round(parse_results.rank1(ranks), 4), "mAP": round(parse_results.mean_average_precision(ranks), 4),
        "mean_rank": round(float(np.mean(ranks)), 2),
    }
    if roofline:
        out["roofline"] = roofline
    if collective:
        out["collective"] = collective
    if args.emu:
        out["data"] = "synthetic (CPU emulation of the kernels: test mode, not a measurement)"

    if rank == 0 and not args.no_extractor and not args.emu and config_id in (2, 3):
        # the extractor, reported separately: images/s on 512x256 prints with seeded synthetic weights, in the
        # reference's float32 arithmetic (the f32 matrix cores) and on the 16-bit matrix cores (bfloat16 operands, f32 sums)
        from shoeprint_image_retrieval_amd import network

        mtype, block, gflop, what = (("VGG16", 16, 48.77, "VGG16 features[:16] images/s (512x256 uint8 -> [256,128,64] f32)")
                                     if config_id == 2 else
                                     ("ResNet50", 7, 17.13, "ResNet50 layer3 images/s (512x256 uint8 -> [1024,32,16] f32)"))
        imgs = torch.randint(0, 256, (32, 512, 256), dtype=torch.uint8, device=layers[0][1].device)

        def time_model(compute):
            cfg = {"model": {"type": mtype, "clahe_clip_limit": 2.0, "clahe_tile_grid_size": [8, 8]},
                   "mi355x": {"extractor_dtype": compute}}
            model = network.Model(cfg, block)
            model.extract_device(imgs)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                model.extract_device(imgs)
            e1.record()
            torch.cuda.synchronize()
            ems = e0.elapsed_time(e1) / 5
            model.close()
            return ems

        ems = time_model("float32")
        out["extractor"] = {"metric": what, "value": round(32 / ems * 1e3, 1), "batch": 32, "ms_per_batch": round(ems, 2),
                            "achieved_tflops": round(gflop * 32 / ems, 2), "peak_tflops": PEAK_FP32_TFLOPS,
                            "frac": round(gflop * 32 / ems / PEAK_FP32_TFLOPS, 4), "bound": "mfma", "dtype": "f32",
                            "weights": "seeded synthetic"}
        try:
            ems16 = time_model("bfloat16")
            out["extractor"]["reduced_precision"] = {
                "dtype": "bf16", "value": round(32 / ems16 * 1e3, 1), "ms_per_batch": round(ems16, 2),
                "achieved_tflops": round(gflop * 32 / ems16, 2), "peak_tflops": PEAK_BF16_TFLOPS,
                "frac": round(gflop * 32 / ems16 / PEAK_BF16_TFLOPS, 4), "speedup_over_f32": round(ems / ems16, 2),
                "note": "bfloat16 operands (weights, activations between layers), f32 accumulation, f32 output; "
                        "[mi355x].extractor_dtype = \"bfloat16\""}
        except (ValueError, NotImplementedError) as e:  # a backbone without a 16-bit path says so
            out["extractor"]["reduced_precision"] = {"dtype": "bf16", "value": None, "note": str(e)}
        del imgs

    if cpu is not None:
        v = cpu["pairs_per_s"]
        out["cpu_baseline"] = {"value": round(v, 2), "unit": "pairs/s", "cores": cores, "kind": "port",
                               "cores_from": cores_how, "cores_visible": CORES_VISIBLE, "cpu_model": cpu_model(),
                               "sample": f"queries 0..{cpu['sq'] - 1} x {len(cpu['g_ids'])} gallery items (their matches included) of the "
                                         f"same workload, first layer, oracle compare_maps with a {cores}-process pool, "
                                         f"{cpu['seconds']:.1f} s",
                               "per_core": round(v / cores, 2), "gpu_over_cpu": round(value / v, 1)}
        if True:
            # parity on the ACTUAL workload (multi-layer: the fused matrix against the mean of the oracle's per-layer ones): the GPU's scores of the sampled block against the oracle's, and the full
            # rank vector of every sampled query (ranked within the sampled gallery, by the oracle's rule, from GPU scores)
            full_h = dev.to_host(full)
            block = full_h[np.ix_(range(cpu["sq"]), cpu["g_ids"])]
            err = float(np.abs(block - cpu["matrix"]).max())
            from oracle import ncc_oracle as oracle

            gpu_ranks = np.array([oracle.rank_true_match(block[i], cpu["local_match"][i]) for i in range(cpu["sq"])])
            order_equal, swaps, gap = True, 0, 0.0
            for i in range(cpu["sq"]):
                og, oo = np.argsort(-block[i], kind="stable"), np.argsort(-cpu["matrix"][i], kind="stable")
                diff = np.nonzero(og != oo)[0]
                if diff.size:  # places where the two orders name different items, and how far apart the ORACLE scores those two
                    order_equal = False
                    swaps += int(diff.size)
                    gap = max(gap, float(np.abs(cpu["matrix"][i][og[diff]] - cpu["matrix"][i][oo[diff]]).max()))
            out["parity_sample"] = {"pairs": int(block.size), "queries": int(cpu["sq"]),
                                    "max_abs_err_vs_oracle": float(f"{err:.3e}"), "tolerance": 1e-4,
                                    "true_match_ranks_equal": bool(np.array_equal(gpu_ranks, cpu["ranks"])),
                                    "full_rank_vectors_equal": bool(order_equal),
                                    "rank_vector_places_that_differ": swaps,
                                    "largest_oracle_score_gap_at_such_a_place": float(f"{gap:.3e}"),
                                    "oracle_ranks": [int(r) for r in cpu["ranks"]], "layers": cpu["layers"]}
    return out


if __name__ == "__main__":
    sys.exit(main())
