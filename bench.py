#!/usr/bin/env python3
"""Benchmark of the COMBO-AVS fusion + mask-decoding hot path on MI355X (contract: see the task description).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): COMBO-R50 S4, bs = 8 clips x 5 frames x 224x224 per GPU (weak scaling),
synthetic inputs resident in HBM, random-init weights, one FULL training step = dual-R50 + VGGish forward, SEM mix,
pixel decoder (HIP MSDeformAttn), bilateral fusion, masked decoder, 39-term loss with Hungarian matching, backward,
one RCCL gradient all-reduce, grad-norm clip + AdamW.  Metric: train frames/s (whole job).
Forward + loss + backward are replayed from ONE captured hipGraph (trainer.GraphedTrainStep; `--no-graph` = eager
launches; a failed capture falls back to eager and says so on stderr).  The first warm-up step is eager (MIOpen's
exhaustive find for the backbone convolutions, ~2 min on a box with an empty MIOpen user db; COMBO_MIOPEN_BENCHMARK=0
skips it), the capture happens in the second.
Extra objects: `roofline` for the MSDeformAttn forward core (HBM-bound; duration by HIP events around its launches in two
EAGER runs of the same step right after the timed region - HIP cannot record events inside a captured graph on ROCm 7),
`other_kernels` (the 3xbf16 GEMM kernels against the bf16 MFMA peak, the MSDeformAttn backward against HBM, same event
pass) and `cpu_baseline` (the CPU oracle's full training step on the host cores, bounded sample, rank 0 at N=1 only).
Other modes (not the BASELINE metric): `--mode infer` (eval forward + fused inference tail), `--backbone pvt`
(COMBO-PVTv2-B5), `--grad-comm bf16` (bf16 gradient all-reduce), `--dtype fp32`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def synth_batch(n_clips, T, H, W, device, seed, K=2):
    """BASELINE.md §3 synthetic inputs: uint8 frames / Maskiges, N(0,1) log-mels, one blob GT on frame 0 (S4)."""
    g = torch.Generator().manual_seed(seed)
    batch = []
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    for _ in range(n_clips):
        images = torch.randint(0, 256, (T, 3, H, W), generator=g, dtype=torch.uint8)
        pre = torch.randint(0, 256, (T, 3, H, W), generator=g, dtype=torch.uint8)
        mel = torch.randn(T, 1, 96, 64, generator=g)
        cx, cy = (torch.rand(2, generator=g) * 0.5 + 0.25) * torch.tensor([W, H])
        r = (torch.rand(1, generator=g) * 0.2 + 0.1) * min(H, W)
        blob = ((xx - cx) ** 2 + (yy - cy) ** 2) < r * r
        inst = {"gt_classes": torch.tensor([0, 1], dtype=torch.int64), "gt_masks": torch.stack([~blob, blob])}
        batch.append({"images": images.to(device), "pre_masks": pre.to(device), "audio_log_mel": mel.to(device),
                      "instances": [{k: v.to(device) for k, v in inst.items()}]})
    return batch


def _usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    # cgroup CPU quota (containers report the host's core count through cpu_count / affinity)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def cpu_baseline_child():
    """Runs in a CPU-only child process: training steps (fwd + 39-term loss + bwd) of the CPU oracle
    (oracle/combo_oracle.py, kind "port") on BASELINE config 0 (1 clip x 5 frames): one warm-up step, then a bounded
    sample of ~12 s of CPU work (3..12 steps), median reported."""
    from oracle import combo_oracle as O
    from combo_avs_amd import combo_cfg
    from combo_avs_amd.meta_arch import build_model
    cores = _usable_cores()
    torch.set_num_threads(cores)
    cfg = combo_cfg(os.path.join(ROOT, "configs/avs_s4/COMBO_R50_bs8_90k.yaml"))
    torch.manual_seed(0)
    model = build_model(cfg)
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    params = [k for k, p in model.named_parameters() if p.requires_grad]
    for k in params:
        P[k].requires_grad_(True)
    del model
    batch = synth_batch(1, 5, 224, 224, "cpu", seed=1)

    def step():
        t0 = time.perf_counter()
        losses = O.maskformer_forward(P, batch, num_classes=2, training=True)
        total = sum(losses.values())
        torch.autograd.grad(total, [P[k] for k in params], allow_unused=True)
        return time.perf_counter() - t0
    warm = step()  # first step: allocator / thread-pool warm-up, reported but not counted
    times, budget = [], 12.0  # bounded sample: ~12 s of CPU work (at least 3 steps, at most 12)
    t_all = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_all < budget and len(times) < 12):
        times.append(step())
    times.sort()
    dt = times[len(times) // 2]
    print("CPU_BASELINE " + json.dumps({
        "value": round(5.0 / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": f"COMBO-R50 S4 bs=1 (1 clip x 5 frames 224x224), full step fwd+39-term loss+bwd of the CPU oracle, torch "
                  f"CPU fp32, {cores} threads: median of {len(times)} steps ({dt:.2f} s each, {sum(times):.0f} s of CPU work) "
                  f"after one warm-up step ({warm:.1f} s)"}), flush=True)


def cpu_baseline(timeout_s=300):
    """Bounded: a child process with a hard timeout, so the GPU bench can never hang on the host-side baseline."""
    import subprocess
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child"], env=env, capture_output=True,
                           text=True, timeout=timeout_s)
        for line in r.stdout.splitlines():
            if line.startswith("CPU_BASELINE "):
                return json.loads(line[len("CPU_BASELINE "):])
        return {"value": None, "unit": "frames/s", "cores": _usable_cores(), "kind": "port",
                "sample": "child failed: " + (r.stderr.strip().splitlines() or ["?"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "frames/s", "cores": _usable_cores(), "kind": "port",
                "sample": f"one oracle step did not finish within {timeout_s} s on this host"}


def infer_bench(args, model, batch, world, rank, dev):
    """Eval-mode throughput: dual backbones + VGGish + SEM mix + head + fused upsample/sigmoid/class-mix tail
    (csrc/infer.hip), the whole forward replayed from one hipGraph.  Prints one JSON line (not the BASELINE metric)."""
    model.eval()
    eval_batch = [{k: v for k, v in b.items() if k != "instances"} for b in batch]

    def fwd():
        with torch.no_grad():
            return model(eval_batch)
    for _ in range(2):
        fwd()
    torch.cuda.synchronize()
    step = fwd
    if not args.no_graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            static_out = fwd()
        step = g.replay
    for _ in range(args.warmup):
        step()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        frames = args.clips * 5 * world * args.steps
        print(json.dumps({
            "metric": "inference frames/sec (224x224, 5-frame clips)", "value": round(frames / elapsed, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"COMBO-R50 S4 eval forward, bs={args.clips} clips x 5 frames x 224x224 per GPU -> "
                                   "[K,224,224] semantic maps per frame", "launch": "eager" if args.no_graph else "hipGraph",
                       "parallelism": f"dp{world}"}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clips", type=int, default=8, help="clips per GPU (BASELINE config 2: 8)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"], help="host-PyTorch backbone compute dtype")
    ap.add_argument("--head-dtype", default="fp32", choices=["bf16", "fp32"], help="dense layers of the head")
    ap.add_argument("--grad-comm", default="fp32", choices=["fp32", "bf16"],
                    help="dtype of the gradient all-reduce (fp32 = the reference's DDP semantics; bf16 halves the xGMI bytes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backbone", default="r50", choices=["r50", "pvt"],
                    help="r50 = BASELINE configs[1] (default, the quoted metric); pvt = COMBO-PVTv2-B5 (configs 4-5 family)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of the captured hipGraph step")
    ap.add_argument("--mode", default="train", choices=["train", "infer"],
                    help="train = the BASELINE metric (default); infer = eval-mode forward + fused semantic-inference tail "
                         "(SURVEY 8(f) rank 4: what pred.py times)")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        cpu_baseline_child()
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("COMBO_SINGLE_DEVICE") == "1":  # functional check of the N-rank flow on a 1-GPU box (with gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("COMBO_FORCE_PG") == "1":  # COMBO_FORCE_PG: exercise the RCCL path on one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("COMBO_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import combo_avs_amd  # noqa: F401
    from combo_avs_amd import combo_cfg, msda
    from combo_avs_amd.meta_arch import build_model
    from combo_avs_amd.trainer import FlatAdamW, GraphedTrainStep, train_step

    if os.environ.get("COMBO_MIOPEN_BENCHMARK", "1") == "1":
        # MIOpen exhaustive find for the host-PyTorch backbone convolutions (+8 % frames/s; costs ~2 min of search in
        # the first warm-up step on a box with an empty MIOpen user db; COMBO_MIOPEN_BENCHMARK=0 skips it)
        torch.backends.cudnn.benchmark = True
    cfg_file = "COMBO_R50_bs8_90k.yaml" if args.backbone == "r50" else "COMBO_PVTV2B5_bs8_90k.yaml"
    cfg = combo_cfg(os.path.join(ROOT, "configs/avs_s4", cfg_file))
    torch.manual_seed(0)  # identical random-init weights on every rank (DDP broadcast equivalent)
    model = build_model(cfg).to(dev).train()
    if args.dtype == "bf16":
        model.backbone_dtype = torch.bfloat16
    if args.head_dtype == "bf16":
        model.head_dtype = torch.bfloat16
    opt = FlatAdamW(model, base_lr=cfg.SOLVER.BASE_LR, weight_decay=cfg.SOLVER.WEIGHT_DECAY,
                    backbone_multiplier=cfg.SOLVER.BACKBONE_MULTIPLIER, clip_value=cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE,
                    grad_comm_dtype=torch.bfloat16 if args.grad_comm == "bf16" else torch.float32)
    T, H, W = 5, 224, 224
    batch = synth_batch(args.clips, T, H, W, dev, seed=100 + rank)
    if args.mode == "infer":
        infer_bench(args, model, batch, world, rank, dev)
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    trace_on = os.environ.get("COMBO_BENCH_TRACE") == "1"

    def trace(msg):  # progress markers on stderr (debugging aid for multi-rank runs)
        if trace_on:
            torch.cuda.synchronize()
            print(f"[bench rank {rank} +{time.perf_counter() - t_start:.1f}s] {msg}", file=sys.stderr, flush=True)
    t_start = time.perf_counter()

    if args.no_graph:
        def step(b):
            return train_step(model, opt, b)
        for _ in range(args.warmup):
            step(batch)
        sync()
        msda.start_timing()
    else:
        # forward + loss + backward replayed from one hipGraph (captured during the first warm-up step, after MIOpen's
        # find pass); the MSDeformAttn launches are bracketed by external event-record nodes inside the graph
        graphed = GraphedTrainStep(model, opt)
        # device-side timing of the MSDeformAttn forward launches INSIDE the graph replays (HIP cannot record events in a
        # captured graph): every launch gets a slot {min start, done, sum of ticks, launches}; nodes keep theirs over replays
        n_slots = 256
        ts_buf = torch.zeros(n_slots, 4, dtype=torch.int64, device=dev)
        ts_buf[:, 0] = -1  # ~0ull
        from combo_avs_amd import _lib as _clib
        _clib.check(_clib.lib().combo_msda_set_timing_buffer(ts_buf.data_ptr(), n_slots), "combo_msda_set_timing_buffer")
        trace("model built")
        train_step(model, opt, batch)  # eager: MIOpen find / hipBLASLt heuristics / lazy init
        trace("eager step done")
        try:
            graphed(batch)  # captures
            trace("captured + first replayed step done")
            step = graphed
        except Exception as exc:  # noqa: BLE001 - a failed capture must not cost the run: fall back to eager launches
            print(f"[bench] hipGraph capture failed ({type(exc).__name__}: {exc}); running eager", file=sys.stderr, flush=True)
            args.no_graph = True

            def step(b):
                return train_step(model, opt, b)
        for _ in range(max(args.warmup - 1, 0)):
            step(batch)
            trace("warm-up step done")
        sync()
        ts_buf[:, 2:] = 0  # count only the launches of the timed region
        sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(batch)
        trace("timed step done")
    sync()
    elapsed = time.perf_counter() - t0
    trace("timed region done")
    graph_fwd_us = None
    if not args.no_graph:
        torch.cuda.synchronize()
        tsv = ts_buf.cpu()
        _clib.lib().combo_msda_set_timing_buffer(None, 0)
        khz = _clib.lib().combo_wall_clock_khz()
        n_l = int(tsv[:, 3].sum())
        if khz > 0 and n_l > 0:
            graph_fwd_us = float(tsv[:, 2].sum()) / n_l / khz * 1e3
            graph_fwd_launches = n_l
    if not args.no_graph:
        # event records cannot be captured into a hipGraph on ROCm 7 (hipEventRecordExternal is rejected), so the
        # kernel durations come from two eager runs of the very same step right after the timed region
        msda.start_timing()
        for _ in range(2):
            train_step(model, opt, batch)
    kt = msda.stop_timing()
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    frames = args.clips * T * world * args.steps
    value = frames / elapsed
    # roofline of the dominant HIP kernel: MSDeformAttn forward core. Algorithmic bytes per frame-layer = 3.29 MB
    # (value 1.05 + loc 0.79 + w 0.40 + out 1.05); one launch processes clips*T frames.
    bt = args.clips * T
    S, M, D, L, Pn = 1029, 8, 32, 3, 4
    fwd_bytes = bt * (S * M * D * 4 * 2 + S * M * L * Pn * 3 * 4)
    roof = None
    if kt["fwd_us"] or graph_fwd_us:
        eager_us = sum(kt["fwd_us"]) / len(kt["fwd_us"]) if kt["fwd_us"] else None
        # launch duration inside the TIMED graph replays (device-side timestamps of the kernel itself) when available,
        # else the HIP-event figure of the eager steps after the timed region
        avg_us = graph_fwd_us if graph_fwd_us else eager_us
        achieved = fwd_bytes / (avg_us * 1e-6) / 1e9
        # HBM traffic per launch from the PMC pass committed under profiles/ (FETCH_SIZE doubled as the gfx950 guide
        # prescribes for 16-B/lane streams, + WRITE_SIZE); only valid for the shape it was collected on.
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_msda_pmc.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                rec = json.load(f).get("msda_fwd_tap_d32", {})
            if rec.get("frames_per_launch") == bt:
                traffic = rec.get("hbm_bytes_per_launch")
        roof = {"kernel": "msda_fwd_tap_d32", "bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": traffic, "avg_launch_us": round(avg_us, 2),
                "launches": graph_fwd_launches if graph_fwd_us else len(kt["fwd_us"]),
                "timing": ("device-side wall-clock timestamps of the kernel over the launches of the timed graph replays"
                           if graph_fwd_us else "HIP events around the launches of two eager steps after the timed region"),
                "eager_hip_event_us": round(eager_us, 2) if eager_us else None,
                "algorithmic_bytes_per_launch": fwd_bytes,
                # measured ceilings of this chip (tools/clock_probe.py, profiles/r01_clock_probe.txt): a 1 GiB device copy
                # moves 4.75 TB/s; hipBLASLt's bf16 GEMM reaches 1.37 PFLOP/s at the 1400 W package limit (sclk ~1.9 GHz)
                "measured_ceilings": {"hbm_copy_GBps": 4750.0, "hipblaslt_bf16_TFLOPs": 1374.0},
                # by time the largest single kernel of the step is the forward / dX GEMM (~10 %, MFMA-bound, reported in
                # other_kernels.gemm_nt_x3); this object stays on the north-star's core op, which is HBM-bound
                "largest_kernel_by_time": "gemm_nt2_kernel (other_kernels.gemm_nt_x3)"}
    # secondary rooflines (same HIP-event pass): the two hand-written 3xbf16 GEMM kernels against the dense bf16 MFMA
    # peak (2.5 PFLOP/s); MFMA flops = 3 products x 2*M*N*K.  Only the large launches (>= 1 GFLOP) are counted.
    kernels = {}
    for kind in ("gemm_nt_x3", "gemm_tn_x3", "conv3x3_x3", "conv3x3_wgrad_x3"):  # conv3x3: FPN 3x3 as implicit GEMMs, meta = (tokens, Cout, 9*Cin)
        evs = [(us, meta) for us, meta in kt.get("kernels", {}).get(kind, []) if meta and 2.0 * meta[0] * meta[1] * meta[2] >= 1e9]
        if evs:
            flops = sum(3 * 2.0 * m[0] * m[1] * m[2] for _, m in evs)
            tsum = sum(us for us, _ in evs) * 1e-6
            kernels[kind] = {"bound": "mfma", "launches_timed": len(evs), "avg_launch_us": round(tsum / len(evs) * 1e6, 1),
                             "achieved": round(flops / tsum / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s (bf16 MFMA, 3 products per fp32 MAC)",
                             "frac": round(flops / tsum / 2.5e15, 4), "fp32_equivalent_tflops": round(flops / 3 / tsum / 1e12, 1)}
    grp = kt.get("kernels", {}).get("gemm_tn_x3_grouped", [])
    if grp:  # all weight-gradient GEMMs of the head in one grouped launch per step (ops.linear.deferred_dw)
        tsum = sum(us for us, _ in grp) * 1e-6
        flops = sum(3 * m[0] for _, m in grp)
        kernels["gemm_tn_x3 (grouped launches, %d problems per step)" % (sum(m[1] for _, m in grp) // 2)] = {  # (2 eager steps timed)
            "bound": "mfma", "launches_timed": len(grp), "avg_launch_us": round(tsum / len(grp) * 1e6, 1),
            "achieved": round(flops / tsum / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s (bf16 MFMA, 3 products per fp32 MAC)",
            "frac": round(flops / tsum / 2.5e15, 4), "fp32_equivalent_tflops": round(flops / 3 / tsum / 1e12, 1)}
    bwd = kt.get("bwd_us") or []
    if bwd:
        bwd_bytes = bt * 5.53e6  # SURVEY 8(d)-style count: value grad 1.05 + loc/w in 1.19 + grads out 1.19 + grad_out 1.05 + value 1.05 MB
        kernels["msda_bwd (value + loc/w kernels)"] = {"bound": "hbm", "avg_launch_us": round(sum(bwd) / len(bwd), 1),
                                                       "achieved": round(bwd_bytes / (sum(bwd) / len(bwd) * 1e-6) / 1e9, 1),
                                                       "peak": 8000.0, "unit": "GB/s",
                                                       "frac": round(bwd_bytes / (sum(bwd) / len(bwd) * 1e-6) / 8e12, 4)}
    if rank == 0:
        out = {
            "metric": "train frames/sec (224x224, 5-frame clips)", "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"COMBO-{'R50' if args.backbone == 'r50' else 'PVTv2-B5'} S4, bs={args.clips} clips x 5 frames x 224x224 per GPU, full train step "
                                   "(fwd + 39-term loss + bwd + all-reduce + clip + AdamW), random-init weights",
                       "launch": "eager" if args.no_graph else "hipGraph (fwd+loss+bwd captured; all-reduce + AdamW eager)",
                       "grad_all_reduce": args.grad_comm,
                       "global_batch_clips": args.clips * world, "frames_per_clip": T, "parallelism": f"dp{world}",
                       "precision": "bf16 backbones (host PyTorch), fp32 head + HIP kernels" if args.dtype == "bf16" else "fp32"},
            "roofline": roof,
            "other_kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
