#!/usr/bin/env python3
"""Headline benchmark of the FLAIR-1 segmentation hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N>1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or
started plainly — then this process only SPAWNS the N ranks (one per GPU, RCCL over xGMI) and never touches a GPU
itself; it exits with the worst child exit code and rank 0's JSON line is the output.

One "step" = one pass of the hot path over one batch of synthetic tiles already resident in HBM:
U-Net/ResNet34 forward -> fused CE/argmax/confusion-matrix head -> backward -> (RCCL gradient all-reduce,
N>1) -> SGD.  Workload = BASELINE.json configs[1]: 5-channel 512x512 tiles, 13 classes, bf16 compute,
batch 32 per GPU (weak scaling).  Rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline      dominant kernel, timed live with HIP events on its launch stream during the last timed step
  cpu_baseline  the oracle (torch-CPU fp32 restatement of the reference's CPU path) on a bounded sample
  surface       (N=1, outside the timed region) the same step driven through the reference's own surface:
                segmentation_task_training.training_step -> loss.backward() -> torch.optim.SGD.step()
                (task_module.py:82-86, tasks_utils.py:88-95), with uint8 labels and with the reference's fp32 one-hot mask
                (data_loader.py:65-69), and predict_step at batch 1 (data_module.py:100)
  exchange      (N>1) ranks that took part in a collective, every rank's own ms per step, and the exposed communication:
                ms per step of the timed region minus ms per step of the same ranks with the gradient exchange switched off
Rehearsal of the N>1 branch without N GPUs (tests): FLAIR_BENCH_BACKEND=gloo, FLAIR_BENCH_DEVICES=0,0 (rank -> device).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver (multi-process runs)

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "flair-1_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

TRAIN_GFLOP_PER_TILE = 190.66   # fwd + dgrad + wgrad, C=13, 5 channels
FWD_GFLOP_PER_TILE = 64.24
PEAK = {"bf16": 2500.0, "f32": 157.3}   # dense MFMA TFLOP/s, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def collect_profile(L):
    n = L.lib().flair_profile_stop()
    if n < 0:
        return []
    out = []
    name = C.create_string_buffer(96)
    ms, fl, by, cnt = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
    for i in range(n):
        L.lib().flair_profile_kernel(i, name, 96, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by))
        out.append({"kernel": name.value.decode(), "ms": ms.value, "launches": cnt.value, "flops": fl.value, "bytes": by.value})
    return sorted(out, key=lambda r: -r["ms"])


PMC_FILE = "profiles/r3_pmc_traffic.json"


def pmc_traffic(kernel):
    """(HBM bytes per launch of `kernel`, commit the PMC passes were taken at) from the committed PMC passes
    (scripts/pmc_traffic.py over rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, regenerated with the
    kernel profiles every round); (None, None) when no pass covers the kernel.  It is a constant of that commit, not a
    measurement of this run: `traffic_source_commit` says how old it is."""
    try:
        t = json.load(open(os.path.join(ROOT, PMC_FILE)))
        return t["kernels"][kernel]["hbm_bytes_per_launch"], t.get("commit")
    except Exception:  # noqa: BLE001
        return None, None


def host_cores():
    """CPU cores this process may actually use: cgroup quota (the GPU box gives a 16-core share of a big host),
    else the affinity mask."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, min(n, int(os.environ.get("FLAIR_CPU_BASELINE_THREADS", "16"))))


def cpu_baseline(batch, size, classes, iters, budget_s=12.0, big_batch=32):
    """The reference's CPU path (torch-CPU fp32 NCHW: model.py:64 + task_module.py:65-80 + SGD), restated
    by oracle/, timed on this host's cores on bounded samples of the same workload: training steps at
    batch 2 (BASELINE config 1; the headline `value`) and at the bench batch, forward-only (eval, no_grad) at batch 2."""
    from oracle import seg_step
    from oracle import unet_resnet34 as om
    cores = host_cores()
    torch.set_num_threads(cores)
    m = om.seeded_model(5, classes, seed=2022).train()
    g = torch.Generator().manual_seed(2022)

    def data(b):
        x = torch.randn(b, 5, size, size, generator=g)
        lab = torch.randint(0, classes, (b, size, size), generator=g)
        return x, torch.stack([(lab == i) for i in range(classes)], 1).float()

    def train_leg(b, max_iters, budget):
        x, msk = data(b)

        def one():
            for p in m.parameters():
                p.grad = None
            loss, _, _ = seg_step.step_torch(m(x), msk)
            loss.backward()
            seg_step.sgd_step_(m.parameters(), 0.02)

        if b <= 2:
            one()  # warm-up
        t0 = time.perf_counter()
        done = 0
        while done < max_iters and (done == 0 or (time.perf_counter() - t0) < budget):
            one()
            done += 1
        return b * done / (time.perf_counter() - t0), done, time.perf_counter() - t0

    v2, n2, t2 = train_leg(batch, iters, budget_s)
    vb, nb, tb = train_leg(big_batch, 1, budget_s)      # one step of the bench batch (~10 s)
    m.eval()
    x, _ = data(batch)
    with torch.no_grad():
        m(x)
        t0 = time.perf_counter()
        nf = 0
        while nf < 40 and (time.perf_counter() - t0) < budget_s * 0.6:
            m(x)
            nf += 1
        tf = time.perf_counter() - t0
    return {"value": round(v2, 3), "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": f"{n2} train steps (fwd+CE+bwd+SGD) of batch {batch} x 5x{size}x{size}, torch-CPU fp32 oracle, "
                      f"{torch.get_num_threads()} threads, {t2:.1f} s",
            "train_bs%d_tiles_per_s" % big_batch: round(vb, 3), "train_bs%d_sample" % big_batch: f"{nb} step, {tb:.1f} s",
            "inference_tiles_per_s": round(batch * nf / tf, 3), "inference_sample": f"{nf} eval forwards of batch {batch}, {tf:.1f} s"}


def roofline_obj(prof, args, full_workload, kernel=None):
    """The dominant kernel of a profiled step (or `kernel`): algorithmic flops (MFMA-bound) or bytes (HBM-bound) per launch
    over its average HIP-event duration on its launch stream."""
    tot = sum(r["ms"] for r in prof)
    d = next((r for r in prof if r["kernel"] == kernel), prof[0]) if kernel else prof[0]
    per_launch_s = d["ms"] / 1e3 / d["launches"]
    if d["flops"] > 0:
        ach = d["flops"] / d["launches"] / per_launch_s / 1e12
        traffic, commit = pmc_traffic(d["kernel"]) if full_workload else (None, None)
        return {"kernel": d["kernel"], "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK[args.dtype],
                "unit": "TFLOP/s", "frac": round(ach / PEAK[args.dtype], 4), "traffic": traffic,
                "traffic_unit": f"HBM bytes/launch (PMC passes of this command, {PMC_FILE})", "traffic_source_commit": commit,
                "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
                "launches_per_step": d["launches"], "avg_launch_us": round(per_launch_s * 1e6, 2),
                "share_of_kernel_time": round(d["ms"] / tot, 3)}
    ach = d["bytes"] / d["launches"] / per_launch_s / 1e9
    traffic, commit = pmc_traffic(d["kernel"]) if full_workload else (None, None)
    return {"kernel": d["kernel"], "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source_commit": commit,
            "launches_per_step": d["launches"], "avg_launch_us": round(per_launch_s * 1e6, 2),
            "share_of_kernel_time": round(d["ms"] / tot, 3)}


def surface_throughput(args, dev, img, lab):
    """tiles/s of the hot path driven the way src/flair drives it: the LightningModule mirror's
    training_step (forward through the smp-surface module, fused criterion), loss.backward() through autograd, and
    torch.optim.SGD.step() on the nn.Parameters; predict_step at the reference's predict batch size 1."""
    import flair_amd
    from flair_amd.task_module import segmentation_task_predict, segmentation_task_training
    classes = {i + 1: [1.0, f"c{i + 1}"] for i in range(args.classes)}
    cfg = {"model_framework": {"model_provider": "SegmentationModelsPytorch",
                               "SegmentationModelsPytorch": {"encoder_decoder": "resnet34_unet"}},
           "use_metadata": False, "channels": [1, 2, 3, 4, 5], "classes": classes}
    os.environ["FLAIR_AMD_DTYPE"] = args.dtype
    model = flair_amd.FLAIR_ModelFactory(cfg, compute_dtype=args.dtype).to(dev)
    crit = flair_amd.FusedCrossEntropyLoss(weight=torch.ones(args.classes)).to(dev)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    task = segmentation_task_training(model, classes, criterion=crit, optimizer=opt).to(dev)
    task.setup("fit")
    task.train()
    out = {"path": "flair_amd.segmentation_task_training.training_step -> loss.backward() -> torch.optim.SGD.step() "
                   "(task_module.py:82-86, tasks_utils.py:88-95); zero_grad(set_to_none=True) as Lightning does"}

    def run(batch, n):
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            task.training_step(batch, 0).backward()
            opt.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            opt.zero_grad(set_to_none=True)
            task.training_step(batch, i).backward()
            opt.step()
        torch.cuda.synchronize()
        return args.batch * n / (time.perf_counter() - t0)

    n = max(3, min(args.steps, 10))
    out["train_tiles_per_s"] = round(run({"img": img, "msk": lab}, n), 2)
    out["train_labels"] = "uint8 class index (B,H,W)"
    onehot = torch.zeros(args.batch, args.classes, args.size, args.size, dtype=torch.float32, device=dev)
    onehot.scatter_(1, lab.long().unsqueeze(1), 1.0)
    out["train_onehot_tiles_per_s"] = round(run({"img": img, "msk": onehot}, n), 2)
    out["train_onehot_labels"] = "fp32 one-hot (B,C,H,W), the reference's batch['msk'] (data_loader.py:65-69)"
    del onehot
    pred = segmentation_task_predict(model, args.classes).to(dev).eval()
    one = {"img": img[:1], "msk": lab[:1]}
    with torch.no_grad():
        for _ in range(3):
            pred.predict_step(dict(one), 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = 30
        for i in range(m):
            pred.predict_step(dict(one), i)
        torch.cuda.synchronize()
        out["predict_bs1_tiles_per_s"] = round(m / (time.perf_counter() - t0), 2)
        big = {"img": img, "msk": lab}
        pred.predict_step(dict(big), 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            pred.predict_step(dict(big), i)
        torch.cuda.synchronize()
        out["predict_bs%d_tiles_per_s" % args.batch] = round(args.batch * 5 / (time.perf_counter() - t0), 2)
    del task, pred, model, opt
    torch.cuda.empty_cache()
    return out


def spawn_ranks(n, cmd, env=None, timeout=None):
    """Start `cmd` n times as ranks 0..n-1 of one node (what torch.distributed.run / Lightning's DDP launcher does for the
    reference, src/flair/tasks.py:83-88) and wait.  Rank 0 inherits stdout; the other ranks' stdout goes to stderr.
    Returns the worst exit code.  The caller must not have initialised the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ if env is None else env)
    base.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                 "HSA_ENABLE_IPC_MODE_LEGACY": base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    procs = []
    for r in range(n):
        e = dict(base)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        procs.append(subprocess.Popen(cmd, env=e, stdout=None if r == 0 else sys.stderr))
    t_end = None if timeout is None else time.time() + timeout
    worst = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in alive:       # one rank failed: the others would wait in a collective for ever
                    q.terminate()
        if alive:
            if t_end is not None and time.time() > t_end:
                for q in alive:
                    q.kill()
                worst = worst or 124
            time.sleep(0.05)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="tiles per GPU per step")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--classes", type=int, default=13)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--train-only", action="store_true",
                    help="skip the eval-mode inference leg (profiling runs: only train-step launches in the trace)")
    ap.add_argument("--kernels", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--no-alone", action="store_true", help="skip the extra single-stream step behind `roofline_alone`")
    ap.add_argument("--force-exchange", action="store_true",
                    help="rehearse the RCCL gradient exchange (events, side stream, buckets) with a single rank")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain start: become the launcher.  device_count() does not initialise the GPU on this image.
        have = torch.cuda.device_count()
        if have < args.gpus and not os.environ.get("FLAIR_BENCH_DEVICES"):
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible on this node", file=sys.stderr)
            raise SystemExit(2)
        raise SystemExit(spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    devmap = os.environ.get("FLAIR_BENCH_DEVICES")   # rehearsal: several ranks on one GPU
    dev_index = int(devmap.split(",")[local_rank]) if devmap else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = os.environ.get("FLAIR_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; "gloo" rehearses the branch on CPU sockets
    if world > 1 or args.force_exchange:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    import flair_amd
    from flair_amd import _lib as L

    torch.manual_seed(2022)
    model = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=args.classes,
                                   compute_dtype=args.dtype).to(dev).train()
    # configs/flair-1-config.yaml:50 lr 0.02 would diverge on pure-noise tiles over many steps; the step
    # arithmetic is identical for any lr, so use a small one to keep activations finite (DVFS-realistic data).
    trainer = flair_amd.SegTrainer(model, lr=1e-3, overlap=not args.no_overlap, force_exchange=args.force_exchange)
    g = torch.Generator().manual_seed(2022 + rank)  # per-rank tiles
    img = torch.randn(args.batch, 5, args.size, args.size, generator=g).to(dev)
    lab = torch.randint(0, args.classes, (args.batch, args.size, args.size), generator=g).to(torch.uint8).to(dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_step(img, lab)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - 1:
            L.check(L.lib().flair_profile_start(4096))  # events around every kernel of the LAST timed step
        trainer.train_step(img, lab)
    sync()
    dt = time.perf_counter() - t0
    prof = collect_profile(L)
    loss = float(trainer.loss.item())

    full_workload = args.dtype == "bf16" and args.batch == 32 and args.size == 512

    # the dominant kernel alone: one more step with the weight gradients on the caller's stream (outside the timed region)
    prof_alone = []
    if world == 1 and not args.no_alone:   # (single rank only: a lone extra step would wait in the all-reduce for ever)
        L.lib().flair_tune_set(b"FLAIR_WGRAD_STREAM", 0)
        trainer.train_step(img, lab)
        L.check(L.lib().flair_profile_start(4096))
        trainer.train_step(img, lab)
        prof_alone = collect_profile(L)
        L.lib().flair_tune_set(b"FLAIR_WGRAD_STREAM", 1)

    # inference tiles/s (eval-mode forward + argmax(softmax)), same tiles
    inf_iters = 0 if args.train_only else max(3, args.steps // 2)
    for _ in range(2 if inf_iters else 0):
        trainer.predict(img)
    sync()
    t1 = time.perf_counter()
    for _ in range(inf_iters):
        trainer.predict(img)
    sync()
    dti = time.perf_counter() - t1
    prof_inf = []
    if inf_iters and rank == 0:
        L.check(L.lib().flair_profile_start(4096))
        trainer.predict(img)
        prof_inf = collect_profile(L)

    # fp32 parity mode (exact v_mfma_f32_16x16x4_f32 arithmetic, the mode the mask / logit parity is asserted in): same
    # workload, a few steps, outside the timed region
    fp32 = None
    if rank == 0 and world == 1 and args.dtype == "bf16" and not args.train_only:
        del trainer, model
        torch.cuda.empty_cache()
        m32 = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=args.classes,
                                     compute_dtype="f32").to(dev).train()
        t32 = flair_amd.SegTrainer(m32, lr=1e-3)
        for _ in range(2):
            t32.train_step(img, lab)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n32 = 4
        for _ in range(n32):
            t32.train_step(img, lab)
        torch.cuda.synchronize()
        d32 = (time.perf_counter() - t2) / n32
        v32 = args.batch / d32
        fp32 = {"fp32_parity_tiles_per_s": round(v32, 2), "fp32_parity_ms_per_step": round(1e3 * d32, 3),
                "fp32_parity_frac_of_157TF": round(v32 * TRAIN_GFLOP_PER_TILE / 1e3 / PEAK["f32"], 4)}
        del t32, m32
        torch.cuda.empty_cache()

    exchange = None
    if world > 1:
        own_ms = 1e3 * dt / args.steps
        # the same ranks without the gradient exchange (each rank's own compute; the replicas drift apart from here on, which
        # no longer matters): what the exchange leaves exposed is the difference to the timed region
        trainer.exchange = False
        k2 = max(2, min(args.steps, 10))
        trainer.train_step(img, lab)
        sync()
        t3 = time.perf_counter()
        for _ in range(k2):
            trainer.train_step(img, lab)
        sync()
        solo_ms = 1e3 * (time.perf_counter() - t3) / k2
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones)
        per_rank = [torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([own_ms, solo_ms], dtype=torch.float64, device=dev))
        t = torch.tensor([dt, dti], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dti = float(t[0]), float(t[1])
        solo_max = max(float(p[1]) for p in per_rank)
        exchange = {"backend": "rccl" if backend == "nccl" else backend, "collective_ranks": int(round(float(ones[0]))),
                    "per_rank_ms_per_step": [round(float(p[0]), 3) for p in per_rank],
                    "ms_per_step_without_exchange": round(solo_max, 3),
                    "exposed_comm_ms_per_step": round(1e3 * dt / args.steps - solo_max, 3),
                    "overlap": not args.no_overlap, "buckets": len(trainer.buckets),
                    "bucket_mbytes": [round(4e-6 * (e - b), 2) for b, e in trainer.buckets]}

    # the same step through the reference's own surface (LightningModule hooks, autograd, torch.optim.SGD)
    surface = None
    if rank == 0 and world == 1 and not args.train_only:
        surface = surface_throughput(args, dev, img, lab)

    if rank == 0:
        tiles = args.batch * world * args.steps
        value = tiles / dt
        res = {
            "metric": "512x512x5 tiles/s (train fwd+bwd)", "value": round(value, 2), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"U-Net/ResNet34 5-ch {args.size}x{args.size} {args.classes}-class train step "
                                   f"(fwd + fused CE/argmax/confmat + bwd + SGD), bs={args.batch}/GPU, random-init weights",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "grad_allreduce": "none" if world == 1 else ("rccl bucketed, overlapped" if not args.no_overlap else "rccl bucketed")},
            "inference_tiles_per_s": round(args.batch * world * inf_iters / dti, 2) if inf_iters else None,
            "train_tflops": round(value * TRAIN_GFLOP_PER_TILE / 1e3, 2),
            "frac_of_mfma_peak_whole_step": round(value * TRAIN_GFLOP_PER_TILE / 1e3 / (PEAK[args.dtype] * world), 4),
            "final_loss": round(loss, 5),
        }
        if prof:
            tot = sum(r["ms"] for r in prof)
            res["roofline"] = roofline_obj(prof, args, full_workload)
            res["roofline"]["note"] = ("timed region as it runs: in backward this kernel shares the chip with the persistent weight-gradient "
                                       "kernel of the side stream (space sharing, 128 of 256 CUs in bf16 mode); `roofline_alone` is the "
                                       "same kernel with every CU")
            # sum over BOTH streams (weight gradients run beside the BN / data-gradient chain): exceeds ms_per_step
            res["kernel_time_ms_per_step"] = round(tot, 3)
            res["kernel_time_note"] = "sum of per-kernel HIP-event times over two overlapping streams"
            if args.kernels:
                for r in prof:
                    tf = r["flops"] / (r["ms"] / 1e3) / 1e12 if r["ms"] > 0 else 0
                    gb = r["bytes"] / (r["ms"] / 1e3) / 1e9 if r["ms"] > 0 else 0
                    print(f"{r['kernel']:28s} {r['ms']:9.3f} ms {r['launches']:5d} launches {tf:8.1f} TFLOP/s {gb:9.1f} GB/s(alg)", file=sys.stderr)
        if prof_alone:
            ra = roofline_obj(prof_alone, args, full_workload, kernel=res.get("roofline", {}).get("kernel"))
            ra["note"] = ("the same kernel in one extra step with the weight-gradient side stream off (FLAIR_WGRAD_STREAM=0): "
                          "no other kernel shares the CUs; `roofline` above is the timed region as it runs")
            res["roofline_alone"] = ra
        if prof_inf:
            res["inference_roofline"] = roofline_obj(prof_inf, args, full_workload)
        if fp32:
            res.update(fp32)
        if exchange:
            res["exchange"] = exchange
        if surface:
            surface["train_frac_of_value"] = round(surface["train_tiles_per_s"] / value, 4)
            res["surface"] = surface
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(2, args.size, args.classes, iters=24)
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
