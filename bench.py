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


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (scripts/pmc_traffic.py over rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same command); None when no pass covers it."""
    try:
        import json as _j
        t = _j.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_pmc_traffic.json")))
        return t["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        return None


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


def cpu_baseline(batch, size, classes, iters, budget_s=25.0):
    """The reference's CPU path (torch-CPU fp32 NCHW: model.py:64 + task_module.py:65-80 + SGD), restated
    by oracle/, timed on this host's cores on a bounded sample of the same workload."""
    from oracle import seg_step
    from oracle import unet_resnet34 as om
    cores = host_cores()
    torch.set_num_threads(cores)
    m = om.seeded_model(5, classes, seed=2022).train()
    g = torch.Generator().manual_seed(2022)
    x = torch.randn(batch, 5, size, size, generator=g)
    lab = torch.randint(0, classes, (batch, size, size), generator=g)
    msk = torch.stack([(lab == i) for i in range(classes)], 1).float()

    def one():
        for p in m.parameters():
            p.grad = None
        loss, _, _ = seg_step.step_torch(m(x), msk)
        loss.backward()
        seg_step.sgd_step_(m.parameters(), 0.02)

    one()  # warm-up
    t0 = time.perf_counter()
    done = 0
    while done < iters and (time.perf_counter() - t0) < budget_s:
        one()
        done += 1
    dt = time.perf_counter() - t0
    return {"value": round(batch * done / dt, 3), "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": f"{done} train steps (fwd+CE+bwd+SGD) of batch {batch} x 5x{size}x{size}, torch-CPU fp32 oracle, "
                      f"{torch.get_num_threads()} threads, {dt:.1f} s"}


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
    ap.add_argument("--force-exchange", action="store_true",
                    help="rehearse the RCCL gradient exchange (events, side stream, buckets) with a single rank")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain start: become the launcher.  device_count() does not initialise the GPU on this image.
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible on this node", file=sys.stderr)
            raise SystemExit(2)
        raise SystemExit(spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_exchange:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

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

    if world > 1:
        t = torch.tensor([dt, dti], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dti = float(t[0]), float(t[1])

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
            d = prof[0]
            per_launch_s = d["ms"] / 1e3 / d["launches"]
            if d["flops"] > 0:
                ach = d["flops"] / d["launches"] / per_launch_s / 1e12
                res["roofline"] = {"kernel": d["kernel"], "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK[args.dtype],
                                   "unit": "TFLOP/s", "frac": round(ach / PEAK[args.dtype], 4),
                                   "traffic": pmc_traffic(d["kernel"]) if args.dtype == "bf16" and args.batch == 32 and args.size == 512 else None,
                                   "traffic_unit": "HBM bytes/launch (PMC, profiles/r1_pmc_traffic.json)",
                                   "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
                                   "launches_per_step": d["launches"], "avg_launch_us": round(per_launch_s * 1e6, 2),
                                   "share_of_kernel_time": round(d["ms"] / tot, 3)}
            else:
                ach = d["bytes"] / d["launches"] / per_launch_s / 1e9
                res["roofline"] = {"kernel": d["kernel"], "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                                   "launches_per_step": d["launches"], "avg_launch_us": round(per_launch_s * 1e6, 2),
                                   "share_of_kernel_time": round(d["ms"] / tot, 3)}
            # sum over BOTH streams (weight gradients run beside the BN / data-gradient chain): exceeds ms_per_step
            res["kernel_time_ms_per_step"] = round(tot, 3)
            res["kernel_time_note"] = "sum of per-kernel HIP-event times over two overlapping streams"
            if args.kernels:
                for r in prof:
                    tf = r["flops"] / (r["ms"] / 1e3) / 1e12 if r["ms"] > 0 else 0
                    gb = r["bytes"] / (r["ms"] / 1e3) / 1e9 if r["ms"] > 0 else 0
                    print(f"{r['kernel']:28s} {r['ms']:9.3f} ms {r['launches']:5d} launches {tf:8.1f} TFLOP/s {gb:9.1f} GB/s(alg)", file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(2, args.size, args.classes, iters=24)
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
