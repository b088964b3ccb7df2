"""Two data-parallel ranks sharing the one GPU of a test box (gloo carries the HIP tensors; the 8-GPU RCCL run
is the driver's): the native backward's per-stage events, the side-stream bucketed all-reduce and the folded 1/world SGD
step must give every rank the same weights, equal to a single-process step on the mean of the two ranks' gradients
computed with PER-RANK BatchNorm statistics (no SyncBN, like the reference's DDP — SURVEY.md §8e)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(rank):
    g = torch.Generator().manual_seed(2022 + rank)  # per-rank seed, SURVEY.md §8d config 3
    return torch.randn(2, 5, 64, 64, generator=g), torch.randint(0, 13, (2, 64, 64), generator=g, dtype=torch.uint8)


def _model(seed):
    import flair_amd
    torch.manual_seed(seed)
    return flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=13, compute_dtype="f32")


def _worker(rank, world, port, overlap, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import flair_amd
    dev = torch.device("cuda:0")
    model = _model(100 + rank).to(dev).train()  # different initial weights: the trainer must broadcast rank 0's
    trainer = flair_amd.SegTrainer(model, lr=0.05, overlap=overlap)
    img, lab = _batch(rank)
    for _ in range(2):
        loss = trainer.train_step(img.to(dev), lab.to(dev))
    torch.cuda.synchronize()
    # eval-mode forward on each rank WITHOUT any collective of its own: the running statistics must already be rank 0's
    model.eval()
    with torch.no_grad():
        ev = model(_batch(7)[0].to(dev)).cpu()
    out[rank] = (model.flat_parameters().detach().cpu(), float(loss), model.flat_buffers().detach().cpu(), ev)
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_ranks_one_gpu_match_single_process_average(dev, overlap):
    import flair_amd
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    w0, w1 = out[0][0], out[1][0]
    assert torch.equal(w0, w1)  # replicas stay bit-identical
    # single-process replay: two replicas from rank 0's initial weights, gradients averaged by hand
    reps = [_model(100).to(dev).train() for _ in range(2)]
    trainers = [flair_amd.SegTrainer(m, lr=0.05) for m in reps]
    for _ in range(2):
        for r, (m, t) in enumerate(zip(reps, trainers)):
            img, lab = _batch(r)
            t.lr = 0.0  # gradients only
            t.train_step(img.to(dev), lab.to(dev))
        mean = (trainers[0].grads + trainers[1].grads) / 2
        for m in reps:
            m.flat_parameters().data.add_(mean, alpha=-0.05)
    ref = reps[0].flat_parameters().detach().cpu()
    assert torch.equal(reps[0].flat_parameters(), reps[1].flat_parameters())
    err = (w0 - ref).abs().max().item()
    assert err < 2e-6, err  # (g0 + g1) * (lr / 2) vs ((g0 + g1) / 2) * lr: rounding only
    start = _model(100).to(dev).flat_parameters().detach().cpu()
    assert (w0 - start).abs().max() > 1e-4  # and the step did move the weights
    # torch DDP broadcast_buffers=True (tasks.py:83-88 keeps the default): every rank holds RANK 0's running statistics — the
    # ones a single process fed rank 0's batches accumulates — so rank 1's eval logits equal rank 0's bit for bit
    assert torch.equal(out[0][2], out[1][2])
    assert torch.equal(out[0][3], out[1][3])
    b_ref = reps[0].flat_buffers().detach().cpu()
    assert torch.allclose(out[1][2], b_ref, rtol=1e-5, atol=1e-6), float((out[1][2] - b_ref).abs().max())
    assert not torch.allclose(out[1][2], reps[1].flat_buffers().detach().cpu(), rtol=1e-3, atol=1e-4)   # (rank 1's own differ)


def test_bench_n_gt_1_branch_runs_end_to_end_on_gloo():
    """Every line of bench.py's N>1 branch (self-spawned ranks, process group, per-rank tiles, barrier + max-over-ranks timing,
    the exchange object) executed once before the driver's 8-GPU RCCL run: two ranks on the one GPU of this box, gloo instead
    of RCCL (FLAIR_BENCH_BACKEND / FLAIR_BENCH_DEVICES), small tiles."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FLAIR_BENCH_BACKEND="gloo", FLAIR_BENCH_DEVICES="0,0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2",
                        "--size", "64", "--dtype", "f32", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4 and line["scaling"] == "weak"
    ex = line["exchange"]
    assert ex["collective_ranks"] == 2 and ex["backend"] == "gloo" and len(ex["per_rank_ms_per_step"]) == 2
    assert line["value"] > 0 and abs(line["value"] - 4 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-2 * line["value"]
    assert "surface" not in line and "cpu_baseline" not in line   # rank 0 at N = 1 only
