#!/usr/bin/env python3
"""train_ours.py -- MI355X counterpart of the reference entry point (train_ours.py:730-824).

Keeps what the hot path needs from the reference trainer: YAML config with `model.name/args`,
one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from torch.distributed.run), per-rank seeds
(`seed + rank`, train_ours.py:737), the iteration body (forward -> Lap+census loss with the 0.1
weighting that flips at 10k iterations -> backward -> Adam, train_ours.py:250-277), the loss
all-reduce for logging (myutils/utils.py:80-92) and the checkpoint layout
{model:{name,states}, optimizer, config, trainer} (train_ours.py:621-671) incl. --resume.
Unlike the reference (whose fwd+bwd sits inside model.no_sync()) gradients ARE averaged across
ranks every step (one flat RCCL all-reduce).  Data: synthetic batches; the HDF5
pipeline, TensorBoard, validation and early stopping of the reference are out of scope.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_ours.py -c config/train_ours.yml -id run
    python train_ours.py -c config/train_ours.yml -id run --iterations 20
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist
import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ebfi_amd.dp import reduce_tensor  # noqa: E402
from ebfi_amd.engine import Engine, synthetic_batch, synthetic_batch_from_raw_events  # noqa: E402


def init_distributed_mode():
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        rank, world, gpu = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    else:
        rank, world, gpu = 0, 1, 0
    torch.cuda.set_device(gpu)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://", world_size=world, rank=rank,
                                device_id=torch.device("cuda", gpu))
        dist.barrier()
    return rank, world, gpu


def save_checkpoint(path, eng, config, iteration):
    """Same keys as the reference's _save_checkpoint so infer_ours.py / the reference can read it."""
    state = {"model": {"name": config["model"]["name"], "states": eng.model.state_dict()},
             "optimizer": {"name": "Adam", "states": eng.optimizer.state_dict()},
             "config": config,
             "trainer": {"training_mode": "iteration_based_train", "iteration": iteration, "monitor_best": None}}
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(state, path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "config", "train_ours.yml"))
    ap.add_argument("-id", "--runid", default="run")
    ap.add_argument("-r", "--resume", default=None, help="checkpoint to resume from")
    ap.add_argument("-seed", "--seed", type=int, default=123)
    ap.add_argument("--iterations", type=int, default=None)
    ap.add_argument("--precision", default="bf16x3", choices=["fp32", "bf16x3", "bf16"],
                    help="matrix-core operands of the convs: bf16x3 = split bf16 pairs, fp32-grade accuracy (default); fp32 = exact")
    ap.add_argument("--graph", action="store_true", help="replay forward+loss+backward from a captured hipGraph")
    ap.add_argument("--raw-events", action="store_true",
                    help="build the event tensor from synthetic raw event lists with the device events_to_stack kernel "
                         "(the reference's data path, h5dataset.py:327-352) instead of drawing voxel counts directly")
    args = ap.parse_args()
    with open(args.config) as fh:
        config = yaml.safe_load(fh)
    tr = config.get("trainer", {})
    iterations = args.iterations or int(tr.get("iterations", 100))
    rank, world, gpu = init_distributed_mode()
    device = torch.device("cuda", gpu)
    assert config["model"]["name"] == "EVFIAutoEx", "only the EVFIAutoEx hot path is implemented"

    eng = Engine(config["model"]["args"], device=device, precision=args.precision,
                 lr=float(config["optimizer"]["args"]["lr"]), seed=args.seed,      # same init on every rank
                 graph=args.graph or bool(tr.get("graph", False)))
    start = 0
    if args.resume:
        cpt = torch.load(args.resume, map_location=device)
        eng.model.load_state_dict(cpt["model"]["states"], strict=False)
        eng.optimizer.load_state_dict(cpt["optimizer"]["states"])
        start = eng.iteration = int(cpt["trainer"]["iteration"])
    B, H, W = int(tr.get("batch_size", 8)), int(tr.get("height", 256)), int(tr.get("width", 256))
    TB = int(config["model"]["args"]["TB"])
    out_dir = os.path.join(tr.get("output_path", "./output"), "models", config.get("experiment", "Ours"), args.runid)
    save_period = int(tr.get("save_period", 0))

    t0, frames = time.perf_counter(), 0
    for it in range(start, iterations):
        # one fresh synthetic batch per iteration, different on every rank (seed + rank, like the reference)
        make = synthetic_batch_from_raw_events if args.raw_events else synthetic_batch
        batch = make(B, H, W, TB, device=device, seed=args.seed + 1000 * it, rank=rank)
        loss = reduce_tensor(eng.train_step(*batch).clone())
        frames += B * world
        if rank == 0 and (it % 10 == 0 or it == iterations - 1):
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("Iteration: %d/%d train_loss: %.4e  %.1f frames/s" % (it, iterations, loss.item(), frames / dt), flush=True)
        if rank == 0 and save_period and it and it % save_period == 0:
            save_checkpoint(os.path.join(out_dir, "checkpoint-iteration%d.pth" % it), eng, config, it)
        if world > 1:
            dist.barrier()
    if rank == 0:
        save_checkpoint(os.path.join(out_dir, "checkpoint-iteration%d.pth" % iterations), eng, config, iterations)
        print("saved", os.path.join(out_dir, "checkpoint-iteration%d.pth" % iterations))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
