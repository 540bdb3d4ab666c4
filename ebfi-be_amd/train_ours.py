#!/usr/bin/env python3
"""train_ours.py -- MI355X counterpart of the reference entry point (train_ours.py:730-824).

Keeps what the hot path needs from the reference trainer: YAML config with `model.name/args`, `optimizer`,
`lr_scheduler`, `trainer.{accu_step, lr_min, iteration_based_train.*}`; one process per GPU (RANK / LOCAL_RANK /
WORLD_SIZE from torch.distributed.run), per-rank seeds (`seed + rank`, train_ours.py:737); the iteration body of
train_ours.py:250-347 in the reference's order -- forward -> Lap+census loss (0.1 weighting flips at 10k iterations,
divided by accu_step) -> backward -> every accu_step-th pass: Adam step, loss all-reduce for logging
(myutils/utils.py:80-92), periodic checkpoint, THEN lr_scheduler.step() (gated by lr_change_rate and lr_min, :335-338);
and the checkpoint layout {model:{name,states}, lr_scheduler:{name,states}, optimizer:{name,states}, config,
trainer:{training_mode, iteration, monitor_best}} (train_ours.py:621-671) with --resume / --reset as in
_resume_checkpoint (:673-716): training continues at trainer.iteration + 1.
Unlike the reference (whose fwd+bwd sits inside model.no_sync()) gradients ARE averaged across ranks every optimiser
step (one flat RCCL all-reduce).  Data: synthetic batches or, with --data, recorded clips through
ebfi_amd.clipdata (the tensor contract of dataloader/h5dataset.py:283-295 from .npz clips, or .h5 when h5py is installed);
TensorBoard, validation and early stopping of the reference are out of scope.

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

TRAINING_MODE = "iteration_based_train"
CHECKPOINT_KEYS = ("model", "lr_scheduler", "optimizer", "config", "trainer")       # train_ours.py:628-644


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


def build_lr_scheduler(config, optimizer):
    """`eval(config['lr_scheduler']['name'])(optimizer, **args)` of the reference (train_ours.py:760-761) for the schedulers
    of torch.optim.lr_scheduler; YAML floats such as `step_size: !!float 2e5` are made integral where torch expects it.
    No `lr_scheduler` section -> None.  An unknown name fails loudly instead of training without a schedule."""
    sec = config.get("lr_scheduler")
    if not sec or not sec.get("name"):
        return None
    cls = getattr(torch.optim.lr_scheduler, sec["name"], None)
    if cls is None:
        raise ValueError("lr_scheduler '%s' is not a torch.optim.lr_scheduler class" % sec["name"])
    args = dict(sec.get("args") or {})
    if "step_size" in args:
        args["step_size"] = int(args["step_size"])
    return cls(optimizer, **args)


def trainer_settings(config, cli_iterations=None):
    """iterations / save_period / lr_change_rate from trainer.iteration_based_train (reference layout, train_ours.yml:79-98)
    with the flat keys of this repo's small config as fallback; lr_min and accu_step from trainer."""
    tr = config.get("trainer", {}) or {}
    ib = tr.get("iteration_based_train", {}) or {}
    get = lambda k, d: ib.get(k, tr.get(k, d))
    return {"iterations": int(cli_iterations or float(get("iterations", 100))),
            "save_period": int(get("save_period", 0)),
            "lr_change_rate": max(1, int(get("lr_change_rate", 1))),
            "lr_min": float(tr.get("lr_min", 1e-6)),
            "accu_step": max(1, int(tr.get("accu_step", 1))),
            "log_step": max(1, int(get("train_log_step", 10)))}


def checkpoint_state(eng, scheduler, config, iteration, monitor_best=None):
    """The reference's checkpoint dict (train_ours.py:628-655), key for key; `iteration` = index of the last completed
    optimiser step, resumed at +1 (:695)."""
    sched_name = (config.get("lr_scheduler") or {}).get("name")
    return {"model": {"name": config["model"]["name"], "states": eng.model.state_dict()},
            "lr_scheduler": {"name": sched_name, "states": scheduler.state_dict() if scheduler is not None else {}},
            "optimizer": {"name": config["optimizer"]["name"], "states": eng.optimizer.state_dict()},
            "config": config,
            "trainer": {"training_mode": TRAINING_MODE, "iteration": int(iteration), "monitor_best": monitor_best}}


def save_checkpoint(path, eng, scheduler, config, iteration):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(checkpoint_state(eng, scheduler, config, iteration), path)


def resume_checkpoint(path, eng, scheduler, config, reset=False, map_location="cpu"):
    """Resumer + _resume_checkpoint of the reference (myutils/utils.py:178-215, train_ours.py:673-716): optimiser and
    scheduler states are restored only without --reset and when the training mode matches, each only when the configured
    name equals the checkpoint's; the model is always loaded (strict=False) when its name matches.  Returns the first
    iteration to run."""
    cpt = torch.load(path, map_location=map_location, weights_only=False)
    start = 0
    tr = cpt["trainer"]
    if not reset and tr.get("training_mode") == TRAINING_MODE:
        if config["optimizer"]["name"] == cpt["optimizer"]["name"]:
            eng.optimizer.load_state_dict(cpt["optimizer"]["states"])
        if scheduler is not None and (config.get("lr_scheduler") or {}).get("name") == cpt["lr_scheduler"]["name"]:
            scheduler.load_state_dict(cpt["lr_scheduler"]["states"])
            for group, lr in zip(eng.optimizer.param_groups, scheduler.get_last_lr()):
                group["lr"] = lr
        start = int(tr["iteration"]) + 1
    if config["model"]["name"] == cpt["model"]["name"]:
        eng.model.load_state_dict(cpt["model"]["states"], strict=False)
    eng.iteration = start
    return start


def real_data_passes(path, config, B, TB, device, rank, world, seed):
    """Endless stream of (Frame, Event, T, GTEx, LatentF) passes from recorded clips (ebfi_amd.clipdata): the dataset keys
    are the reference's (config train_dataloader.dataset, train_ours.yml:115-150); defaults = its shipped values."""
    from ebfi_amd import clipdata
    ds_cfg = ((config.get("train_dataloader") or {}).get("dataset") or {})
    ds = clipdata.ClipDataset(path, time_bins=int(ds_cfg.get("time_bins", TB)),
                              frames_per_period=int(ds_cfg.get("NumFramePerPeriod", 16)),
                              frames_per_blurry=int(ds_cfg.get("NumFramePerBlurry", 16)),
                              exposure_method=ds_cfg.get("ExposureMethod", "Custom"),
                              exposure_time=ds_cfg.get("ExposureTime", [9, 10, 11, 12, 13, 14, 15]),
                              device=device, seed=seed, **clipdata.dataset_args_from_config(ds_cfg))
    if len(ds) < B * world:
        raise SystemExit("--data: %d periods in %s, need at least batch_size x world = %d" % (len(ds), path, B * world))
    for batch in clipdata.batches(ds, B, rank=rank, world=world, seed=seed):
        for inputs in clipdata.model_inputs(batch):
            yield inputs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "config", "train_ours.yml"))
    ap.add_argument("-id", "--runid", default="run")
    ap.add_argument("-r", "--resume", default=None, help="checkpoint to resume from")
    ap.add_argument("--reset", action="store_true", help="with --resume: load the model only, restart optimiser / schedule / count")
    ap.add_argument("-seed", "--seed", type=int, default=123)
    ap.add_argument("--iterations", type=int, default=None)
    ap.add_argument("--precision", default="bf16x3", choices=["fp32", "bf16x3", "bf16"],
                    help="matrix-core operands of the convs: bf16x3 = split bf16 pairs, fp32-grade accuracy (default); fp32 = exact")
    ap.add_argument("--graph", action="store_true", help="replay forward+loss+backward from a captured hipGraph")
    ap.add_argument("--host-data", action="store_true",
                    help="draw every synthetic batch on the host (bit-identical across machines, ~0.5 s per B=8 256x256 batch) "
                         "instead of with the device generator (default: the data path must not be slower than the 20 ms step)")
    ap.add_argument("--no-f16-backward", action="store_true",
                    help="bf16x3 precision: keep the data / weight gradients on the split-precision kernels (3 MFMAs per product) "
                         "instead of the fp16 single-product ones with delayed operand scales (ebfi_amd.f16scale)")
    ap.add_argument("--no-f16-forward", action="store_true",
                    help="keep the 128 -> 1600 KernelConv of Modification on the split-precision kernel in the forward pass "
                         "(default with the fp16 backward: fp16 operands, Engine(forward_f16='filters'))")
    ap.add_argument("--data", default=None,
                    help="recorded clips instead of synthetic batches: a directory of .npz clips (or .h5 in the reference's layout "
                         "when h5py is installed), a datalist .txt, or one clip file (ebfi_amd.clipdata); the dataset section of "
                         "the config (train_dataloader.dataset, reference keys) sets periods / exposure / crop")
    ap.add_argument("--raw-events", action="store_true",
                    help="build the event tensor from synthetic raw event lists with the device events_to_stack kernel "
                         "(the reference's data path, h5dataset.py:327-352) instead of drawing voxel counts directly")
    args = ap.parse_args()
    with open(args.config) as fh:
        config = yaml.safe_load(fh)
    tr = config.get("trainer", {}) or {}
    st = trainer_settings(config, args.iterations)
    rank, world, gpu = init_distributed_mode()
    device = torch.device("cuda", gpu)
    assert config["model"]["name"] == "EVFIAutoEx", "only the EVFIAutoEx hot path is implemented"
    assert config["optimizer"]["name"] == "Adam", "only Adam (config/train_ours.yml) is implemented"
    oargs = config["optimizer"].get("args", {}) or {}

    eng = Engine(config["model"]["args"], device=device, precision=args.precision, lr=float(oargs.get("lr", 1e-4)),
                 betas=tuple(oargs.get("betas", (0.9, 0.999))), seed=args.seed,      # same init on every rank
                 graph=args.graph or bool(tr.get("graph", False)), accu_step=st["accu_step"],
                 backward_f16=False if args.no_f16_backward else None, forward_f16=None if args.no_f16_forward else "filters")
    scheduler = build_lr_scheduler(config, eng.optimizer.inner)
    start = resume_checkpoint(args.resume, eng, scheduler, config, reset=args.reset, map_location=device) if args.resume else 0
    B, H, W = int(tr.get("batch_size", 8)), int(tr.get("height", 256)), int(tr.get("width", 256))
    TB = int(config["model"]["args"]["TB"])
    out_dir = os.path.join(tr.get("output_path", "./output"), "models", config.get("experiment", "Ours"), args.runid)

    # throughput is counted from the end of the first iteration after which the engine is in its steady state (the first ones
    # pay module load, allocator growth, the just-in-time calibration of the fp16 operand scales and, with --graph, the
    # capture: Engine.settled): what is logged is the steady-state rate, whole job (all ranks)
    t0, frames, it = None, 0, start
    make = synthetic_batch_from_raw_events if args.raw_events else synthetic_batch
    real = real_data_passes(args.data, config, B, TB, device, rank, world, args.seed) if args.data else None
    while it < st["iterations"]:
        for micro in range(st["accu_step"]):
            if real is not None:
                # one pass per latent frame of the loaded periods, in the reference's order (train_ours.py:226-251)
                batch = next(real)
            else:
                # one fresh synthetic batch per pass, different on every rank (seed + rank, like the reference); drawn by the
                # device generator unless --host-data: the host draw alone would cap the loop at ~15 frames/s
                batch = make(B, H, W, TB, device=device, seed=args.seed + 1000 * (it * st["accu_step"] + micro), rank=rank,
                             on_device=not args.host_data)
            loss = eng.train_step(*batch)
            if t0 is not None:
                frames += B * world
        log_now = it % st["log_step"] == 0 or it == st["iterations"] - 1
        if log_now:                      # the loss all-reduce is for logging only (train_ours.py:278-279): do it when logging
            loss = reduce_tensor(loss.clone())
        lr_now = scheduler.get_last_lr()[0] if scheduler is not None else eng.optimizer.param_groups[0]["lr"]
        if rank == 0 and log_now:
            torch.cuda.synchronize()
            rate = frames / (time.perf_counter() - t0) if t0 is not None and frames else float("nan")
            print("Iteration: %d/%d train_loss: %.4e learning rate: %.4e  %.1f frames/s"
                  % (it, st["iterations"], loss.item(), lr_now, rate), flush=True)
        if t0 is None and eng.settled:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        # periodic checkpoints as train_ours.py:331-333 (saved BEFORE this iteration's scheduler step, like there), plus one
        # after the last iteration
        if rank == 0 and ((st["save_period"] and it % st["save_period"] == 0 and it != 0) or it == st["iterations"] - 1):
            path = os.path.join(out_dir, "checkpoint-iteration%d.pth" % it)
            save_checkpoint(path, eng, scheduler, config, it)
            print("saved", path, flush=True)
        if scheduler is not None and it % st["lr_change_rate"] == 0 and it != 0 and lr_now >= st["lr_min"]:   # :335-338
            scheduler.step()
        it += 1
    if rank == 0 and eng.book is not None:
        # fp16 backward (ebfi_amd.f16scale): optimiser steps skipped because an operand left the fp16 range (expected: 0)
        print("fp16 backward: %d of %d optimiser steps skipped by the overflow guard, %d operand scale slots"
              % (eng.book.skipped_steps(), st["iterations"] - start, len(eng.book.index)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
