"""GPU smoke of the two entry points (train_ours.py / infer_ours.py counterparts): a few training
iterations on synthetic data, checkpoint in the reference layout, resume, inference from it."""
import os
import subprocess
import sys

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ebfi-be_amd")


def test_train_resume_infer(tmp_path):
    cfg = yaml.safe_load(open(os.path.join(PKG, "config", "train_ours.yml")))
    cfg["model"]["args"].update(FrameBasech=16, EventBasech=16, InterCH=16, TB=4, step=2, channels=[4, 4, 8, 8])
    cfg["trainer"].update(batch_size=2, height=64, width=64, output_path=str(tmp_path / "out"))
    cfg["trainer"]["iteration_based_train"].update(iterations=3, save_period=1)
    cfg["lr_scheduler"]["args"].update(step_size=2.0, gamma=0.5)
    cfg_path = tmp_path / "cfg.yml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, os.path.join(PKG, "train_ours.py"), "-c", str(cfg_path), "-id", "t"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    run_dir = tmp_path / "out" / "models" / "Ours" / "t"
    assert (run_dir / "checkpoint-iteration1.pth").exists()        # periodic (save_period 1, never at iteration 0)
    ckpt = run_dir / "checkpoint-iteration2.pth"                    # last completed iteration
    assert ckpt.exists() and not (run_dir / "checkpoint-iteration0.pth").exists()
    cpt = torch.load(str(ckpt), map_location="cpu", weights_only=False)
    # exactly the reference's key set (train_ours.py:628-655): its Resumer reads model / optimizer / lr_scheduler / trainer
    assert set(cpt) == {"model", "lr_scheduler", "optimizer", "config", "trainer"} and cpt["model"]["name"] == "EVFIAutoEx"
    assert cpt["lr_scheduler"]["name"] == "StepLR" and cpt["trainer"]["iteration"] == 2
    assert cpt["trainer"]["training_mode"] == "iteration_based_train"
    assert "learning rate: 1.0000e-04" in r.stdout and "learning rate: 5.0000e-05" not in r.stdout   # decays after 2 counted steps
    assert "ResidualControl.Conv3.0.0.conv2d.weight" in cpt["model"]["states"]
    r = subprocess.run([sys.executable, os.path.join(PKG, "train_ours.py"), "-c", str(cfg_path), "-id", "t2",
                        "--resume", str(ckpt), "--iterations", "5", "--graph", "--raw-events"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "Iteration: 4/5" in r.stdout, r.stderr[-2000:] + r.stdout[-500:]
    assert "Iteration: 2/5" not in r.stdout                         # resumes at trainer.iteration + 1 (train_ours.py:695)
    assert "Iteration: 4/5" in r.stdout and "learning rate: 5.0000e-05" in r.stdout      # schedule restored and continued
    r = subprocess.run([sys.executable, os.path.join(PKG, "infer_ours.py"), "--model_path", str(ckpt), "--batch", "2",
                        "--height", "64", "--width", "64", "--num_ts", "3"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "interpolated 6 frames" in r.stdout, r.stderr[-2000:]


def test_bench_two_rank_rehearsal():
    """bench.py's multi-rank path (barrier, flat-bucket gradient all-reduce, max-over-ranks timing, rank-0 JSON)
    rehearsed with 2 ranks sharing the one GPU over gloo (EBFI_BENCH_REHEARSAL=1); the real N>1 runs use RCCL."""
    import json
    env = dict(os.environ, EBFI_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-bf16-leg"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["cpu_baseline"] is None and d["roofline"]["kernel"].startswith("conv_")
    assert d["config"]["world_size"] == 2 and d["config"]["collective_backend"] == "gloo"
    assert d["config"]["replica_param_checksum"]["ranks_identical"] is True
    # every launch of the profiled pass was timed: per-step launch counts are whole numbers
    for name, k in d["kernels"].items():
        assert float(k["launches_per_step"]).is_integer(), (name, k["launches_per_step"])


def test_graph_replay_step_equals_eager_step():
    """Engine(graph=True) replays fwd + loss + bwd + gradient packing from one hipGraph: same losses and
    parameters as the eager engine over several steps (inputs change between steps).  Run in the mode bench.py times --
    split-precision convs, so the weight-bank refresh (first node of the capture) and the fused ResidualControl node
    (default width 64) are inside the replayed graph and must pick up every optimiser update."""
    import torch
    from ebfi_amd import rc_fused
    from ebfi_amd.engine import Engine, synthetic_batch
    cfg = dict(step=2, channels=[8, 8, 16, 16])
    engines = [Engine(cfg, device="cuda", seed=7, graph=g, precision="bf16x3") for g in (False, True)]
    engines[1].model.load_state_dict(engines[0].model.state_dict())
    assert all(e.bank is not None and rc_fused.fusable(e.model.ResidualControl) for e in engines)
    for it in range(6):                          # (fp16 backward: two eager calibration steps, the capture, three replays)
        batch = synthetic_batch(2, 64, 64, device="cuda", seed=100 + it)
        losses = [e.train_step(*batch) for e in engines]
        assert torch.allclose(losses[0], losses[1], rtol=1e-5, atol=0), (it, losses)
    assert len(engines[1]._graphs) == 1 and engines[1].bucket.views_intact()
    assert all(e.book is not None and e.book.skipped_steps() == 0 for e in engines)
    for (n, a), b in zip(engines[0].model.named_parameters(), engines[1].model.parameters()):
        # (Adam normalises the gradient: where it is rounding-sized the two runs' atomics-ordered sums move a parameter by a
        #  fraction of lr = 1e-4 per step; a missed update would be 1e-4)
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-5), n


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs on the node (the round's GPU box has one)")
def test_bench_two_gpus_rccl():
    """bench.py with 2 ranks on 2 GPUs over RCCL (backend 'nccl'): the hipGraph capture of forward + loss + backward happens
    with the RCCL communicator (and its watchdog thread) alive -- Engine captures with capture_error_mode='thread_local' --
    and the replicas must stay bit-identical after the timed steps (bench.py compares per-rank parameter checksums)."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29641", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-extra-legs"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["collective_backend"] == "nccl"
    assert d["config"]["replica_param_checksum"]["ranks_identical"] is True
    assert "hipGraph replay" in d["config"]["launch"]
