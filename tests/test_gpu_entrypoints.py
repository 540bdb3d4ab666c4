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
    cfg["trainer"].update(iterations=3, batch_size=2, height=64, width=64, output_path=str(tmp_path / "out"))
    cfg_path = tmp_path / "cfg.yml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, os.path.join(PKG, "train_ours.py"), "-c", str(cfg_path), "-id", "t"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ckpt = tmp_path / "out" / "models" / "Ours" / "t" / "checkpoint-iteration3.pth"
    assert ckpt.exists()
    cpt = torch.load(str(ckpt), map_location="cpu")
    assert set(cpt) >= {"model", "optimizer", "config", "trainer"} and cpt["model"]["name"] == "EVFIAutoEx"
    assert "ResidualControl.Conv3.0.0.conv2d.weight" in cpt["model"]["states"]
    r = subprocess.run([sys.executable, os.path.join(PKG, "train_ours.py"), "-c", str(cfg_path), "-id", "t2",
                        "--resume", str(ckpt), "--iterations", "5"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "Iteration: 4/5" in r.stdout, r.stderr[-2000:] + r.stdout[-500:]
    r = subprocess.run([sys.executable, os.path.join(PKG, "infer_ours.py"), "--model_path", str(ckpt), "--batch", "2",
                        "--height", "64", "--width", "64", "--num_ts", "3"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "interpolated 6 frames" in r.stdout, r.stderr[-2000:]
