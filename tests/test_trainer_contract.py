"""Trainer contract on CPU (no kernels run): checkpoint key set = the reference's (train_ours.py:628-655), resume at
trainer.iteration + 1 with optimiser / scheduler / learning rate restored (:673-716, myutils/utils.py:178-215), StepLR
stepped under the reference's gate (:335-338), gradient accumulation over accu_step passes (:259-277), and FlatAdam
loading a torch.optim.Adam checkpoint that lacks entries and carries foreign run-mode flags."""
import copy
import importlib.util
import os

import pytest
import torch
import yaml

from ebfi_amd.dp import FlatAdam, FlatGradBucket
from ebfi_amd.engine import Engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ebfi-be_amd")

SMALL = dict(FrameBasech=8, EventBasech=8, InterCH=8, TB=4, step=2, channels=[4, 4, 8, 8])


def _trainer():
    spec = importlib.util.spec_from_file_location("ebfi_train_ours", os.path.join(PKG, "train_ours.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _config():
    cfg = yaml.safe_load(open(os.path.join(PKG, "config", "train_ours.yml")))
    cfg["model"]["args"].update(SMALL)
    return cfg


def _fake_steps(eng, n, seed=0):
    g = torch.Generator().manual_seed(seed)
    for _ in range(n):
        eng.optimizer.step(torch.randn(eng.optimizer.flat.numel(), generator=g))
        eng.iteration += 1


def test_default_config_names_the_reference_schedule():
    T = _trainer()
    cfg = yaml.safe_load(open(os.path.join(PKG, "config", "train_ours.yml")))
    st = T.trainer_settings(cfg)
    assert cfg["lr_scheduler"] == {"name": "StepLR", "args": {"step_size": 2e5, "gamma": 0.5}}
    assert st["lr_min"] == 1e-6 and st["accu_step"] == 1 and st["lr_change_rate"] == 1
    assert T.trainer_settings(cfg, cli_iterations=7)["iterations"] == 7
    with pytest.raises(ValueError):
        T.build_lr_scheduler({"lr_scheduler": {"name": "NoSuchLR"}}, torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))]))


def test_checkpoint_has_exactly_the_reference_keys_and_resumes(tmp_path):
    T = _trainer()
    cfg = _config()
    cfg["lr_scheduler"]["args"].update(step_size=2.0, gamma=0.5)
    eng = Engine(cfg["model"]["args"], device="cpu", lr=1e-3, seed=1)
    sched = T.build_lr_scheduler(cfg, eng.optimizer.inner)
    _fake_steps(eng, 5)
    for _ in range(5):
        sched.step()                                       # lr 1e-3 -> 2.5e-4 after two decays
    path = str(tmp_path / "checkpoint-iteration4.pth")
    T.save_checkpoint(path, eng, sched, cfg, 4)
    cpt = torch.load(path, map_location="cpu", weights_only=False)
    assert set(cpt) == set(T.CHECKPOINT_KEYS) == {"model", "lr_scheduler", "optimizer", "config", "trainer"}
    assert set(cpt["model"]) == set(cpt["optimizer"]) == set(cpt["lr_scheduler"]) == {"name", "states"}
    assert cpt["lr_scheduler"]["name"] == "StepLR" and cpt["optimizer"]["name"] == "Adam" and cpt["model"]["name"] == "EVFIAutoEx"
    assert cpt["trainer"] == {"training_mode": "iteration_based_train", "iteration": 4, "monitor_best": None}
    # the reference's resume path reads exactly these (myutils/utils.py:186-209): a torch scheduler / optimiser accept them
    probe = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in eng.optimizer.params], lr=1e-3)
    probe.load_state_dict(cpt["optimizer"]["states"])
    torch.optim.lr_scheduler.StepLR(probe, step_size=2, gamma=0.5).load_state_dict(cpt["lr_scheduler"]["states"])

    eng2 = Engine(cfg["model"]["args"], device="cpu", lr=1e-3, seed=2)
    sched2 = T.build_lr_scheduler(cfg, eng2.optimizer.inner)
    start = T.resume_checkpoint(path, eng2, sched2, cfg)
    assert start == 5 and eng2.iteration == 5              # trainer.iteration + 1 (train_ours.py:695)
    assert sched2.get_last_lr() == sched.get_last_lr() == [pytest.approx(2.5e-4)]
    assert eng2.optimizer.param_groups[0]["lr"] == pytest.approx(2.5e-4)
    for a, b in zip(eng.model.state_dict().values(), eng2.model.state_dict().values()):
        assert torch.equal(a, b)
    sa, sb = eng.optimizer.inner.state[eng.optimizer.flat], eng2.optimizer.inner.state[eng2.optimizer.flat]
    assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]) and float(sb["step"]) == 5
    assert eng2.optimizer.param_groups[0]["fused"] == eng.optimizer.param_groups[0]["fused"]

    eng3 = Engine(cfg["model"]["args"], device="cpu", lr=1e-3, seed=3)     # --reset: model only
    sched3 = T.build_lr_scheduler(cfg, eng3.optimizer.inner)
    assert T.resume_checkpoint(path, eng3, sched3, cfg, reset=True) == 0
    assert sched3.get_last_lr() == [1e-3] and not eng3.optimizer.inner.state
    assert torch.equal(next(iter(eng3.model.state_dict().values())), next(iter(eng.model.state_dict().values())))


def test_steplr_gate_of_the_reference_loop():
    """step() when it % lr_change_rate == 0 and it != 0 and lr >= lr_min (train_ours.py:335-338)."""
    T = _trainer()
    cfg = _config()
    cfg["lr_scheduler"]["args"].update(step_size=3.0, gamma=0.1)
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(2))], lr=1e-4)
    sched = T.build_lr_scheduler(cfg, opt)
    st = dict(T.trainer_settings(cfg), lr_min=1e-6)
    lrs = []
    for it in range(14):
        lrs.append(sched.get_last_lr()[0])
        if it % st["lr_change_rate"] == 0 and it != 0 and sched.get_last_lr()[0] >= st["lr_min"]:
            sched.step()
    # iteration 0 never steps: the first decay lands after 3 counted steps (its 1..3); below lr_min the schedule freezes
    assert lrs[:4] == pytest.approx([1e-4] * 4) and lrs[4] == pytest.approx(1e-5) and lrs[7] == pytest.approx(1e-6)
    assert lrs[10] == pytest.approx(1e-7) and lrs[13] == pytest.approx(1e-7)


def test_accumulation_window_sums_then_steps_once():
    eng = Engine(SMALL, device="cpu", lr=1e-2, seed=5, accu_step=3)
    n = eng.optimizer.flat.numel()
    before = eng.optimizer.flat.detach().clone()
    grads = [torch.randn(n, generator=torch.Generator().manual_seed(s)) for s in range(3)]
    took = [eng._finish_micro_step(g.clone()) for g in grads]
    assert took == [False, False, True] and eng.iteration == 1
    ref = torch.nn.Parameter(before.clone())
    opt = torch.optim.Adam([ref], lr=1e-2)
    ref.grad = grads[0] + grads[1] + grads[2]
    opt.step()
    assert torch.allclose(eng.optimizer.flat.detach(), ref.detach(), rtol=1e-6, atol=1e-8)
    assert not eng._finish_micro_step(grads[0].clone()) and eng.iteration == 1      # next window has started


def test_flat_adam_loads_partial_torch_state_and_keeps_its_own_run_flags():
    torch.manual_seed(4)
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 3, 3), torch.nn.Conv2d(3, 2, 1))
    ref_opt = torch.optim.Adam(net.parameters(), lr=5e-3)
    x = torch.randn(1, 2, 6, 6)
    for _ in range(2):
        ref_opt.zero_grad()
        net(x).square().sum().backward()
        ref_opt.step()
    sd = copy.deepcopy(ref_opt.state_dict())
    del sd["state"][3]                                     # a parameter that never received a gradient
    assert sd["param_groups"][0]["fused"] is None and sd["param_groups"][0]["foreach"] is None
    net2 = copy.deepcopy(net)
    opt = FlatAdam(list(net2.parameters()), lr=1e-2)
    flags = {k: opt.param_groups[0][k] for k in ("fused", "foreach", "capturable")}
    opt.load_state_dict(sd)
    assert {k: opt.param_groups[0][k] for k in flags} == flags and opt.param_groups[0]["lr"] == 5e-3
    st = opt.inner.state[opt.flat]
    assert float(st["step"]) == 2
    off = opt._offsets[3]
    assert st["exp_avg"][off:].abs().sum() == 0 and st["exp_avg"][:off].abs().sum() > 0
    bucket = FlatGradBucket(net2)
    net2(x).square().sum().backward()
    opt.step(bucket.gather())                              # usable right away
    assert torch.isfinite(opt.flat).all()
