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


def test_bench_two_rank_rehearsal(tmp_path):
    """bench.py's multi-rank path (barrier, flat-bucket gradient all-reduce, max-over-ranks timing, rank-0 JSON)
    rehearsed with 2 ranks sharing the one GPU over gloo (EBFI_BENCH_REHEARSAL=1); the real N>1 runs use RCCL."""
    import json
    detail = str(tmp_path / "detail.json")
    env = dict(os.environ, EBFI_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-bf16-leg", "--detail", detail],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["cpu_baseline"] is None and d["roofline"]["kernel"].startswith("conv_")
    assert d["config"]["world_size"] == 2 and d["config"]["collective_backend"] == "gloo"
    assert len(lines[0]) < 4096                     # the line the driver parses stays small; the tables are in the detail file
    full = json.load(open(detail))
    assert full["config"]["replica_param_checksum"]["ranks_identical"] is True
    # every launch of the profiled pass was timed: per-step launch counts are whole numbers
    for name, k in full["kernels"].items():
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


def test_bench_plain_form_launches_its_own_ranks():
    """`python bench.py --gpus 2` -- the form the driver uses, no launcher, no WORLD_SIZE -- starts its two ranks as a child
    process (torch.distributed.run), relays rank 0's one JSON line and the exit code.  Two ranks share the one GPU of this
    box over gloo (EBFI_BENCH_REHEARSAL=1); strict graph capture is the default for N > 1 and must hold."""
    import json
    env = dict(os.environ, EBFI_BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-extra-legs"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and d["config"]["global_batch"] == 16
    assert d["config"]["collectives_per_step"] == 1 and d["config"]["graph_capture_failed"] is False
    assert "hipGraph replay" in d["config"]["launch"]
    # a mismatch between --gpus and the launcher's world size is an error, not a silent 1-rank run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr + r.stdout


def test_bench_form_stays_native_under_strict(tmp_path):
    """The driver's bench form (`python bench.py --gpus 1 ...`: default widths, the split-precision leg with the fp16 backward, the
    exact-fp32 leg, the three inference legs) with EBFI_STRICT_NATIVE=1: a convolution of the default model dispatched to torch /
    MIOpen anywhere on these paths raises EbfiNativeError and fails the run (round 5: ResidualControl's scalar 1x1 convolutions
    still printed 'runs on torch' here).  Also what the driver needs from stdout: ONE line, < 4 KB, with roofline and config."""
    import json
    env = dict(os.environ, EBFI_STRICT_NATIVE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "EBFI_BENCH_REHEARSAL", "EBFI_BENCH_FORCE_DIST"):
        env.pop(k, None)
    detail = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-ops", "--detail", detail], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "runs on torch" not in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["roofline"]["kernel"].startswith("conv_") and d["config"]["graph_capture_failed"] is False
    assert set(d["inference_frames_per_s"]) == {"config2_fp32", "config2_bf16x3", "config5_hd_bf16x3"}
    full = json.load(open(detail))
    assert full["inference"]["config5_hd_bf16x3"]["kernelconv_fac_fused"]["kernel"] == "conv_fwd_f16_ws/kernelconv_fac_img"
    assert full["fp32_exact_mode"]["ms_per_step"] > d["ms_per_step"]


def test_bench_single_rank_over_rccl():
    """What a one-GPU box can exercise of the RCCL path: bench.py as ONE rank with the process group initialised on backend
    'nccl' (= RCCL), its watchdog thread alive during the hipGraph capture (strict), and the step's collective -- the
    all-reduce of the wire buffer, gradients + overflow flag -- going through RCCL at world size 1 (EBFI_BENCH_FORCE_DIST=1)."""
    import json
    env = dict(os.environ, EBFI_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29667", WORLD_SIZE="1", RANK="0",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("EBFI_BENCH_REHEARSAL", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--no-extra-legs", "--no-ops", "--no-inference", "--no-cpu-baseline", "--strict-graph"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["collective_backend"] == "nccl" and d["config"]["collectives_per_step"] == 1
    assert "hipGraph replay" in d["config"]["launch"] and d["config"]["graph_capture_failed"] is False
    assert d["config"]["fp16_steps_skipped"] == 0


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
    assert d["n_gpus"] == 2 and d["config"]["collective_backend"] == "nccl"      # (bench.py exits non-zero if replicas diverge)
    assert "hipGraph replay" in d["config"]["launch"]


def test_hoisted_inference_is_bit_identical():
    """ClipInterpolator (what infer_ours.py runs): the timestamp-independent prefix computed once per clip + the per-timestamp
    rest replayed from a hipGraph gives, bit for bit, what calling the model once per timestamp gives (reference loop
    infer_ours.py:113-118) -- in both conv modes, with and without the graph, on a size that needs the pad / crop path."""
    from ebfi_amd import conv
    from ebfi_amd.engine import ClipInterpolator, synthetic_batch
    from ebfi_amd.model import EVFIAutoEx
    from ebfi_amd.engine import DEFAULT_MODEL_ARGS
    torch.manual_seed(5)
    cfg = dict(DEFAULT_MODEL_ARGS, step=2, channels=[8, 8, 16, 16])
    model = EVFIAutoEx(**cfg).cuda().eval()
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() > 1:
                p.copy_(torch.randn_like(p) * (1.2 / p[0].numel() ** 0.5))
    stamps = [0.0, 0.25, 0.8125]
    for (h, w) in ((64, 64), (60, 76)):
        frame, event, _, gtex, _ = synthetic_batch(2, h, w, 16, device="cuda", seed=3)
        for prec in ("bf16x3", "fp32"):
            plain = ClipInterpolator(model, precision=prec, graph=False, hoist=False)
            ref = plain(frame, event, gtex, stamps)
            assert ref.shape == (2, 3, 3, h, w) and ref.std() > 1e-3
            # the un-hoisted eager path IS the module call of the reference loop
            conv.set_compute_dtype(prec)
            try:
                with torch.no_grad(), (plain.bank.active() if plain.bank is not None else torch.no_grad()):
                    direct = model(frame, event, torch.full((2, 1), stamps[1], device="cuda"), gtex)[-1]
            finally:
                conv.set_compute_dtype("fp32")
            assert torch.equal(direct, ref[:, 1])
            for graph in (False, True):
                got = ClipInterpolator(model, precision=prec, graph=graph, hoist=True, group=1)(frame, event, gtex, stamps)
                assert torch.equal(got, ref), (prec, graph, (got - ref).abs().max().item())
                # several timestamps per pass (round 6; the default picks as many as fit a pixel budget: all three here, and a group
                # of two with a short last pass): the same frames to fp32 rounding -- the kernels choose tile geometries by
                # problem size, so bit-equality is only promised for group=1
                for group in (None, 2):
                    interp = ClipInterpolator(model, precision=prec, graph=graph, hoist=True, group=group)
                    got = interp(frame, event, gtex, stamps)
                    assert interp.last_group == (3 if group is None else 2)
                    assert got.shape == ref.shape and ((got - ref).abs().max() / ref.abs().max()).item() < 1e-5, (prec, graph, group)
    # a second clip through the SAME captured graph picks up its own prefix
    interp = ClipInterpolator(model, precision="bf16x3", graph=True, hoist=True)
    a = synthetic_batch(2, 64, 64, 16, device="cuda", seed=3)
    b = synthetic_batch(2, 64, 64, 16, device="cuda", seed=4)
    ra, rb, ra2 = (interp(v[0], v[1], v[3], stamps) for v in (a, b, a))
    assert torch.equal(ra, ra2) and not torch.equal(ra, rb) and len(interp._captured) == 1


def test_overflow_in_first_micro_step_of_a_graph_window_skips_the_update():
    """Gradient accumulation + hipGraph replay + fp16 backward (round-3 advisory): the overflow guard must stay raised over the
    whole accumulation window.  The clear used to sit inside the captured region, so the replay of micro-step 1 wiped the
    flag micro-step 0 had raised and Adam applied saturated gradients.  Now the window's first call clears it eagerly: an
    overflow forced in micro-step 0 (event counts scaled by 1e6: the staged activations leave their one-step-old scales) skips
    the window's update -- parameters, moments, step count untouched, the skip counted -- and the next window goes through."""
    from ebfi_amd.engine import Engine, synthetic_batch
    cfg = dict(step=2, channels=[8, 8, 16, 16])
    eng = Engine(cfg, device="cuda", seed=7, graph=True, precision="bf16x3", accu_step=2)
    with torch.no_grad():
        for p in eng.model.parameters():
            if p.dim() > 1:
                p.copy_(torch.randn_like(p) * (1.2 / p[0].numel() ** 0.5))
    mk = lambda s: list(synthetic_batch(2, 64, 64, device="cuda", seed=s))
    for it in range(8):                                   # 4 windows: 2 calibration steps (eager), capture, one replayed window
        eng.train_step(*mk(100 + it))
    assert eng.iteration == 4 and len(eng._graphs) == 1 and eng.book.skipped_steps() == 0
    st = eng.optimizer.inner.state[eng.optimizer.flat]
    before = (eng.optimizer.flat.detach().clone(), st["exp_avg"].clone(), float(st["step"]))
    bad = mk(200)
    bad[1] = bad[1] * 1e6                                 # event counts x1e6 in micro-step 0: every activation the weight
    #                                                       gradients stage outgrows its one-step-old fp16 scale by 2^20
    eng.train_step(*bad)
    assert int(eng.book.guard[0].item()) == 1             # raised by micro-step 0 ...
    eng.train_step(*mk(201))                              # ... micro-step 1 is clean and replays the same graph
    assert eng.book.skipped_steps() == 1 and eng.iteration == 5
    assert torch.equal(before[0], eng.optimizer.flat.detach()) and torch.equal(before[1], st["exp_avg"])
    assert float(st["step"]) == before[2]
    for it in range(4):                                   # the scales have followed / recovered: the next windows update again
        eng.train_step(*mk(300 + it))
    assert eng.book.skipped_steps() <= 2 and not torch.equal(before[0], eng.optimizer.flat.detach())
    assert torch.isfinite(eng.optimizer.flat).all()


@pytest.mark.parametrize("native", [True, False])
def test_guarded_adam_skip_leaves_state_untouched(native, monkeypatch):
    """guard[0] != 0: parameters, both moments and the step counter stay as they are and guard[1] counts the skip -- in the
    native launch (ebfi_adam_step_guarded) and in the host-side check in front of torch's own step (the path taken for
    options the native kernel does not cover)."""
    from ebfi_amd.dp import FlatAdam
    torch.manual_seed(1)
    params = [torch.nn.Parameter(torch.randn(300, 7).cuda()), torch.nn.Parameter(torch.randn(11).cuda())]
    opt = FlatAdam(params, lr=1e-2)
    if not native:
        monkeypatch.setattr(FlatAdam, "_native_step", lambda self, g, guard=None, flag=None: False)
    guard = torch.zeros(2, dtype=torch.int32, device="cuda")
    g = torch.randn(opt.flat.numel(), device="cuda")
    opt.step(g, guard=guard)                              # a clean step first (creates the state)
    st = opt.inner.state[opt.flat]
    snap = (opt.flat.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), float(st["step"]))
    guard[0] = 1
    opt.step(torch.full_like(g, 65504.0), guard=guard)
    torch.cuda.synchronize()
    assert torch.equal(snap[0], opt.flat.detach()) and torch.equal(snap[1], st["exp_avg"]) and torch.equal(snap[2], st["exp_avg_sq"])
    assert float(st["step"]) == snap[3] == 1.0 and guard.tolist() == [1, 1]
    guard[0] = 0
    opt.step(g, guard=guard)
    assert float(st["step"]) == 2.0 and guard.tolist() == [0, 1] and not torch.equal(snap[0], opt.flat.detach())
