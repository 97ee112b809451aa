"""The training step as one hipGraph launch (voxelmorph_amd/graph.py) and the multi-rank step with the HIP path computing.

* a replayed step changes no bit of what the eager step produces (weights after N steps, Adam state, device step counter);
* the two-rank data-parallel step (scripts/torch/train.py:151-154 replaced by one all-reduce of the flat bucket) run as two processes
  that SHARE the one GPU of the box: both ranks end on the same weights, and those are the weights of a one-process step on the
  concatenated batch (the oracle plays no part: the HIP kernels compute on every rank);
* `bench.py --gpus 2` on the shared device: the line reports two ranks and names its exchange.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vxm():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device")
    import voxelmorph_amd
    from voxelmorph_amd import _lib
    _lib.lib()
    return voxelmorph_amd


def _setup(vxm, shape, B, seed=3, int_steps=7, lr=1e-3):
    from voxelmorph_amd.optim import FlatAdam
    torch.manual_seed(seed)
    model = vxm.networks.VxmDense(shape, int_steps=int_steps, int_downsize=2).cuda()
    with torch.no_grad():
        model.flow.weight.mul_(2e4)                 # a field of ~0.2 voxels: the gather kernels see real displacements
    opt = FlatAdam(model, lr=lr)
    g = torch.Generator(device="cuda").manual_seed(seed + 1)
    src = torch.rand((B, 1) + shape, device="cuda", generator=g)
    trg = torch.rand((B, 1) + shape, device="cuda", generator=g)
    ncc, reg = vxm.losses.NCC().loss, vxm.losses.Grad("l2", loss_mult=2).loss

    def fwd():
        y, pre = model(src, trg)
        return ncc(trg, y) + reg(None, pre)
    return model, opt, fwd, (src, trg)


@pytest.mark.parametrize("shape,B", [((32, 32, 32), 1), ((48, 64, 32), 2)])
def test_graph_replay_changes_no_bit_of_the_training(vxm, shape, B):
    from voxelmorph_amd.graph import GraphedStep
    steps = 6
    _, opt_e, fwd_e, _ = _setup(vxm, shape, B)
    eager = GraphedStep(fwd_e, opt_e, enabled=False)
    losses_e = [float(eager()) for _ in range(steps)]
    _, opt_g, fwd_g, _ = _setup(vxm, shape, B)
    graphed = GraphedStep(fwd_g, opt_g, eager_steps=2)
    losses_g = [float(graphed()) for _ in range(steps)]
    torch.cuda.synchronize()
    assert graphed.replays == steps - 2 and graphed.graph is not None
    assert opt_e.step_count == steps and opt_g.step_count == steps
    assert torch.equal(opt_e.flat_param, opt_g.flat_param), float((opt_e.flat_param - opt_g.flat_param).abs().max())
    assert torch.equal(opt_e.exp_avg, opt_g.exp_avg) and torch.equal(opt_e.exp_avg_sq, opt_g.exp_avg_sq)
    assert max(abs(a - b) for a, b in zip(losses_e, losses_g)) < 1e-6, (losses_e, losses_g)
    assert losses_e[-1] != losses_e[0]              # it trains


def test_graph_follows_a_learning_rate_schedule(vxm):
    """lr / betas / eps are by-value arguments of the captured Adam launch: GraphedStep re-captures when a schedule moves them, so a replayed
    run equals the eager run of the same schedule bit for bit"""
    from voxelmorph_amd.graph import GraphedStep
    shape = (32, 32, 32)
    runs = []
    for enabled in (False, True):
        _, opt, fwd, _ = _setup(vxm, shape, 1)
        step = GraphedStep(fwd, opt, eager_steps=1, enabled=enabled)
        for k in range(6):
            if k == 3:
                opt.lr = 1e-4
            step()
        torch.cuda.synchronize()
        runs.append((opt.flat_param.clone(), step))
    assert runs[1][1].recaptures == 1 and runs[1][1].replays == 5
    assert torch.equal(runs[0][0], runs[1][0]), float((runs[0][0] - runs[1][0]).abs().max())


def test_graph_sees_refilled_inputs_and_eager_code_after_replay(vxm):
    """inputs are static tensors refilled in place; an eager forward after replays must use the weights the graph left (the packed
    operators cached per parameter are re-derived: the graph bumps the parameters' version counters)"""
    from voxelmorph_amd.graph import GraphedStep
    shape = (32, 32, 32)
    model, opt, fwd, (src, trg) = _setup(vxm, shape, 1)
    step = GraphedStep(fwd, opt, eager_steps=1)
    for _ in range(3):
        step()
    l_same = float(step())
    src.copy_(torch.rand_like(src))
    l_new = float(step())
    assert abs(l_new - l_same) > 1e-4                # the replay read the refilled tensor
    with torch.no_grad():
        l_eager = float(fwd())                       # eager forward on the post-replay weights ...
    model2, opt2, fwd2, (s2, t2) = _setup(vxm, shape, 1)
    opt2.flat_param.copy_(opt.flat_param)
    opt2._params_rewritten()
    s2.copy_(src)
    with torch.no_grad():
        l_fresh = float(fwd2())                      # ... equals a fresh model loaded with those weights
    assert abs(l_eager - l_fresh) < 1e-6, (l_eager, l_fresh)


def test_adam_device_counter_matches_host_counter(vxm):
    from voxelmorph_amd.optim import FlatAdam
    torch.manual_seed(0)
    m1 = vxm.networks.VxmDense((16, 16, 16), int_steps=1).cuda()
    m2 = vxm.networks.VxmDense((16, 16, 16), int_steps=1).cuda()
    m2.load_state_dict(m1.state_dict())
    o1, o2 = FlatAdam(m1, lr=1e-3, capturable=True), FlatAdam(m2, lr=1e-3, capturable=False)
    for k in range(5):
        g = torch.randn(o1.n, device="cuda")
        for o in (o1, o2):
            o.zero_grad()
            o.flat_grad.copy_(g)
            o.step()
    torch.cuda.synchronize()
    assert o1.step_count == o2.step_count == 5
    assert float((o1.flat_param - o2.flat_param).abs().max()) <= 1e-9      # bias corrections: device pow() against the host's, rounded to fp32
    o1.step_count = 100
    assert o1.step_count == 100


_WORKER = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import torch
import voxelmorph_amd as vxm
from voxelmorph_amd import dist as vdist
from voxelmorph_amd.optim import FlatAdam
from voxelmorph_amd.graph import GraphedStep
rank, local, world = vdist.init_from_env()
assert world == 2 and torch.cuda.current_device() == 0
shape, steps = (32, 48, 32), 4
torch.manual_seed(7)
g = torch.Generator(device="cuda").manual_seed(11)
SRC = torch.rand((2, 1) + shape, device="cuda", generator=g)        # the global batch: rank r owns pair r
TRG = torch.rand((2, 1) + shape, device="cuda", generator=g)
ncc, reg = vxm.losses.NCC().loss, vxm.losses.Grad("l2", loss_mult=2).loss

def run(lo, hi, comm_world, graphed, direct=True):
    torch.manual_seed(5)
    model = vxm.networks.VxmDense(shape, int_steps=7, int_downsize=2).cuda()
    with torch.no_grad():
        model.flow.weight.mul_(2e4)
    opt = FlatAdam(model, lr=1e-3, direct_grads=direct)
    if comm_world == 1:
        opt.world, opt.group = 1, None
    opt.broadcast_params(0)
    src, trg = SRC[lo:hi].clone(), TRG[lo:hi].clone()
    def fwd():
        y, pre = model(src, trg)
        return ncc(trg, y) + reg(None, pre)
    step = GraphedStep(fwd, opt, eager_steps=1, enabled=graphed)
    grads = []
    for _ in range(steps):
        step()
        grads.append(opt.flat_grad.clone())          # after the all-reduce: the summed gradient
    torch.cuda.synchronize()
    return opt.flat_param.clone(), grads, step.replays

out = {}
for graphed in (False, True):
    p2, g2, replays = run(rank, rank + 1, 2, graphed)                 # two ranks, one pair each, the all-reduce between them
    other = p2.cpu()
    torch.distributed.broadcast(other, 1)                             # rank 1's weights
    out["ranks_equal_%%d" %% graphed] = bool(torch.equal(other, p2.cpu()))
    out["replays_%%d" %% graphed] = replays
    if rank == 0:
        p1, g1, _ = run(0, 2, 1, False)                               # one process, both pairs (loss = mean over the batch)
        out["weights_vs_one_process_%%d" %% graphed] = float((p2 - p1).abs().max())
        out["first_grad_rel_%%d" %% graphed] = float((0.5 * g2[0] - g1[0]).norm() / g1[0].norm())
    vdist.barrier()
# gradients that reach the optimiser through autograd's p.grad instead of the flat-bucket sink (FlatAdam(direct_grads=False); the 2-D network and
# a parameter used twice take the same route): the replayed multi-rank step must keep handing them to opt.step() (ADVICE round 5)
pe, _, _ = run(rank, rank + 1, 2, False, direct=False)
pg, _, rp = run(rank, rank + 1, 2, True, direct=False)
pd, _, _ = run(rank, rank + 1, 2, False, direct=True)
out["autograd_grads_graph_vs_eager"] = float((pe - pg).abs().max())
out["autograd_grads_vs_sink"] = float((pe - pd).abs().max())
out["autograd_replays"] = rp
vdist.barrier()
if rank == 0:
    print("RESULT " + json.dumps(out))
"""


def test_two_rank_step_on_the_shared_device_equals_one_process_on_the_concatenated_batch(vxm, tmp_path):
    script = os.path.join(tmp_path, "dp_worker.py")
    with open(script, "w") as f:
        f.write(_WORKER % dict(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VXM_SHARE_DEVICE="1", VXM_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29561", script], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    print(res)
    for graphed in (0, 1):
        assert res["ranks_equal_%d" % graphed]
        assert res["first_grad_rel_%d" % graphed] < 1e-5
        # 4 Adam steps of 1e-3: the update m / sqrt(v) of a parameter whose gradient is ~0 amplifies the 5e-8 summation-order difference of
        # the gradients (measured 2.1e-5 = 2 % of one step); a wrong average (sum instead of mean, a missing rank) would show as ~1e-3
        assert res["weights_vs_one_process_%d" % graphed] < 1e-4
    assert res["replays_0"] == 0 and res["replays_1"] == 3
    # a replay that dropped the p.grad gradients would run Adam on zeros from the second replay on: ~1e-3 per step
    assert res["autograd_replays"] == 3 and res["autograd_grads_graph_vs_eager"] < 1e-6 and res["autograd_grads_vs_sink"] < 1e-5, res


def test_bench_two_ranks_on_the_shared_device(vxm):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VXM_SHARE_DEVICE="1", VXM_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "3", "--shape", "32,48,32",
                          "--no-extra-configs", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 2 and line["scaling"] == "weak"
    assert line["comm"]["ranks_seen"] == 2 and "gloo" in line["comm"]["backend"]
    assert line["submission"]["graph_replays"] == 3 and line["submission"]["device_allocs"] == 0
    assert line["value"] > 0


def test_range_report_separates_heavy_tails_from_ordinary_batches(vxm):
    """voxelmorph_amd.diagnostics: the share of values that fall below 2^-18 of their staged tile's largest magnitude (where the fp16-piece
    engine keeps an absolute instead of a relative error bound) is ~0 for an ordinary batch -- noise pairs, the bench's workload -- and
    large for the heavy-tailed one of test_full_size_step_with_heavy_tailed_activations_on_all_three_engines; the guard of GraphedStep moves
    the process to the three-piece engine for the latter only."""
    import warnings
    from voxelmorph_amd import diagnostics
    from voxelmorph_amd.graph import GraphedStep
    from voxelmorph_amd.torch import functional as VF
    if VF.FP32_ENGINE != "f16x2":
        pytest.skip("the guard acts on the fp16-piece engine")
    shape = (64, 96, 128)
    model, opt, fwd, (src, trg) = _setup(vxm, shape, 1, seed=5)

    def step():
        opt.zero_grad()
        fwd().backward()
    rep = diagnostics.range_report(step)
    print("noise pair: worst", rep["worst"])
    assert rep["recommended_engine"] == "f16x2" and len(rep["tensors"]) >= 20
    g = torch.Generator(device="cuda").manual_seed(9)
    for img in (src, trg):
        idx = torch.randint(0, img.numel(), (img.numel() // 1000,), device="cuda", generator=g)
        img.view(-1)[idx] *= torch.exp2(12.0 + 8.0 * torch.rand(idx.numel(), device="cuda", generator=g))
    rep2 = diagnostics.range_report(step)
    print("heavy tails: worst", rep2["worst"])
    assert rep2["recommended_engine"] == "split"
    keep = VF.FP32_ENGINE
    try:
        gs = GraphedStep(fwd, opt, eager_steps=1, range_guard=True)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            gs()
        assert VF.FP32_ENGINE == "split" and any("three-piece" in str(w.message) for w in caught)
        gs()                                   # captured and replayed on the engine the guard chose
        gs()
        assert gs.replays == 2
    finally:
        VF.FP32_ENGINE = keep


def test_refused_capture_falls_back_to_launch_by_launch(vxm, monkeypatch):
    """a hipGraph capture the runtime refuses must not take the training down: GraphedStep warns, drops the operators "packed" inside the
    failed capture (they were never executed) and goes on launch by launch -- with the same weights as a run that never tried"""
    import warnings
    from voxelmorph_amd.graph import GraphedStep
    shape, steps = (32, 32, 32), 5
    _, opt_e, fwd_e, _ = _setup(vxm, shape, 1)
    eager = GraphedStep(fwd_e, opt_e, enabled=False)
    for _ in range(steps):
        eager()
    _, opt_g, fwd_g, _ = _setup(vxm, shape, 1)
    step = GraphedStep(fwd_g, opt_g, eager_steps=2)
    real_zero = opt_g.zero_grad

    def refusing_zero_grad():
        if torch.cuda.is_current_stream_capturing():          # something inside the captured region that the runtime does not take
            raise RuntimeError("capture refused (test)")
        real_zero()
    monkeypatch.setattr(opt_g, "zero_grad", refusing_zero_grad)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for _ in range(steps):
            step()
    torch.cuda.synchronize()
    assert step.graph is None and step.replays == 0 and "capture refused" in step.capture_error
    assert any("launch-by-launch" in str(w.message) for w in caught)
    assert opt_g.step_count == steps and torch.equal(opt_e.flat_param, opt_g.flat_param)
