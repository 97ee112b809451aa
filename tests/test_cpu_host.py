"""CPU-side tests: the C-ABI library loads and exports every declared symbol, the host mirror keeps the
reference's API surface / checkpoint format, the product refuses to run without a HIP device, and the
data-parallel path is correct by construction (world_size-2 gloo)."""
import inspect
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import voxelmorph_amd as vxm                      # noqa: E402
from voxelmorph_amd import _lib                   # noqa: E402
from oracle import ref_loader                     # noqa: E402
from oracle import vxm_oracle as orc              # noqa: E402


@pytest.fixture(scope="session", autouse=True)
def built_library():
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["bash", os.path.join(ROOT, "voxelmorph_amd", "csrc", "build.sh")])


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "vxm_hip.h")).read()
    declared = set(re.findall(r"\b(vxm_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (vxm_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    h = _lib.lib()                                  # ctypes load + argtypes for every entry point
    assert h.vxm_version() >= 100
    assert h.vxm_conv3d_k3_packed_elems(48, 32) == 6 * 27 * 2 * 2 * 64
    assert h.vxm_conv3d_k3_bwd_weight_workspace_bytes(64, 32, 1, 8, 8, 16) > 0


def test_comm_library_exports_declared_symbols():
    """include/vxm_comm.h <-> libvxm_comm.so <-> voxelmorph_amd/comm.py; no collective is issued without a GPU."""
    from voxelmorph_amd import comm
    header = open(os.path.join(ROOT, "include", "vxm_comm.h")).read()
    declared = set(re.findall(r"\b(vxm_[a-z0-9_]+)\s*\(", header))
    assert declared == set(comm.SIGNATURES), declared ^ set(comm.SIGNATURES)
    out = subprocess.check_output(["nm", "-D", "--defined-only", comm.LIB_PATH]).decode()
    assert declared <= set(re.findall(r" T (vxm_[a-z0-9_]+)", out))
    h = comm.lib()
    assert h.vxm_comm_world() == 0
    assert h.vxm_allreduce_sum_f32(None, 4, None) != 0 and b"not initialised" in h.vxm_comm_last_error_string()
    assert h.vxm_comm_destroy() == 0
    with pytest.raises(ValueError):
        comm.NativeComm(0, 1, b"short")


def test_kernels_are_gfx950_only():
    blob = open(_lib.LIB_PATH, "rb").read()
    archs = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert archs == {b"gfx950"}, archs


def test_no_cpu_fallback():
    st = vxm.layers.SpatialTransformer((8, 8, 8))
    with pytest.raises(_lib.VxmHipError, match="no CPU fallback"):
        st(torch.zeros(1, 1, 8, 8, 8), torch.zeros(1, 3, 8, 8, 8))
    with pytest.raises(_lib.VxmHipError):
        vxm.losses.MSE().loss(torch.zeros(4), torch.zeros(4))
    model = vxm.networks.VxmDense((16, 16, 16))
    with pytest.raises(_lib.VxmHipError):
        model(torch.zeros(1, 1, 16, 16, 16), torch.zeros(1, 1, 16, 16, 16))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "voxelmorph_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("oracle/", "").lower() or f == "_none_", (dirpath, f)


def test_api_surface_matches_reference_signatures():
    sig = inspect.signature(vxm.networks.VxmDense.__init__)
    assert list(sig.parameters)[1:] == ["inshape", "nb_unet_features", "nb_unet_levels", "unet_feat_mult",
                                        "nb_unet_conv_per_level", "int_steps", "int_downsize", "bidir", "use_probs",
                                        "src_feats", "trg_feats", "unet_half_res"]
    assert sig.parameters["int_steps"].default == 7 and sig.parameters["int_downsize"].default == 2
    assert list(inspect.signature(vxm.layers.SpatialTransformer.__init__).parameters)[1:] == ["size", "mode"]
    assert list(inspect.signature(vxm.layers.VecInt.__init__).parameters)[1:] == ["inshape", "nsteps"]
    assert list(inspect.signature(vxm.layers.ResizeTransform.__init__).parameters)[1:] == ["vel_resize", "ndims"]
    assert list(inspect.signature(vxm.networks.Unet.__init__).parameters)[1:] == [
        "inshape", "infeats", "nb_features", "nb_levels", "max_pool", "feat_mult", "nb_conv_per_level", "half_res"]
    assert list(inspect.signature(vxm.losses.Grad.__init__).parameters)[1:] == ["penalty", "loss_mult"]
    m = vxm.networks.VxmDense((32, 32, 32))
    for attr in ("unet_model", "flow", "resize", "fullsize", "bidir", "integrate", "transformer", "config"):
        assert hasattr(m, attr)
    assert m.integrate.nsteps == 7 and m.integrate.scale == 1 / 128 and m.resize.factor == 0.5 and m.fullsize.factor == 2
    assert m.resize.mode == "trilinear" and m.transformer.grid.dtype == torch.float32
    assert m.transformer.grid.shape == (1, 3, 32, 32, 32) and float(m.transformer.grid[0, 1, 3, 7, 2]) == 7.0
    assert float(m.flow.weight.abs().max()) < 1e-3 and float(m.flow.bias.abs().max()) == 0.0       # networks.py:214-215
    with pytest.raises(NotImplementedError):
        vxm.networks.VxmDense((32, 32, 32), use_probs=True)
    with pytest.raises(AssertionError):
        vxm.layers.VecInt((8, 8, 8), -1)
    with pytest.raises(ValueError):
        vxm.networks.Unet((32, 32, 32), infeats=2, nb_features=8)          # int features need nb_levels
    with pytest.raises(ValueError):
        vxm.networks.Unet((32, 32, 32), infeats=2, nb_features=[[8], [8]], nb_levels=2)
    m0 = vxm.networks.VxmDense((32, 32, 32), int_steps=0)
    assert m0.integrate is None and m0.resize is None and m0.fullsize is None


def test_planar_model_surface(g_planar):
    """2-D images: same classes, reference state-dict layout ([Cout,Cin,3,3] weights, 2-channel flow conv, 2-D grids),
    bilinear resize mode, and the no-CPU-fallback rule."""
    m = vxm.networks.VxmDense((32, 48), int_steps=5)
    assert [k for k in m.state_dict().keys()] == [str(k) for k in g_planar["state_keys"]]
    shapes = {k: tuple(v.shape) for k, v in vxm.networks.VxmDense((16, 16)).state_dict().items()}
    assert [str(shapes[str(k)]) for k in g_planar["state_keys"]] == [str(s) for s in g_planar["state_shapes"]]
    assert m.resize.mode == "bilinear" and m.flow.weight.shape == (2, 16, 3, 3) and m.transformer.grid.shape == (1, 2, 32, 48)
    assert m.integrate.transformer.grid.shape == (1, 2, 16, 24)
    x = torch.zeros(1, 1, 32, 48)
    with pytest.raises(_lib.VxmHipError):
        m(x, x)
    with pytest.raises(_lib.VxmHipError):
        vxm.losses.NCC().loss(x, x)
    with pytest.raises(_lib.VxmHipError):
        vxm.losses.Grad("l2").loss(None, torch.zeros(1, 2, 8, 8))
    with pytest.raises(NotImplementedError):
        vxm.networks.Unet((32,), infeats=2)


def test_semisupervised_seg_model_surface(tmp_path):
    """Constructor arguments / attributes of the TF reference model (tf/networks.py:293-367) on the torch-side API,
    config capture + checkpoint round trip through LoadableModel, and the no-CPU-fallback rule."""
    m = vxm.networks.VxmDenseSemiSupervisedSeg((16, 16, 16), nb_labels=4, seg_resolution=2, int_steps=3)
    assert isinstance(m.vxm_model, vxm.networks.VxmDense) and m.vxm_model.integrate.nsteps == 3
    assert tuple(m.seg_transformer.grid.shape) == (1, 3, 8, 8, 8) and m.seg_resize.factor == 0.5
    assert m.config["nb_labels"] == 4 and m.config["int_steps"] == 3
    mb = vxm.networks.VxmDenseSemiSupervisedSeg((16, 16, 16), nb_labels=4, bidir_labels=True)
    assert mb.bidir and mb.vxm_model.bidir
    path = str(tmp_path / "semi.pt")
    m.save(path)
    m2 = vxm.networks.VxmDenseSemiSupervisedSeg.load(path, "cpu")
    assert [k for k in m2.state_dict()] == [k for k in m.state_dict()]
    x = torch.zeros(1, 1, 16, 16, 16)
    with pytest.raises(_lib.VxmHipError):
        m(x, x, torch.zeros(1, 4, 8, 8, 8))


def test_data_path_host_side(tmp_path):
    """voxelmorph_amd.data: the npz / npy subset of py/utils.load_volfile and the no-CPU rule of the loader;
    scripts/train.py keeps the reference's flags and defaults (scripts/torch/train.py:52-91)."""
    from voxelmorph_amd import data as vdata
    v = np.random.default_rng(0).random((4, 5, 6))
    np.savez(tmp_path / "a.npz", vol=v, seg=np.zeros((4, 5, 6)))
    np.save(tmp_path / "b.npy", v)
    np.savez(tmp_path / "c.npz", only=v)
    assert np.array_equal(vdata.load_volfile(str(tmp_path / "a.npz")), v)
    assert np.array_equal(vdata.load_volfile(str(tmp_path / "a.npz"), np_var="seg"), np.zeros((4, 5, 6)))
    assert np.array_equal(vdata.load_volfile(str(tmp_path / "b.npy")), v)
    assert np.array_equal(vdata.load_volfile(str(tmp_path / "c.npz")), v)          # single variable: taken whatever its name
    assert vdata.load_volfile(v) is not None
    with pytest.raises(ValueError):
        vdata.load_volfile(str(tmp_path / "missing.npz"))
    with pytest.raises(ValueError):
        vdata.PairLoader([v, v], device="cpu")
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import train as train_cli
    a = train_cli.parse(["--img-list", "x.txt"])
    assert (a.batch_size, a.epochs, a.steps_per_epoch, a.lr, a.int_steps, a.int_downsize, a.image_loss, a.weight, a.bidir) == \
        (1, 1500, 100, 1e-4, 7, 2, "mse", 0.01, False)
    (tmp_path / "list.txt").write_text("a\nb\n\n")
    assert train_cli.read_file_list(str(tmp_path / "list.txt"), prefix="p/", suffix=".npz") == ["p/a.npz", "p/b.npz"]


def test_state_dict_and_checkpoint_format(g_network, tmp_path):
    m = vxm.networks.VxmDense((16, 16, 16))
    keys = list(m.state_dict().keys())
    assert keys == [str(k) for k in g_network["state_keys"]]
    assert [str(tuple(v.shape)) for v in m.state_dict().values()] == [str(s) for s in g_network["state_shapes"]]
    assert sum(p.numel() for p in m.parameters()) == int(g_network["n_params"]) == 327331
    path = os.path.join(tmp_path, "ck.pt")
    m.save(path)
    ck = torch.load(path)
    assert set(ck) == {"config", "model_state"} and not any(k.endswith(".grid") for k in ck["model_state"])
    assert ck["config"] == dict(inshape=(16, 16, 16), nb_unet_features=None, nb_unet_levels=None, unet_feat_mult=1,
                                nb_unet_conv_per_level=1, int_steps=7, int_downsize=2, bidir=False, use_probs=False,
                                src_feats=1, trg_feats=1, unet_half_res=False)
    again = vxm.networks.VxmDense.load(path, "cpu")
    for (k, a), (_, b) in zip(m.state_dict().items(), again.state_dict().items()):
        assert torch.equal(a, b), k


_SWITCH_WORKER = r"""
import os, sys, types, importlib.util
sys.path.insert(0, %(root)r)
from oracle import ref_loader
ref_root = ref_loader.REFERENCE_ROOT
os.environ["VXM_BACKEND"] = "pytorch"
os.environ["VXM_DEVICE"] = "mi355x"
ne = types.ModuleType("neurite"); ne.__version__ = "0.2"; sys.modules["neurite"] = ne
for name in ["pystrum", "pystrum.pynd", "pystrum.pynd.ndutils", "skimage", "skimage.measure"]:
    sys.modules[name] = types.ModuleType(name)
sys.modules["pystrum"].pynd = sys.modules["pystrum.pynd"]; sys.modules["pystrum.pynd"].ndutils = sys.modules["pystrum.pynd.ndutils"]
sys.modules["skimage"].measure = sys.modules["skimage.measure"]
init = os.path.join(ref_root, "voxelmorph", "__init__.py")
text = open(init).read()
# INTEGRATION.md section A, applied to the text of the reference's package init in memory: one branch in front of `backend == 'pytorch'`
old = "if backend == 'pytorch':"
assert text.count(old) == 1
switch = ("if backend == 'pytorch' and os.environ.get('VXM_DEVICE') == 'mi355x':\n"
          "    import voxelmorph_amd\n"
          "    from voxelmorph_amd.torch import layers, networks, losses\n"
          "el" + old)
spec = importlib.util.spec_from_file_location("voxelmorph", init, submodule_search_locations=[os.path.dirname(init)])
mod = importlib.util.module_from_spec(spec)
sys.modules["voxelmorph"] = mod
exec(compile(text.replace(old, switch), init, "exec"), mod.__dict__)
import voxelmorph as vxm
import voxelmorph_amd
from voxelmorph_amd.torch import layers, networks, losses
assert vxm.networks is networks and vxm.layers is layers and vxm.losses is losses
assert vxm.networks.VxmDense is voxelmorph_amd.networks.VxmDense
assert "voxelmorph.torch" not in sys.modules                       # the reference's torch backend was never imported
# what scripts/torch/train.py:140-181 and register.py:78-87 touch, through the switched package
model = vxm.networks.VxmDense(inshape=(32, 32, 32), nb_unet_features=[[16, 32, 32, 32], [32, 32, 32, 32, 32, 16, 16]], bidir=False,
                              int_steps=7, int_downsize=2)
assert type(model).__module__ == "voxelmorph_amd.torch.networks"
for cls in (vxm.losses.NCC, vxm.losses.MSE, vxm.losses.Grad, vxm.losses.Dice, vxm.layers.SpatialTransformer, vxm.layers.VecInt,
            vxm.layers.ResizeTransform):
    assert cls.__module__.startswith("voxelmorph_amd.torch."), cls
assert callable(vxm.losses.Grad('l2', loss_mult=2).loss) and hasattr(model, "save") and hasattr(vxm.networks.VxmDense, "load")
assert vxm.generators.scan_to_scan is not None and vxm.py.utils.default_unet_features() == [[16, 32, 32, 32], [32, 32, 32, 32, 32, 16, 16]]
keys = sorted(k for k in model.state_dict() if not k.endswith(".grid"))
print("SWITCH_OK", len(keys))
"""


@pytest.mark.skipif(not ref_loader.reference_available(), reason="reference tree not present")
def test_integration_switch_binds_this_package_inside_the_reference(tmp_path):
    """INTEGRATION.md section A exercised: the three-line branch a maintainer would add to voxelmorph/__init__.py:32-45, applied in memory to the
    reference's own package init (read from /root/reference at run time, nothing copied), makes `vxm.networks / layers / losses` THIS
    package's modules while `vxm.generators` and `vxm.py` stay the reference's; the reference's torch backend is never imported."""
    script = os.path.join(tmp_path, "switch_worker.py")
    with open(script, "w") as f:
        f.write(_SWITCH_WORKER % dict(root=ROOT))
    out = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "SWITCH_OK 24" in out.stdout, out.stdout[-500:]
    # and the state_dict keys behind that count are the reference's (test_state_dict_and_checkpoint_format pins them to the golden list)


@pytest.mark.skipif(not ref_loader.reference_available(), reason="reference tree not present")
def test_checkpoint_interop_with_live_reference(tmp_path):
    ref = ref_loader.load_reference()
    mine = vxm.networks.VxmDense((16, 16, 16), int_steps=5)
    p1 = os.path.join(tmp_path, "mine.pt")
    mine.save(p1)
    theirs = ref.networks.VxmDense.load(p1, "cpu")                  # our file -> reference loader
    for k, v in mine.state_dict().items():
        assert torch.equal(v, theirs.state_dict()[k]), k
    p2 = os.path.join(tmp_path, "theirs.pt")
    theirs.save(p2)
    back = vxm.networks.VxmDense.load(p2, "cpu")                    # reference file -> our loader
    assert back.config == theirs.config
    for k, v in theirs.state_dict().items():
        assert torch.equal(v, back.state_dict()[k]), k


@pytest.mark.parametrize("kw", [dict(), dict(nb_features=[[4, 8], [8, 8, 4]]), dict(nb_features=8, nb_levels=3, nb_conv_per_level=2),
                                dict(half_res=True), dict(nb_features=[[8, 8], [8, 8]])])
def test_unet_plan_matches_reference_bookkeeping(kw):
    net = vxm.networks.Unet((32, 32, 32), infeats=2, **kw)
    plan = net.plan([2])
    shapes = orc.state_dict_shapes((32, 32, 32), **kw)[:-2]          # drop the flow conv
    want = [(s[1], s[0]) for n, s in shapes if n.endswith("weight")]
    assert [(c[0], c[1]) for c in plan.convs] == want
    assert plan.ch[plan.out] == net.final_nf
    names = ["unet_model." + n for n, _ in net.named_parameters()]
    assert names == [n for n, _ in shapes]


def test_shard_range_and_flat_bucket():
    from voxelmorph_amd import dist as vdist
    from voxelmorph_amd.optim import FlatAdam
    assert vdist.shard_range(32, 3, 8) == (12, 16)
    with pytest.raises(AssertionError, match="multiple of the nr of gpus"):
        vdist.shard_range(6, 0, 4)
    m = vxm.networks.VxmDense((16, 16, 16))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    opt = FlatAdam(m, lr=1e-4)
    assert opt.n == 327331 and opt.flat_param.is_contiguous()
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    ptrs = [p.data_ptr() for p in m.parameters()]
    assert ptrs[0] == opt.flat_param.data_ptr() and sorted(ptrs) == ptrs            # views into one bucket, in order
    with pytest.raises(_lib.VxmHipError):
        opt.step()                                                                 # Adam is a HIP kernel


_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import voxelmorph_amd as vxm
from voxelmorph_amd import dist as vdist
from voxelmorph_amd.optim import FlatAdam
from oracle import vxm_oracle as orc
rank, local, world = vdist.init_from_env(backend="gloo")
inshape = (16, 16, 16)
rng = np.random.default_rng(3)
src = torch.from_numpy(rng.random((2, 1) + inshape).astype(np.float32))
trg = torch.from_numpy(rng.random((2, 1) + inshape).astype(np.float32))
lo, hi = vdist.shard_range(2, rank, world)
model = vxm.networks.VxmDense(inshape, int_steps=2)
model.load_state_dict(orc.seeded_state_dict(inshape, seed=rank, flow_std=0.1), strict=False)   # ranks start different
opt = FlatAdam(model, lr=1e-4, direct_grads=False)
opt.broadcast_params(0)                                                                        # ... and are made equal
sd = {k: v for k, v in model.named_parameters()}
loss, _ = orc.train_step_loss(src[lo:hi], trg[lo:hi], sd, "mse", 0.01, int_steps=2)         # oracle = compute on CPU
loss.backward()
opt.load_grads_from_params()
opt.reduce_grads()
mean_grad = opt.flat_grad / world
# single-process reference on the global batch
ref = vxm.networks.VxmDense(inshape, int_steps=2)
ref.load_state_dict(orc.seeded_state_dict(inshape, seed=0, flow_std=0.1), strict=False)
rsd = {k: v for k, v in ref.named_parameters()}
rl, _ = orc.train_step_loss(src, trg, rsd, "mse", 0.01, int_steps=2)
rl.backward()
rg = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
rel = float((mean_grad - rg).norm() / rg.norm())
lt = torch.tensor([float(loss)], dtype=torch.float64)
dist.all_reduce(lt)
assert rel < 1e-5, rel
assert abs(float(lt) / world - float(rl)) < 1e-6
assert vdist.max_over_ranks(float(rank), "cpu") == world - 1
vdist.barrier()
print("rank", rank, "ok", rel)
"""


def test_data_parallel_equivalence_gloo_world2(tmp_path):
    script = os.path.join(tmp_path, "worker.py")
    with open(script, "w") as f:
        f.write(_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", script],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


_COMM_WORKER = """
import os, sys, json, types
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
import bench
from voxelmorph_amd import comm as vcomm, dist as vdist
# (1) the evidence a multi-rank bench line carries about its exchange, over a stand-in optimiser on the gloo group
opt = types.SimpleNamespace(comm=None, world=dist.get_world_size(), n=327331, group=None, flat_grad=torch.zeros(327331))
ev = bench.comm_evidence(opt, torch.device("cpu"))
assert ev["backend"].startswith("torch.distributed") and ev["ranks_seen"] == 2 and ev["bucket_bytes"] == 4 * 327331 and ev["allreduce_us"] > 0
assert "transport" in ev
# (2) a rank that cannot LOAD libvxm_comm.so: every rank must come back with None (no rank left waiting in the unique-id broadcast)
if dist.get_rank() == 1:
    vcomm.LIB_PATH = "/nonexistent/libvxm_comm.so"
    vcomm._lib = None
got = vcomm.NativeComm.try_from_torch_dist()
assert got is None, got
# (3) required=True turns that into an error on EVERY rank (bench.py: never a silently different exchange) -- native_comm() itself only
# engages on the nccl backend, so the rule is checked on its decision function
os.environ["VXM_COMM"] = ""
assert vdist.native_comm(required=True) is None          # gloo: not a HIP job, nothing required
real_backend = dist.get_backend
dist.get_backend = lambda *a, **k: "nccl"                # a HIP job whose communicator cannot be built (the library is gone on rank 1, see (2))
try:
    assert vdist.native_comm(required=False) is None
    try:
        vdist.native_comm(required=True)
        raise SystemExit("native_comm(required=True) returned without its communicator")
    except RuntimeError as exc:
        assert "VXM_COMM=torch" in str(exc)
finally:
    dist.get_backend = real_backend
dist.barrier()
print("ok")
"""


def test_comm_evidence_and_library_agreement_gloo_world2(tmp_path):
    """bench.py's `comm` object over a 2-rank gloo group (fields and types the scaling run will leave behind), and the three-round
    agreement of `NativeComm.try_from_torch_dist`: one rank without the library -> both ranks fall back together, nobody hangs."""
    script = os.path.join(tmp_path, "comm_worker.py")
    with open(script, "w") as f:
        f.write(_COMM_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29543", script],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stdout.count("ok") == 2
    assert "cannot be loaded on 1 of 2 ranks" in out.stderr


def test_bench_gpus_n_self_launches_one_rank_per_gpu():
    """`python bench.py --gpus 2` with no torchrun environment (how the scaling driver invokes it) re-launches itself under
    torch.distributed.run with two ranks on a 127.0.0.1 rendezvous; on this GPU-less host both ranks get past the process-group
    initialisation (gloo) and the world-size check, then stop at the explicit no-fallback message."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert out.stderr.count("bench.py needs an MI355X") == 2, out.stdout[-2000:] + out.stderr[-3000:]
    assert "WORLD_SIZE" not in out.stderr.split("bench.py needs an MI355X")[0][-300:]


def test_bench_reports_the_roofline_that_binds_the_kernel():
    """bench.py prices a kernel against the roofline with the larger minimum time: the fp32 conv (119 GFLOP, 0.88 GB per launch) is an
    MFMA kernel; a 16-channel bf16 layer (0.095 TFLOP, 0.44 GB) is an HBM kernel whose MFMA fraction is kept beside; a 32-channel bf16
    layer (0.38 TFLOP, 0.88 GB) is an MFMA kernel again; byte-only kernels (the warps) stay HBM."""
    sys.path.insert(0, ROOT)
    import bench
    t8 = bench.binding_roofline("k_conv3d_k3_t8<1>", dict(launches=10, ms=9.7, flops=10 * 118.9e9, nominal=10 * 118.9e9, bytes=0.0))
    assert t8["bound"] == "mfma" and t8["unit"] == "TFLOP/s" and abs(t8["frac"] - 122.6 / 157.3) < 1e-2 and t8["peak"] == bench.FP32_MFMA_PEAK_TFLOPS
    a = bench.binding_roofline("k_bf16_conv<1,8,0>", dict(launches=2, ms=0.36, flops=2 * 0.12e12, nominal=2 * 0.095e12, bytes=2 * 0.44e9))
    assert a["bound"] == "hbm" and a["peak"] == bench.HBM_PEAK_GBS and abs(a["achieved"] - 2444.4) < 1.0
    assert abs(a["mfma_frac"] - (0.19e12 / 0.36e-3 / 1e12) / bench.BF16_MFMA_PEAK_TFLOPS) < 1e-9 and a["algorithmic_per_launch"] == 0.44e9
    b = bench.binding_roofline("k_bf16_conv<2,6,0>", dict(launches=1, ms=0.44, flops=0.40e12, nominal=0.38e12, bytes=0.88e9))
    assert b["bound"] == "mfma" and abs(b["achieved"] - 0.38e12 / 0.44e-3 / 1e12) < 1e-6 and "mfma_frac" not in b
    w = bench.binding_roofline("warp3d_fwd", dict(launches=4, ms=0.2, flops=0.0, nominal=0.0, bytes=4 * 0.14e9))
    assert w["bound"] == "hbm" and abs(w["frac"] - 2800.0 / 8000.0) < 1e-9
    for r in (t8, a, b, w):
        assert r["traffic"] is None and r["avg_launch_ms"] > 0
    # split-fp32 kernels run on the 16-bit matrix pipe with six (three bf16 pieces) or three (two fp16 pieces) MFMAs per fp32 MAC block: priced
    # against 2500 / 6 resp. 2500 / 3, not against the fp32-MFMA peak; the piece scheme is the last template argument of the region label
    sp = bench.binding_roofline("k_s3_conv<1,4,1,3>", dict(launches=5, ms=2.94, flops=594.5e9, nominal=594.5e9, bytes=0.0))
    assert sp["bound"] == "mfma" and abs(sp["peak"] - 2500.0 / 6.0) < 1e-9 and abs(sp["achieved"] - 202.2) < 0.1 and abs(sp["frac"] - 202.2 / 416.67) < 1e-3
    assert abs(sp["matrix_pipe_tflops"] - 6 * sp["achieved"]) < 1e-9 and "1.29 x the fp32-MFMA peak" in sp["peak_note"]
    s2 = bench.binding_roofline("k_s3_conv<2,4,1,2>", dict(launches=5, ms=2.94, flops=594.5e9, nominal=594.5e9, bytes=0.0))
    assert abs(s2["peak"] - 2500.0 / 3.0) < 1e-9 and abs(s2["matrix_pipe_tflops"] - 3 * s2["achieved"]) < 1e-9 and "fp16 x 2" in s2["peak_note"]
    for label, peak in (("k_s3_bwd_weight<2>", 2500.0 / 3.0), ("k_s3_bwd_weight<3>", 2500.0 / 6.0)):
        assert abs(bench.binding_roofline(label, dict(launches=5, ms=3.9, flops=547e9, nominal=547e9, bytes=0.0))["peak"] - peak) < 1e-9


def test_split_engine_enumerates_its_operators(monkeypatch):
    """`functional._s3_jobs`: the operators of the default VxmDense plan that one batched launch packs for the split kernels when the
    two finest levels qualify (stand-in for the C-side shape test: levels 0 and 1, channel counts in multiples of 8): forward operators of
    the plain layers, and for a training step the adjoints `conv_bwd_data` will ask for -- per segment, 48-channel results as 32 + 16."""
    from voxelmorph_amd.torch import functional as VF
    m = vxm.networks.VxmDense((160, 192, 224), int_steps=0)
    plan = m.unet_model.plan(m._feats, extra=((m.flow.out_channels, 1.0),))
    params = list(m.unet_model.conv_params()) + [m.flow.weight, m.flow.bias]
    monkeypatch.setattr(VF, "FP32_ENGINE", "split")
    monkeypatch.setattr(VF, "S3_UP", False)
    monkeypatch.setattr(VF, "s3_route", lambda c0, up0, c1, cout, B, D, H, W: (not up0) and c0 % 8 == 0 and c1 % 8 == 0 and cout >= 8 and D >= 80)
    fwd = [(tuple(w.shape[:2]), lo, hi, flip, seg0) for w, lo, hi, flip, seg0 in VF._s3_jobs(plan, params, 1, (160, 192, 224), False, False)]
    assert fwd == [((32, 16), 0, 16, False, 16), ((16, 32), 0, 32, False, 32), ((16, 16), 0, 16, False, 16)]          # enc1, rem1, rem2
    train = [(tuple(w.shape[:2]), lo, hi, flip, seg0) for w, lo, hi, flip, seg0 in VF._s3_jobs(plan, params, 1, (160, 192, 224), True, False)]
    adj = [j for j in train if j[3]]
    assert ((32, 16), 0, 16, True, 32) in adj and ((16, 32), 0, 32, True, 16) in adj and ((16, 16), 0, 16, True, 16) in adj
    assert ((32, 48), 32, 48, True, 32) in adj and ((32, 64), 32, 64, True, 32) in adj              # skip segments of rem0 and dec3
    assert ((3, 16), 0, 16, True, 3) not in adj                                                      # 3 channels: not a split operand
    assert [j for j in train if not j[3]] == fwd


def test_channel_blocked_tensor_rule(monkeypatch):
    """`functional._blocked_tensors`: which activations of the fused U-Net are kept channel-blocked between the split kernels.  With every
    shape test of the C side answering yes for multiples-of-8 channels at the two finest levels (stand-ins below), the default VxmDense plan
    yields the outputs of remaining[0], remaining[1] and -- round 6, late: the three few-channel kernels around the flow conv take the layout
    flags -- remaining[2], whose consumer is the 3-channel flow conv (VXM_BLOCKED_LAST=0, or a predicate of those kernels saying no, leaves it
    planar); the first layer's output and the encoder / decoder outputs are read by MaxPool / as skip or upsampled segments.  A U-Net with two
    convolutions per level also keeps the first output of every encoder / decoder pair -- except the first layer's, whose producer is the
    few-input-channel kernel -- and VXM_BLOCKED=0 / the other engines keep everything planar."""
    import types
    from voxelmorph_amd.torch import functional as VF
    stub = types.SimpleNamespace(
        vxm_conv3d_k3_s3_layout_ok=lambda c0, c1, up, cout, H, pieces: int(pieces == 2 and c1 == 0 and not up and c0 % 8 == 0 and cout % 8 == 0 and H >= 8),
        vxm_conv3d_k3_s3_bwd_weight_ok=lambda c, cout, B, D, H, W: int(c % 16 == 0 and cout % 16 == 0 and D >= 80),
        vxm_conv3d_k3_fewout_ok=lambda x, xbs, y, ybs, cin, cout, W: int(cout <= 4 and W % 4 == 0),
        vxm_conv3d_k3_fewch_bwd_weight_ok=lambda x0, c0, bs0, x1, c1, bs1, dz, dzbs, cout, pieces, W: int(pieces == 2 and c0 == 16 and c1 == 0 and cout <= 3 and few["bw"]),
        vxm_conv3d_k3_fwd_layout_ok=lambda x0, c0, bs0, x1, c1, bs1, wp, cout, B, D, H, W: int(c0 + c1 <= 4 and cout % 8 == 0))
    few = {"bw": True}
    monkeypatch.setattr(VF, "_lib", types.SimpleNamespace(lib=lambda: stub))
    monkeypatch.setattr(VF, "FP32_ENGINE", "f16x2")
    monkeypatch.setattr(VF, "BLOCKED", True)
    fine = lambda *a: a[-3] >= 80                      # D of the launch: levels 0 and 1 of a 160x192x224 volume
    monkeypatch.setattr(VF, "s3_route", lambda c0, up0, c1, cout, B, D, H, W: (not up0) and c0 % 8 == 0 and c1 % 8 == 0 and cout >= 8 and D >= 80)
    for name in ("s3u_route", "s3u_bwd_low_route", "s3u_bwd_weight_route"):
        monkeypatch.setattr(VF, name, lambda *a: fine(*a))
    shape = (160, 192, 224)
    m = vxm.networks.VxmDense(shape, int_steps=0)
    plan = m.unet_model.plan(m._feats, extra=((m.flow.out_channels, 1.0),))
    convs = [op for op in plan.ops if op["kind"] == "conv"]
    rem0, rem1, rem2 = convs[8]["dst"], convs[9]["dst"], convs[10]["dst"]        # execution order: 4 encoder, 4 decoder, 3 remaining, flow
    assert (plan.ch[rem0], plan.ch[rem1], plan.ch[rem2], plan.lvl[rem0], plan.lvl[rem1], plan.lvl[rem2]) == (32, 16, 16, 0, 0, 0)
    assert plan.consumers[rem2] == [plan.producer[plan.out]] and plan.ch[plan.out] == 3
    assert VF._blocked_tensors(plan, 1, shape) == frozenset({rem0, rem1, rem2})
    assert VF._blocked_tensors(plan, 4, shape) == frozenset({rem0, rem1, rem2})
    monkeypatch.setattr(VF, "BLOCKED_LAST", False)
    assert VF._blocked_tensors(plan, 1, shape) == frozenset({rem0, rem1})
    monkeypatch.setattr(VF, "BLOCKED_LAST", True)
    few["bw"] = False                                    # the flow conv's weight gradient would not take a blocked x: the tensor stays planar
    plan.__dict__.pop("_blocked_cache", None)
    assert VF._blocked_tensors(plan, 1, shape) == frozenset({rem0, rem1})
    few["bw"] = True
    plan.__dict__.pop("_blocked_cache", None)
    monkeypatch.setattr(VF, "BLOCKED", False)
    assert VF._blocked_tensors(plan, 1, shape) == frozenset()
    monkeypatch.setattr(VF, "BLOCKED", True)
    monkeypatch.setattr(VF, "FP32_ENGINE", "split")
    assert VF._blocked_tensors(plan, 1, shape) == frozenset()
    monkeypatch.setattr(VF, "FP32_ENGINE", "f16x2")
    # two ConvBlocks per level, 16 features everywhere: (first, second) pairs at the two finest levels
    u = vxm.networks.Unet(shape, infeats=2, nb_features=16, nb_levels=3, nb_conv_per_level=2)
    plan2 = u.plan([2])
    got = VF._blocked_tensors(plan2, 1, shape)
    convs2 = [op for op in plan2.ops if op["kind"] == "conv"]
    firsts = {convs2[2]["dst"]}                            # encoder level 1, first conv (level 0's first conv is the first layer: excluded)
    assert firsts <= got and convs2[0]["dst"] not in got
    for t in got:                                          # every member: one plain conv consumer, multiple-of-16 channels, not the network output
        assert len(plan2.consumers[t]) == 1 and plan2.ch[t] % 16 == 0 and t != plan2.out
        assert tuple(plan2.ops[plan2.consumers[t][0]]["src"]) == (t, False, None)


def test_bf16_engine_enumerates_the_weight_operators_of_a_step():
    """`functional_bf16._pack_jobs`: what one pass over the default VxmDense plan packs in its single launch — the forward operator of
    each of the 12 convolutions; for a training step also the adjoint per input segment (two for the decoder layers that read
    cat([upsample(x), skip]), none for the first layer unless the inputs want gradients)."""
    from voxelmorph_amd.torch import functional_bf16 as VB
    m = vxm.networks.VxmDense((32, 32, 32), int_steps=0)
    plan = m.unet_model.plan(m._feats, extra=((m.flow.out_channels, 1.0),))
    params = list(m.unet_model.conv_params()) + [m.flow.weight, m.flow.bias]
    fwd = VB._pack_jobs(plan, params, 2, False, False)
    assert [(tuple(w.shape[:2]), lo, n, flip) for w, lo, n, flip in fwd] == [
        ((16, 2), 0, 2, False), ((32, 16), 0, 16, False), ((32, 32), 0, 32, False), ((32, 32), 0, 32, False), ((32, 32), 0, 32, False),
        ((32, 64), 0, 64, False), ((32, 64), 0, 64, False), ((32, 64), 0, 64, False), ((32, 48), 0, 48, False), ((16, 32), 0, 32, False),
        ((16, 16), 0, 16, False), ((3, 16), 0, 16, False)]
    train = VB._pack_jobs(plan, params, 2, True, False)
    assert len(train) == 27 and [j for j in train if not j[3]] == fwd
    adj = [(tuple(w.shape[:2]), lo, n) for w, lo, n, flip in train if flip]
    assert ((32, 48), 0, 32) in adj and ((32, 48), 32, 16) in adj and adj.count(((32, 64), 0, 32)) == 3 and adj.count(((32, 64), 32, 32)) == 3
    assert not any(sh == (16, 2) for sh, _, _ in adj)
    with_inputs = VB._pack_jobs(plan, params, 2, True, True)
    assert len(with_inputs) == 28 and ((16, 2), 0, 2) in [(tuple(w.shape[:2]), lo, n) for w, lo, n, flip in with_inputs if flip]


def test_pack_cache_key_sees_reseated_storage_and_invalidate_accepts_modules():
    """the packed-operator cache key is (version counter, storage address, generation) -- `p.data = other` moves the
    address without touching the counter; a write INTO the same storage needs `voxelmorph_amd.invalidate_packs` (exported, takes a module
    or an iterable of parameters); VXM_FP32_ENGINE is validated."""
    import subprocess
    import sys
    import voxelmorph_amd as vxm
    from voxelmorph_amd.torch import functional_bf16 as VB
    lin = torch.nn.Conv3d(2, 4, 3)
    k0 = VB._ver(lin.weight)
    lin.weight.data = lin.weight.data.clone()
    k1 = VB._ver(lin.weight)
    assert k0 != k1 and k0[0] == k1[0]                       # same version counter, another address
    lin.weight.data.mul_(2.0)                                # invisible to the counter and the address ...
    assert VB._ver(lin.weight) == k1
    vxm.invalidate_packs(lin)                                # ... hence the exported call
    assert VB._ver(lin.weight) != k1 and VB._ver(lin.bias)[2] == 1
    with torch.no_grad():
        lin.weight.add_(1.0)                                 # what every optimiser does: bumps the counter
    assert VB._ver(lin.weight)[0] == k1[0] + 1
    from voxelmorph_amd.torch import functional as VF
    assert VF.FP32_ENGINE in ("f16x2", "split", "native") and VF.s3_pieces() == (2 if VF.FP32_ENGINE == "f16x2" else 3)
    r = subprocess.run([sys.executable, "-c", "import voxelmorph_amd"], env=dict(os.environ, VXM_FP32_ENGINE="splitt"),
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode != 0 and "VXM_FP32_ENGINE" in r.stderr
    r = subprocess.run([sys.executable, "-c", "import torch, voxelmorph_amd\nfrom voxelmorph_amd.torch import functional_bf16 as VB\n"
                        "w = torch.nn.Parameter(torch.zeros(3))\nassert VB._ver(w) != VB._ver(w)\nassert VB._inplace_ver(w) == VB._inplace_ver(w)"],
                       env=dict(os.environ, VXM_PACK_CACHE="0"), capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr


def test_nifti_and_mgz_io_without_nibabel(tmp_path):
    """voxelmorph_amd.nifti + data.load_volfile / save_volfile (py/utils.py:69-158): NIfTI-1 round trips (dtype, Fortran data order,
    sform affine, gzip), a hand-built big-endian header with qform only and scl_slope scaling, the reference's default LIA affine,
    mgz, and the keyword arguments of load_volfile."""
    import gzip
    import struct
    from voxelmorph_amd import data as vdata
    from voxelmorph_amd import nifti
    rng = np.random.default_rng(0)
    aff = np.array([[-1.5, 0.1, 0.0, 90.0], [0.2, 0.0, 2.0, -126.0], [0.0, -1.0, 0.1, 72.0], [0, 0, 0, 1.0]])
    for dt in (np.float32, np.int16, np.uint8, np.float64):
        for name in ("v.nii", "v.nii.gz"):
            v = (rng.random((5, 6, 7)) * 100).astype(dt)
            vdata.save_volfile(v, str(tmp_path / name), aff)
            got, a2 = vdata.load_volfile(str(tmp_path / name), ret_affine=True)
            assert got.dtype == dt and np.array_equal(got, v)
            np.testing.assert_allclose(a2, aff, rtol=0, atol=1e-5)          # (the header stores float32)
    raw = gzip.open(tmp_path / "v.nii.gz").read()
    assert struct.unpack("<i", raw[:4])[0] == 348 and raw[344:348] == b"n+1\0" and struct.unpack("<f", raw[108:112])[0] == 352.0
    assert struct.unpack("<8h", raw[40:56])[:4] == (3, 5, 6, 7)
    assert np.frombuffer(raw, "<f8", 3, 352).tolist() == v.reshape(-1, order="F")[:3].tolist()      # x fastest on disk
    # default affine of the reference (LIA, volume centre at the origin) when none is given
    vdata.save_volfile(v, str(tmp_path / "d.nii"))
    _, ad = vdata.load_volfile(str(tmp_path / "d.nii"), ret_affine=True)
    np.testing.assert_allclose(ad, [[-1, 0, 0, 2.5], [0, 0, 1, -3.5], [0, -1, 0, 3.0], [0, 0, 0, 1]], atol=1e-6)
    # 4-D flow fields round-trip too ([3, D, H, W] as register.py --warp writes them)
    w4 = rng.standard_normal((3, 4, 5, 6)).astype(np.float32)
    vdata.save_volfile(w4, str(tmp_path / "w.nii.gz"), aff)
    assert np.array_equal(vdata.load_volfile(str(tmp_path / "w.nii.gz")), w4)
    # a big-endian file with a qform only (90 degree rotation about z, zooms 2 / 3 / 4) and scaled int16 data
    hdr = bytearray(348)
    struct.pack_into(">i", hdr, 0, 348)
    struct.pack_into(">8h", hdr, 40, 3, 2, 3, 4, 1, 1, 1, 1)
    struct.pack_into(">2h", hdr, 70, 4, 16)
    struct.pack_into(">8f", hdr, 76, 1.0, 2.0, 3.0, 4.0, 1, 1, 1, 1)
    struct.pack_into(">3f", hdr, 108, 352.0, 0.5, 10.0)
    struct.pack_into(">2h", hdr, 252, 1, 0)
    s2 = float(np.sqrt(0.5))
    struct.pack_into(">6f", hdr, 256, 0.0, 0.0, s2, 7.0, 8.0, 9.0)
    hdr[344:348] = b"n+1\0"
    data = np.arange(24, dtype=">i2")
    (tmp_path / "be.nii").write_bytes(bytes(hdr) + b"\0\0\0\0" + data.tobytes())
    vb, ab = nifti.read_nifti(str(tmp_path / "be.nii"))
    assert vb.dtype == np.float64 and vb.shape == (2, 3, 4) and vb[1, 0, 0] == 0.5 * 1 + 10 and vb[0, 1, 0] == 0.5 * 2 + 10
    np.testing.assert_allclose(ab, [[0, -3, 0, 7], [2, 0, 0, 8], [0, 0, 4, 9], [0, 0, 0, 1]], atol=1e-6)
    # mgz: big-endian MGH header, float data, direction cosines + centre
    mh = bytearray(284)
    struct.pack_into(">7ih", mh, 0, 1, 2, 3, 4, 1, 3, 0, 1)
    struct.pack_into(">3f", mh, 30, 1.0, 1.0, 1.0)
    struct.pack_into(">9f", mh, 42, -1, 0, 0, 0, 0, -1, 0, 1, 0)
    struct.pack_into(">3f", mh, 78, 1.0, 2.0, 3.0)
    md = np.arange(24, dtype=">f4")
    with gzip.open(tmp_path / "m.mgz", "wb") as f:
        f.write(bytes(mh) + md.tobytes())
    vm, am = vdata.load_volfile(str(tmp_path / "m.mgz"), ret_affine=True)
    assert vm.shape == (2, 3, 4) and vm[1, 0, 0] == 1.0 and vm[0, 0, 1] == 6.0
    np.testing.assert_allclose(am, [[-1, 0, 0, 2.0], [0, 0, 1, 0.0], [0, -1, 0, 4.5], [0, 0, 0, 1]], atol=1e-6)
    # keyword arguments of the reference's load_volfile
    v0 = vdata.load_volfile(str(tmp_path / "v.nii"))
    p = vdata.load_volfile(str(tmp_path / "v.nii"), add_batch_axis=True, add_feat_axis=True, pad_shape=(8, 8, 8))
    assert p.shape == (1, 8, 8, 8, 1) and np.array_equal(p[0, 1:6, 1:7, 0:7, 0], v0) and float(np.abs(p).sum()) == float(np.abs(v0).sum())
    z = vdata.load_volfile(str(tmp_path / "v.nii"), add_feat_axis=True, resize_factor=0.5)
    assert z.shape[-1] == 1 and z.shape[:3] == (2, 3, 4)                    # scipy's nearest-neighbour zoom rounds 2.5 -> 2, 3.5 -> 4
    with pytest.raises(ValueError):
        vdata.save_volfile(v, str(tmp_path / "x.mgz"))


def test_pack_job_list_covers_the_whole_range_adjoint_of_unfused_upsampled_layers():
    """Host logic, no device: the operators the fused U-Net packs at the start of a step (functional._s3_jobs) must contain every adjoint its
    backward pass asks for.  Round 5 found one missing with a rocprofv3 trace -- the default VxmDense's decoder conv at 40x48x56 has no fused
    low-resolution kernel, so its backward-data runs the adjoint of the WHOLE virtual concat (64 -> 32 outputs of w[32][64]) -- packed on demand
    in the middle of the backward chain, 0.23 ms per step behind the weight gradients' persistent blocks."""
    import voxelmorph_amd as vxm
    from voxelmorph_amd.torch import functional as VF
    shape = (160, 192, 224)
    m = vxm.networks.VxmDense(shape, int_steps=7, int_downsize=2)
    plan = m.unet_model.plan(m._feats, extra=((m.flow.out_channels, 1.0),))
    params = m.unet_model.conv_params() + [m.flow.weight, m.flow.bias]
    jobs = [(tuple(w.shape[:2]), lo, hi, flip, seg0) for w, lo, hi, flip, seg0 in VF._s3_jobs(plan, params, 1, shape, True, False)]
    assert ((32, 64), 0, 64, True, 32) in jobs                 # decoder level 2: the adjoint of the whole 64-channel concat
    assert ((32, 64), 32, 64, True, 32) in jobs                # decoder level 3: only its skip segment (the upsampled one has k_s3u_dlow)
    assert ((32, 48), 32, 48, True, 32) in jobs                # remaining[0]: likewise
    assert not any(j[0] == (32, 48) and j[1] == 0 and j[3] for j in jobs)
    assert len(jobs) == len(set(jobs))
    # inference asks for no adjoint at all, and the first launch that reads a packed operator is the second conv (op 2: conv, pool, conv)
    assert not any(flip for _, _, _, flip, _ in VF._s3_jobs(plan, params, 1, shape, False, False))
    # (round 6: three groups with their own events -- the batched split operators are first read by op 2, the fp32-MFMA operators of the coarse
    # levels by op 6 (conv, pool x 3), the collapsed upsample operators by the first cat([upsample, skip]) layer on the split engine)
    first = VF._prepack_plan(plan, params, 1, shape, True, False, dry=True)
    assert first[:2] == (2, 6) and first[2] > 6 and plan.ops[first[2]]["src"][1]
