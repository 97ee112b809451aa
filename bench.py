#!/usr/bin/env python
"""Headline benchmark: volume-pairs/s of the VxmDense training step on MI355X.

Workload (BASELINE.json `metric`, configs[2]): VxmDense 3-D 160x192x224, int_steps=7, int_downsize=2,
NCC(9^3) + Grad('l2', loss_mult=2) (lambda=1), fp32, Adam lr 1e-4, synthetic U[0,1) volume pairs resident
in HBM.  One step = forward + loss + backward + gradient all-reduce + Adam for `batch_per_gpu` pairs
per rank (weak scaling: per-GPU work fixed as N grows).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: bf16 MFMA dense peak (never the 2:1-sparsity headline)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--shape", type=str, default="160,192,224")
    ap.add_argument("--int-steps", type=int, default=None)
    ap.add_argument("--config", choices=["diffeo_fp32", "dense_bf16", "diffeo_bf16", "semisup_fp32"], default="diffeo_fp32",
                    help="diffeo_fp32 = BASELINE.json configs[2], the headline metric (default); dense_bf16 = configs[1]: int_steps=0, "
                         "MSE + 0.01 Grad, bf16 activations / fp32 accumulate under torch.autocast; diffeo_bf16 = the headline network "
                         "and losses with bf16 activations (an extra, labelled line: NOT the headline metric, which is fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true")                 # only the headline (profiling runs)
    ap.add_argument("--cpu-baseline-steps", type=int, default=2)               # timed CPU steps after one warm-up, at the FULL shape
    ap.add_argument("--cpu-threads", type=int, default=0)                      # 0: min(32, cores), see cpu_baseline()
    ap.add_argument("--gpu-baseline", action="store_true",
                    help="time the reference's ATen call sequence on THIS GPU through stock torch-ROCm now (minutes: this image has no MIOpen "
                         "kernel database for gfx950); without the flag the line carries the measurement committed under profiles/")
    ap.add_argument("--no-gpu-baseline", action="store_true")                  # neither run nor quote the stock torch-ROCm leg
    ap.add_argument("--torch-rocm-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def cpu_baseline(shape, int_steps, timed_steps, threads, image_loss="ncc", lam=1.0):
    """The reference's torch path timed on this host's cores at the benchmark shape itself (SURVEY.md §8d / BASELINE.md §4):
    B = 1, fp32, NCC(9^3) + Grad('l2', x2), Adam lr 1e-4, one warm-up + `timed_steps` timed training steps.

    kind "reference": the UNMODIFIED upstream modules imported from /root/reference (oracle/ref_loader.py; NCC through the
    `.to("cuda")` no-op shim because losses.py:29 hard-codes the device) — only where that tree exists (the build
    container).  kind "port": the oracle's restatement of the same ATen call sequence (oracle/vxm_oracle.py) — the GPU box
    has no /root/reference.  Reported baseline, not the optimisation target."""
    import numpy as np
    from oracle import ref_loader
    from oracle import vxm_oracle as orc
    # torch's intra-op pool degrades beyond one socket's worth of threads (256 threads on the 2x64-core bench host ran
    # 10x slower than 8 threads elsewhere): at most 32 unless told otherwise; the number used is reported as `cores`
    cores = threads if threads > 0 else min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    rng = np.random.default_rng(1234)
    src = torch.from_numpy(rng.random((1, 1) + shape).astype(np.float32))
    trg = torch.from_numpy(rng.random((1, 1) + shape).astype(np.float32))
    if ref_loader.reference_available():
        kind = "reference"
        ref = ref_loader.load_reference()
        torch.manual_seed(0)
        model = ref.networks.VxmDense(shape, int_steps=int_steps, int_downsize=2)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)
        ncc = ref.losses.NCC().loss if image_loss == "ncc" else ref.losses.MSE().loss
        reg = ref.losses.Grad("l2", loss_mult=2).loss

        def step():
            with ref_loader.cuda_alias_to_cpu():
                y, pre = model(src, trg)
                loss = ncc(trg, y) + lam * reg(None, pre)
            opt.zero_grad()
            loss.backward()
            opt.step()
    else:
        kind = "port"
        sd = orc.seeded_state_dict(shape, seed=0, flow_std=1e-5)
        params = [v.requires_grad_() for v in sd.values()]
        opt = torch.optim.Adam(params, lr=1e-4)

        def step():
            opt.zero_grad()
            loss, _ = orc.train_step_loss(src, trg, sd, image_loss, lam, int_steps=int_steps, int_downsize=2)
            loss.backward()
            opt.step()

    step()                                  # warm-up
    times = []
    for _ in range(max(1, timed_steps)):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    return {
        "value": 1.0 / dt, "unit": "volume-pairs/s", "cores": cores, "kind": kind, "s_per_step": dt,
        "sample": "1 warm-up + %d timed training step(s) (fwd+%s+Grad+bwd+Adam, int_steps=%d) at the full %s shape, B=1, fp32, %d torch "
                  "threads of %d host cores: %s s/step" % (len(times), image_loss.upper(), int_steps, "x".join(map(str, shape)), cores,
                                                           os.cpu_count() or 0, ", ".join("%.2f" % t for t in times)),
        "cpu_model": _cpu_model(),
    }


def _counter_file(pattern):
    """The committed counter summary to use: among profiles/<pattern> the one collected on THIS tree's kernel sources (its `_meta.csrc_sha`),
    newest name first; when none matches, the newest, which the callers then refuse as stale with the reason in the line."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=lambda f: (len(os.path.basename(f)), f))
    if not files:
        return None
    for f in reversed(files):
        try:
            with open(f) as fh:
                if (json.load(fh).get("_meta") or {}).get("csrc_sha") == csrc_sha():
                    return f
        except (OSError, ValueError):
            continue
    return files[-1]


def hbm_traffic(kernel, launches_per_step, suffix=""):
    """HBM bytes per launch of the region `kernel` from the committed rocprofv3 PMC summary of this same command
    (profiles/*_hbm_counters.json, written by tools/profile_bench.sh: FETCH_SIZE and WRITE_SIZE in separate --pmc passes,
    kernel-trace only).  Units and gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes: both
    counters are in KiB; FETCH_SIZE tallies the 128-byte requests of coalesced reads at 64 bytes, so the read side is
    multiplied by FETCH_FACTOR.  Calibrated in round 4 (tools/probe/fetch_size_probe.hip under rocprofv3,
    profiles/r04j_fetch_size_probe.json): a streaming read of exactly 1 GiB reports 524,299 KiB at 4, 8 AND 16 bytes per lane and
    for the 18-voxel haloed-row pattern the conv kernels stage with -- the factor is 2.00 whatever the load width (WRITE_SIZE of a
    1 GiB fill: 1,048,576 KiB, factor 1).  The per-voxel GATHER kernels (VecInt, warp) issue 32 / 64-byte requests that the counter
    tallies in full: their FETCH_SIZE equals their input size (31.1 MB for a 31.0 MB field, profiles/r03u_hbm_counters.json), so they
    take factor 1.  A region label names a kernel family ("k<1>"); its template instances ("k<1, 2, 32>", "k<1, 4, 16>") are
    averaged weighted by their dispatch counts.  The file is REFUSED (traffic null, reason in the note) when it does not
    describe this run: no instance of the kernel in it, or a dispatch count that is not launches_per_step x the profiled
    steps recorded in its "_meta" entry.  Returns (bytes_per_launch or None, note)."""
    path = _counter_file("*_hbm_counters%s.json" % suffix)
    if path is None:
        return None, "no profiles/*_hbm_counters%s.json" % suffix
    rel = os.path.relpath(path, ROOT)
    try:
        with open(path) as f:
            ctr = json.load(f)
    except (OSError, ValueError):
        return None, "%s unreadable" % rel
    kernel_ns = kernel.replace(" ", "")
    stem = kernel_ns[:-1] if kernel_ns.endswith(">") else kernel_ns
    while True:            # a region label carries (some of) the template arguments of its kernel, not always in the instance's terms ("k_s3p_conv<1,2>" =
        # NCT, pieces against the instances "k_s3p_conv<1, true>" / "<1, false>" = NCT, BLK): drop trailing arguments until something matches -- never the first
        hits = {k: v for k, v in ctr.items() if k.replace(" ", "") == kernel_ns or k.replace(" ", "").startswith(stem + ",") or k.replace(" ", "").startswith(stem + ">")}
        hits = {k: v for k, v in hits.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
        if hits or "<" not in stem or "," not in stem[stem.index("<"):]:
            break
        stem = stem[:stem.rindex(",")]
    if not hits:
        return None, "%s has no counters for %s (collected on another engine / configuration, or stale: tools/profile_bench.sh)" % (rel, kernel)
    n = sum(v["FETCH_SIZE"]["dispatches"] for v in hits.values())
    sha = (ctr.get("_meta") or {}).get("csrc_sha")
    if sha != csrc_sha():
        return None, "%s was collected on kernel sources %s, this tree is %s: stale profile, re-run tools/profile_bench.sh" % (rel, sha, csrc_sha())
    steps = (ctr.get("_meta") or {}).get("steps_profiled")
    if steps and abs(n - launches_per_step * steps) > 0.5:
        return None, "%s: %d dispatches of %s in %d profiled steps, this run launches %.1f per step: stale profile" % (
            rel, n, kernel, steps, launches_per_step)
    ff = fetch_factor(kernel)
    tot = sum((ff * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"]) * v["FETCH_SIZE"]["dispatches"] for v in hits.values())
    return tot / n * 1024.0, "bytes/launch (%g*FETCH_SIZE + WRITE_SIZE, KiB counters; rocprofv3 --pmc of this command: %s; instances %s)" % (
        ff, rel, ", ".join(sorted(hits)))


def fetch_factor(kernel):
    """FETCH_SIZE -> bytes: 2 for kernels that stream coalesced lines, 1 for the per-voxel gather kernels (see hbm_traffic)"""
    return 1.0 if kernel.startswith(("vecint", "warp3d", "k_vecint", "k_warp3d")) else 2.0


def _counter_field(kernel, field, suffix=""):
    """Mean of a per-kernel field of profiles/*_mfma_counters.json over the template instances a region label names (weighted by their
    dispatch counts), or None -- same staleness rule as hbm_traffic: the summary must have been collected on THESE kernel sources."""
    path = _counter_file("*_mfma_counters%s.json" % suffix)
    if path is None:
        return None
    try:
        with open(path) as f:
            ctr = json.load(f)
    except (OSError, ValueError):
        return None
    if (ctr.get("_meta") or {}).get("csrc_sha") != csrc_sha():
        return None
    kernel_ns = kernel.replace(" ", "")
    stem = kernel_ns[:-1] if kernel_ns.endswith(">") else kernel_ns
    hits = []
    while True:            # a region label carries (some of) the template arguments of its kernel, not always in the instance's terms
        # ("k_s3p_conv<1,2>" = NCT, pieces against the instance "k_s3p_conv<1, true>" = NCT, BLK): drop trailing arguments until something matches
        hits = [v for k, v in ctr.items() if k != "_meta" and (k.replace(" ", "") == kernel_ns or k.replace(" ", "").startswith(stem + ",")
                                                               or k.replace(" ", "").startswith(stem + ">") or k.replace(" ", "") == stem
                                                               or ("<" not in stem and k.replace(" ", "").startswith(stem + "<")))]
        hits = [v for v in hits if field in v]
        if hits or "<" not in stem or "," not in stem[stem.index("<"):]:
            break                      # (the first template argument is never dropped: "k_s3_bwd_weight<3>" is not "k_s3_bwd_weight<2, true>")
        stem = stem[:stem.rindex(",")]
    if not hits:
        return None
    wts = [float((v.get("GRBM_GUI_ACTIVE") or {}).get("dispatches", 1)) for v in hits]
    return sum(v[field] * w for v, w in zip(hits, wts)) / sum(wts)


def shader_clock(kernel, suffix=""):
    """Shader clock the kernel ran at under the profiler (GHz): GRBM_GUI_ACTIVE / 8 XCDs / kernel duration of the same --pmc pass, written
    by tools/rocprof_summary.py into profiles/*_mfma_counters.json."""
    return _counter_field(kernel, "shader_clock_ghz", suffix)


def mfma_util(kernel, suffix=""):
    """Matrix-pipe utilisation of the kernel from the counters: SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)."""
    return _counter_field(kernel, "mfma_util", suffix)


def conv_mfma_util(stats, suffix=""):
    """The north star's "MFMA utilisation of the U-Net convs": counter utilisation of every conv kernel of the per-kernel pass, weighted by the
    time it takes per step (regions with FLOPs = the conv products); None without counters of this tree."""
    rows, tot_ms, tot = {}, 0.0, 0.0
    for name, st in stats.items():
        if not st["flops"]:
            continue
        u = mfma_util(name, suffix)
        if u is None:
            continue
        rows[name] = {"mfma_util": u, "ms": st["ms"]}
        tot_ms += st["ms"]
        tot += u * st["ms"]
    if not rows:
        return None
    conv_ms = sum(st["ms"] for st in stats.values() if st["flops"])
    return {"time_weighted": tot / tot_ms, "covered_fraction_of_conv_time": tot_ms / conv_ms if conv_ms else 0.0,
            "per_kernel": {k: round(v["mfma_util"], 4) for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])},
            "source": "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8) per kernel from the committed rocprofv3 --pmc summary of this tree"}


def binding_roofline(name, st):
    """`roofline` of one kernel region from its totals over the timed steps (`st`: launches, ms, flops, nominal, bytes).
    fp32 kernels are priced on the FLOPs they execute (= the reference formulation's, except the collapsed-upsample kernels, which
    execute fewer), bf16 kernels on the reference formulation's FLOPs (they EXECUTE more: zero-padded channels and K slots are not
    work).  A kernel with both a FLOP and a byte count is priced against the roofline that BINDS it, the one with the larger minimum
    time: a 16-channel bf16 layer moves 64 B per voxel (8 ns per M voxel at 8 TB/s) for 13.8 kFLOP (5.5 ns at 2.5 PFLOP/s) -- an
    HBM kernel that happens to use the MFMA; its MFMA fraction is kept as `mfma_frac`."""
    sec = st["ms"] * 1e-3
    per = {"avg_launch_ms": st["ms"] / st["launches"], "traffic": None}
    hbm = None
    if st["bytes"]:
        ach_b = st["bytes"] / sec / 1e9
        hbm = dict(per, kernel=name, bound="hbm", achieved=ach_b, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach_b / HBM_PEAK_GBS,
                   algorithmic_per_launch=st["bytes"] / st["launches"])
    if not st["flops"]:
        return hbm or dict(per, kernel=name, bound="hbm", achieved=0.0, peak=HBM_PEAK_GBS, unit="GB/s", frac=0.0, algorithmic_per_launch=0.0)
    is_bf = name.startswith("k_bf16")
    is_split = name.startswith(("k_s3_", "k_s3u_", "k_s3p_"))
    alg = st["nominal"] if is_bf else st["flops"]
    # split-fp32 kernels (csrc/conv_s3.hip) run on the 16-bit matrix pipe with `nprod` MFMAs per fp32-equivalent MAC block -- six
    # v_mfma_f32_16x16x32_bf16 (three bf16 pieces, region label ends in ",3>" / "<3>") or three v_mfma_f32_16x16x32_f16 (two fp16 pieces,
    # ",2>" / "<2>"): their ceiling in fp32-equivalent FLOPs is the dense 16-bit peak / nprod -- NOT the fp32-MFMA peak, which they exceed
    nprod = 3.0 if (is_split and (name.rstrip(">").endswith("2") or name.startswith(("k_s3u_bww", "k_s3_bww_pc", "k_s3u_conv_pc")))) else 6.0
    peak = BF16_MFMA_PEAK_TFLOPS if is_bf else (BF16_MFMA_PEAK_TFLOPS / nprod if is_split else FP32_MFMA_PEAK_TFLOPS)
    ach = alg / sec / 1e12
    if hbm is not None and st["bytes"] / (HBM_PEAK_GBS * 1e9) > alg / (peak * 1e12):
        hbm["mfma_frac"] = ach / peak
        return hbm
    out = dict(per, kernel=name, bound="mfma", achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak, algorithmic_per_launch=alg / st["launches"])
    if is_split:
        out["peak_note"] = ("fp32-equivalent FLOPs of a kernel that executes %d 16-bit MFMAs (%s) per fp32 MAC block: peak = 2500 / %d TFLOP/s; the "
                            "same rate is %.2f x the fp32-MFMA peak (157.3)" % (nprod, "fp16 x 2 pieces" if nprod == 3.0 else "bf16 x 3 pieces", nprod,
                                                                               ach / FP32_MFMA_PEAK_TFLOPS))
        out["matrix_pipe_tflops"] = nprod * ach
    return out


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def csrc_sha():
    """Identity of the kernel sources (voxelmorph_amd/buildinfo.py): written into profiles/*_counters.json `_meta` by
    tools/rocprof_summary.py, compared in hbm_traffic() so that counters collected on OTHER kernel sources are refused."""
    sys.path.insert(0, os.path.join(ROOT, "voxelmorph_amd"))
    try:
        import buildinfo                      # by path: no torch / library import needed for a hash
    finally:
        sys.path.pop(0)
    return buildinfo.csrc_sha()


def torch_rocm_worker(shape, int_steps, timed=2):
    """Child process of gpu_baseline(): the reference's ATen call sequence (the oracle's restatement of scripts/torch/train.py:194-223 --
    F.conv3d / max_pool3d / interpolate / grid_sample / the five 9^3 box-filter convs of NCC, torch.optim.Adam over 24 tensors) on cuda:0
    through STOCK torch-ROCm: MIOpen convolutions, ATen elementwise and grid-sampler kernels.  Nothing of voxelmorph_amd is imported."""
    import numpy as np
    from oracle import vxm_oracle as orc
    dev = torch.device("cuda", 0)
    # This image ships no MIOpen kernel / find database for gfx950: with MIOpen enabled every convolution configuration is JIT-compiled on
    # first use (the leg did not finish in 240 s).  VXM_BASELINE_MIOPEN=0 (the default of gpu_baseline()) runs ATen's own convolution path
    # instead (vol2col + rocBLAS GEMM, torch.backends.cudnn.enabled = False) -- still stock torch-ROCm, and said so in the result.
    miopen = os.environ.get("VXM_BASELINE_MIOPEN", "0") == "1"
    torch.backends.cudnn.enabled = miopen
    rng = np.random.default_rng(1234)
    src = torch.from_numpy(rng.random((1, 1) + shape).astype(np.float32)).to(dev)
    trg = torch.from_numpy(rng.random((1, 1) + shape).astype(np.float32)).to(dev)
    sd = {k: v.to(dev) for k, v in orc.seeded_state_dict(shape, seed=0, flow_std=1e-5).items()}
    params = [v.requires_grad_() for v in sd.values()]
    opt = torch.optim.Adam(params, lr=1e-4)

    def step():
        opt.zero_grad()
        loss, _ = orc.train_step_loss(src, trg, sd, "ncc", 1.0, int_steps=int_steps, int_downsize=2)
        loss.backward()
        opt.step()
        return loss

    t0 = time.perf_counter()
    for _ in range(2):                      # warm-up: MIOpen picks its solvers in the first calls
        step()
    torch.cuda.synchronize()
    warm = time.perf_counter() - t0
    torch.cuda.reset_peak_memory_stats()
    times = []
    for _ in range(timed):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    print(json.dumps({"value": 1.0 / dt, "unit": "volume-pairs/s", "kind": "torch-rocm", "ms_per_step": 1e3 * dt,
                      "convolutions": "MIOpen" if miopen else "ATen vol2col + rocBLAS GEMM (torch.backends.cudnn.enabled = False: no MIOpen kernel "
                                                              "database for gfx950 in this image)",
                      "sample": "2 warm-up (%.1f s) + %d timed training steps of the reference's ATen call sequence on cuda:0, B=1, fp32, %s: %s ms"
                                % (warm, timed, "x".join(map(str, shape)), ", ".join("%.1f" % (1e3 * t) for t in times)),
                      "final_loss": float(loss.detach()), "peak_allocated_gb": torch.cuda.max_memory_allocated() / 1e9,
                      "torch": torch.__version__, "hip": torch.version.hip}))


def recorded_gpu_baseline():
    """the newest profiles/*_torch_rocm_baseline.json (a --gpu-baseline run committed by the builder), labelled as recorded"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_torch_rocm_baseline.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    rec["recorded"] = "%s (committed measurement of `bench.py --gpu-baseline`; not re-run in this invocation)" % os.path.relpath(files[-1], ROOT)
    return rec


def gpu_baseline(shape, int_steps, timeout_s=900):
    """SURVEY.md section 8d, second reference point: the reference path on THIS MI355X through stock torch-ROCm, in a child process with a
    deadline (MIOpen's first-call solver search is not ours to bound).  A reported baseline next to `cpu_baseline`, never `value`."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--torch-rocm-baseline-worker", "--shape", ",".join(map(str, shape)),
           "--int-steps", str(int_steps)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": "torch-rocm baseline did not finish in %d s" % timeout_s, "kind": "torch-rocm"}
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or not lines:
        return {"error": "torch-rocm baseline failed (rc %d): %s" % (out.returncode, out.stderr.strip()[-400:]), "kind": "torch-rocm"}
    return json.loads(lines[-1])


def register_extra(vxm, shape, dev, reps=10):
    """`scripts/torch/register.py:87`: model(moving, fixed, registration=True) under torch.no_grad() -- the inference half of the path:
    forward only (U-Net, flow head, resize, VecInt, resize, SpatialTransformer), per-pair latency on one GPU, launches from Python."""
    torch.manual_seed(1234)
    model = vxm.networks.VxmDense(shape, int_steps=7, int_downsize=2).to(dev)
    model.eval()
    src, trg = torch.rand(1, 1, *shape, device=dev), torch.rand(1, 1, *shape, device=dev)
    with torch.no_grad():
        for _ in range(3):
            moved, warp = model(src, trg, registration=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            moved, warp = model(src, trg, registration=True)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    return {"value": 1.0 / dt, "unit": "volume-pairs/s", "ms_per_pair": 1e3 * dt, "host_enqueue_ms_per_pair": 1e3 * host / reps, "reps": reps,
            "dtype": "f32", "workload": "VxmDense 3D %s registration=True forward (register.py:87), int_steps=7, B=1, fp32, torch.no_grad()"
                                        % "x".join(map(str, shape))}


class Workload:
    """One benchmark configuration: model + optimiser + synthetic batch resident in HBM + the training step."""

    def __init__(self, vxm, vdist, name, shape, B, dev, rank, int_steps=None, comm=None, graph=None):
        from voxelmorph_amd.optim import FlatAdam
        self.name, self.B, self.shape = name, B, shape
        self.bf16 = name in ("dense_bf16", "diffeo_bf16")
        self.dense = name == "dense_bf16"
        self.semi = name == "semisup_fp32"
        self.trained = name == "diffeo_fp32_trained_flow"
        self.flow_note = ""
        self.int_steps = (0 if self.dense else 7) if int_steps is None else int_steps
        self.lam = 0.01 if self.dense else 1.0                    # README.md:70: lambda 0.01 with MSE, 1 with NCC
        torch.manual_seed(1234)                                   # identical initial weights on every rank
        if self.semi:                                             # BASELINE.json configs[4]: 30 one-hot labels at half resolution, Dice weight 0.01
            self.model = vxm.networks.VxmDenseSemiSupervisedSeg(shape, 30, int_steps=self.int_steps, int_downsize=2).to(dev)
        else:
            self.model = vxm.networks.VxmDense(shape, int_steps=self.int_steps, int_downsize=2).to(dev)
        self.opt = FlatAdam(self.model, lr=1e-4, comm=comm)
        self.opt.broadcast_params(0)
        torch.manual_seed(1234 + rank)                            # each rank synthesises its own volume pairs in HBM
        self.src = torch.rand(B, 1, *shape, device=dev)
        self.trg = torch.rand(B, 1, *shape, device=dev)
        if self.semi:
            half = tuple(s // 2 for s in shape)
            lab = torch.randint(0, 31, (2, B) + half, device=dev)                  # 0 = background (not among the 30 labels)
            one = lambda l: torch.stack([(l == k) for k in range(1, 31)], 1).float().contiguous()
            self.seg_src, self.seg_trg = one(lab[0]), one(lab[1])
            self.dice = vxm.losses.Dice().loss
        self.img = vxm.losses.MSE().loss if self.dense else vxm.losses.NCC().loss
        self.reg = vxm.losses.Grad("l2", loss_mult=2).loss
        self.wsum = vxm.losses.weighted_sum
        from voxelmorph_amd.pacing import InFlight
        from voxelmorph_amd.graph import GraphedStep
        self.pace = InFlight(2)
        # the step is submitted as ONE hipGraph launch (voxelmorph_amd/graph.py) after two eager steps; VXM_GRAPH=0: every launch from Python
        self.graphed = GraphedStep(self._forward_loss, self.opt, eager_steps=2,
                                   enabled=(os.environ.get("VXM_GRAPH", "1") != "0") if graph is None else bool(graph))
        if self.trained:
            self._make_trained_flow()

    def _make_trained_flow(self):
        """The regime a TRAINED network is in (smooth field, |v| ~ 5 voxels): the random-init flow head predicts
        ~1e-5 voxels, so every gather kernel of the headline step sees a zero field.  Here the flow head gets a bias (a 5 / 3 / 4-voxel
        translation on the half-resolution integration grid) and weights rescaled until the spatially varying part has a standard
        deviation of 0.25 half-resolution voxels: in the last scaling-and-squaring steps every voxel moves by more than one voxel."""
        m = self.model
        with torch.no_grad():
            _, _, vel, disp, _ = m._forward_all(self.src, self.trg)
            std = float((vel - vel.mean(dim=(2, 3, 4), keepdim=True)).std())
            m.flow.weight.mul_(0.25 / max(std, 1e-30))
            m.flow.bias.copy_(torch.tensor([10.0, -6.0, 8.0], device=m.flow.bias.device))       # full-resolution voxels: halved by the resize
            from voxelmorph_amd import invalidate_packs
            invalidate_packs(m)
            _, _, vel, disp, _ = m._forward_all(self.src, self.trg)
            self.flow_note = ("flow head rescaled: velocity on the half-resolution grid mean (%.2f, %.2f, %.2f), std of the varying part %.3f, "
                              "integrated displacement max %.2f full-resolution voxels"
                              % (*[float(v) for v in vel.mean(dim=(0, 2, 3, 4))], float((vel - vel.mean(dim=(2, 3, 4), keepdim=True)).std()),
                                 float(disp.abs().max())))

    def _forward_loss(self):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.bf16):      # bf16: blocked-bf16 activations between the convs
            if self.semi:
                y, pre, yseg = self.model(self.src, self.trg, self.seg_src)
                return self.wsum([self.img(self.trg, y), self.reg(None, pre), self.dice(self.seg_trg, yseg)], [1.0, self.lam, 0.01])
            y, pre = self.model(self.src, self.trg)
            # train.py:205-212 `loss += loss_function(...) * weights[n]`: one launch of this package (losses.weighted_sum), not two ATen ones per term
            return self.wsum([self.img(self.trg, y), self.reg(None, pre)], [1.0, self.lam])

    def step(self):
        # One step = zero_grad + forward + loss + backward + (all-reduce) + Adam, submitted as one hipGraph launch (GraphedStep; eagerly --
        # launch by launch from Python -- for the first two steps, under the per-kernel timer, and with VXM_GRAPH=0).
        # At most two steps in flight (voxelmorph_amd/pacing.py): the host waits for the end of step k - 2 before it enqueues step k.  The
        # GPU never starves, and in eager mode the caching allocator reaches its steady state within the warm-up (with an unbounded
        # run-ahead the timed region holds more blocks in flight than the warm-up ever did, and hipMallocs of 10-15 ms land inside it).
        from voxelmorph_amd import profiler
        self.pace.wait()
        loss = self.graphed.eager() if profiler.ACTIVE is not None else self.graphed()
        self.pace.mark()
        return loss

    def describe(self):
        sh = "x".join(map(str, self.shape))
        if self.dense:
            return ("VxmDense 3D %s, int_steps=%d (CVPR dense), MSE + 0.01 Grad(l2,x2), bf16 activations / fp32 accumulate, fp32 master "
                    "weights, Adam lr 1e-4, %d pair(s)/GPU (BASELINE.json configs[1])" % (sh, self.int_steps, self.B))
        if self.bf16:
            return ("VxmDense 3D %s, int_steps=%d diffeomorphic (int_downsize=2), NCC(9^3)+Grad(l2,x2), bf16 activations / fp32 accumulate "
                    "in the U-Net (everything else fp32), Adam lr 1e-4, %d pair(s)/GPU" % (sh, self.int_steps, self.B))
        if self.semi:
            return ("VxmDenseSemiSupervisedSeg 3D %s, int_steps=%d, 30 one-hot labels at half resolution, NCC(9^3)+Grad(l2,x2)+0.01 Dice, "
                    "fp32, Adam lr 1e-4, %d pair(s)/GPU (BASELINE.json configs[4], per-GPU step)" % (sh, self.int_steps, self.B))
        if getattr(self, "trained", False):
            return ("VxmDense 3D %s, int_steps=%d diffeomorphic (int_downsize=2), NCC(9^3)+Grad(l2,x2), fp32, Adam lr 1e-4, %d pair(s)/GPU; the "
                    "headline step with the field of a TRAINED network (SURVEY section 8d): %s" % (sh, self.int_steps, self.B, self.flow_note))
        return ("VxmDense 3D %s, int_steps=%d diffeomorphic (int_downsize=2), NCC(9^3)+Grad(l2,x2), fp32, Adam lr 1e-4, %d pair(s)/GPU "
                "(BASELINE.json configs[2])" % (sh, self.int_steps, self.B))


def timed_steps(wl, steps, vdist, dev, timer=None):
    """EXACTLY `steps` steps between barrier + synchronize on both sides; MAX over ranks.  timer: bracket every C-ABI launch
    with HIP events on the launch stream (per-kernel pass); None: nothing but the steps is inside the region (the `value` pass)."""
    from voxelmorph_amd import profiler
    if timer is not None:
        profiler.install(timer)
    vdist.barrier()
    torch.cuda.synchronize()
    wl.pace.wait_s = 0.0
    ms0 = torch.cuda.memory_stats(dev)
    replays0 = wl.graphed.replays
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = wl.step()
    # the host's share: Python + launches of the region, without the time it spent waiting for step k - 2 (Workload.step keeps two in flight)
    timed_steps.host_enqueue_s = time.perf_counter() - t0 - wl.pace.wait_s
    torch.cuda.synchronize()
    vdist.barrier()
    elapsed = vdist.max_over_ranks(time.perf_counter() - t0, dev)
    if timer is not None:
        profiler.uninstall()
    ms1 = torch.cuda.memory_stats(dev)
    # what the caching allocator did inside the region (hipMalloc / hipFree calls stall the host for 10+ ms each) and how the steps were submitted
    timed_steps.submission = {
        "graph_replays": wl.graphed.replays - replays0, "eager_steps": steps - (wl.graphed.replays - replays0),
        "capture_error": wl.graphed.capture_error,
        # the dynamic-range guard of the fp16-piece engine ran on the first (eager) step of this workload: worst share of a tensor's values in the
        # absolute-error regime, and the engine the process continued on (voxelmorph_amd/graph.py; off with VXM_RANGE_GUARD=0)
        "range_guard": None if wl.graphed.range_report is None else {
            "worst_share_below_2^-18_of_tile_max": (wl.graphed.range_report["worst"] or {}).get("share_below_2^-18_of_tile_max"),
            "share_limit": wl.graphed.range_report["share_limit"], "engine": wl.graphed.range_report["recommended_engine"]},
        "device_allocs": ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0),
        "device_frees": ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0),
        "alloc_retries": ms1.get("num_alloc_retries", 0) - ms0.get("num_alloc_retries", 0),
        "allocator_calls": ms1.get("allocation.all.allocated", 0) - ms0.get("allocation.all.allocated", 0),
        "reserved_gb": ms1.get("reserved_bytes.all.current", 0) / 1e9}
    return elapsed, float(loss.detach())


# Vector-ALU issue roofline of the fused NCC march (csrc/losses.hip k_ncc_fused_fwd/bwd<4>): NOT an HBM kernel, although SURVEY.md section 8d
# lists it as one -- its 55 / 83 MB of algorithmic traffic would take 7 / 10 us, its arithmetic (direct 9-tap box sums of five products in
# three directions, no running sums: fp32 accuracy) takes ten times that.  Wave-instructions per (block, slice) counted in the gfx950 ISA of
# this tree (static count of the march's body + its two inner loops -- staging, and the W pass with its 16-byte LDS reads -- at 3 / 3 and 3 / 2
# trips): forward 4 waves x 327 = 1308 VALU (+ 176 LDS; before the 16-byte reads and the hoisted staging addresses: 417 + 94 per wave),
# backward 4 x 168 = 672 (+ 136 LDS; before: 248 + 53); a wave-instruction occupies its SIMD for 4 cycles (the model that matched the VecInt gather within
# 25 %); 1024 SIMDs at 2.4 GHz.  (block, slice) pairs of a launch: columns of 8 x 32 pixels x depth segments x
# (segment + 8 halo slices), csrc/losses.hip ncc_segment.
NCC_VALU_PER_BLOCK_SLICE = {"ncc_fwd": 1308.0, "ncc_bwd": 672.0}


def ncc_valu_roofline(name, st, shape, B):
    D, H, W = shape
    cols = ((W + 31) // 32) * ((H + 7) // 8) * B
    seg = 40
    while seg > 20 and cols * ((D + seg - 1) // seg) < 1024:
        seg = (seg + 1) // 2
    block_slices = cols * ((D + seg - 1) // seg) * (seg + 8)
    instr = NCC_VALU_PER_BLOCK_SLICE[name] * block_slices
    min_ms = instr * 4.0 / 1024.0 / 2.4e9 * 1e3
    ms = st["ms"] / st["launches"]
    return {"bound": "valu", "valu_wave_instructions_per_launch": instr, "valu_min_ms": min_ms, "valu_frac": min_ms / ms,
            "note": "vector-ALU issue roofline (4 cycles per wave-instruction, 1024 SIMDs, 2.4 GHz); the HBM figure (gbs) is kept for reference"}


def kernel_table(stats, steps, shape=None, B=1):
    kernels = {}
    for name, st in stats.items():
        ent = {"launches_per_step": st["launches"] / steps, "ms_per_step": st["ms"] / steps, "avg_launch_ms": st["ms"] / st["launches"]}
        if st["flops"]:
            ent["tflops"] = st["flops"] / (st["ms"] * 1e-3) / 1e12                # FLOPs the kernel executes (MFMA utilisation)
            if st.get("nominal", 0.0) > st["flops"] * 1.001:                      # collapsed-upsample kernels: reference formulation
                ent["nominal_tflops"] = st["nominal"] / (st["ms"] * 1e-3) / 1e12
        if st["bytes"]:
            ent["gbs"] = st["bytes"] / (st["ms"] * 1e-3) / 1e9
        if name in NCC_VALU_PER_BLOCK_SLICE and shape is not None and len(shape) == 3:
            ent.update(ncc_valu_roofline(name, st, shape, B))
        kernels[name] = ent
    return kernels


def comm_evidence(opt, dev):
    """What a multi-rank run leaves behind about its exchange (the scaling run is the driver's, not the builder's): which library
    carries the all-reduce, how many ranks its communicator spans, and the median of 20 timed all-reduces of the 1.31 MB bucket."""
    import statistics
    from voxelmorph_amd.optim import FlatAdam
    tbackend = torch.distributed.get_backend() if torch.distributed.is_initialized() else "none"
    ev = {"backend": "libvxm_comm" if opt.comm is not None else "torch.distributed(%s%s)" % (
              tbackend, ", bucket staged through the host" if tbackend == "gloo" and opt.flat_grad.is_cuda else ""),
          "ranks_seen": opt.world, "bucket_bytes": 4 * opt.n}
    if opt.comm is None and tbackend == "nccl" and os.environ.get("VXM_COMM", "") != "torch":
        from voxelmorph_amd import comm as vcomm
        ev["native_error"] = vcomm.NativeComm.last_failure or "native communicator not attempted (backend %s)" % (
            torch.distributed.get_backend() if torch.distributed.is_initialized() else "none")
    if opt.comm is not None:
        from voxelmorph_amd import comm as vcomm
        ev["ranks_seen"] = int(vcomm.lib().vxm_comm_world())
        ev["rccl_version"] = vcomm.rccl_version()
    buf = torch.zeros_like(opt.flat_grad)
    sync = torch.cuda.synchronize if buf.is_cuda else (lambda: None)         # (the CPU test drives this over gloo)
    times = []
    for i in range(25):
        sync()
        t0 = time.perf_counter()
        FlatAdam.all_reduce_sum(opt, buf)          # (unbound: the CPU test passes a stand-in with the same attributes)
        sync()
        if i >= 5:
            times.append((time.perf_counter() - t0) * 1e6)
    ev["allreduce_us"] = statistics.median(times)
    ev["transport"] = _transport_note()
    return ev


def _transport_note():
    """How the ranks are wired, from what this process can see without a debug re-run: NCCL_DEBUG=INFO lines RCCL wrote to the
    file named by NCCL_DEBUG_FILE (bench.py sets both for multi-rank runs), reduced to the channel / transport lines."""
    path = os.environ.get("NCCL_DEBUG_FILE", "")
    path = path.replace("%h", os.uname().nodename).replace("%p", str(os.getpid()))
    try:
        with open(path) as f:
            lines = [l.strip() for l in f if (" via " in l or "Connected all" in l or "xGMI" in l.upper() or "XGMI" in l)]
    except OSError:
        return "no RCCL debug file"
    kinds = sorted({l.split(" via ", 1)[1].split()[0] for l in lines if " via " in l})
    return {"via": kinds, "sample": lines[:3]}


def main():
    args = parse()
    if args.torch_rocm_baseline_worker:
        torch_rocm_worker(tuple(int(v) for v in args.shape.split(",")), 7 if args.int_steps is None else args.int_steps)
        return
    from voxelmorph_amd import dist as vdist
    # `python bench.py --gpus N` without a torchrun environment: become `python -m torch.distributed.run --nproc-per-node N
    # ... bench.py --gpus N ...` (one rank per GPU, 127.0.0.1 rendezvous); under torchrun this returns at once
    vdist.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # transport evidence of the multi-rank run (comm_evidence): RCCL's INFO lines of the init phase go to a per-process file
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH")
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/vxm_rccl_%h_%p.log")
    rank, local, world = vdist.init_from_env()
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X (HIP device); there is no CPU fallback for the product path", file=sys.stderr)
        sys.exit(2)
    import voxelmorph_amd as vxm
    from voxelmorph_amd import profiler
    from voxelmorph_amd.torch import functional as VF

    dev = torch.device("cuda", local)
    shape = tuple(int(s) for s in args.shape.split(","))
    B = args.batch_per_gpu
    # Multi-rank: the libvxm_comm.so communicator carries the exchange.  When it cannot be built (or fails its known-answer self-check)
    # every rank raises (they agree on the decision): a scaling run never silently measures another exchange than the one it reports.
    # VXM_COMM=torch asks for torch.distributed's RCCL instead (reported in `comm.backend`); a job whose torch.distributed backend is not
    # 'nccl' (VXM_DIST_BACKEND=gloo: ranks sharing one device in the 1-GPU tests) has no native communicator to build.
    comm = vdist.native_comm(required=world > 1 and os.environ.get("VXM_COMM", "") != "torch")
    wl = Workload(vxm, vdist, args.config, shape, B, dev, rank, args.int_steps, comm)
    args.int_steps = wl.int_steps
    bf16, dense = wl.bf16, wl.dense

    for _ in range(args.warmup):
        wl.step()
    # Python's cyclic GC would otherwise run its first full (generation-2) collection somewhere inside the timed
    # region: a one-off ~40 ms host stall (measured) that stops kernel submission.  Collect now and move the survivors
    # to the permanent generation; the GC stays enabled (a training loop does the same after its first steps).
    import gc
    gc.collect()
    gc.freeze()
    # pass 1 -- `value`: the K timed steps and nothing else (no event bracketing; library defaults: the weight-gradient launches share the
    # chip with the backward-data chain on the second stream)
    elapsed, final_loss = timed_steps(wl, args.steps, vdist, dev)
    host_ms = timed_steps.host_enqueue_s / args.steps * 1e3
    submission = timed_steps.submission
    # pass 2 -- per-kernel table and `roofline`: every C-ABI launch bracketed by HIP events on the launch stream, the full-resolution
    # launches serialised (one kernel on the chip at a time, so that a launch's duration is its own)
    # every launch on ONE stream (the weight-gradient launches of the coarse levels too: a launch that shares the chip with the other stream's
    # kernels would be timed longer than rocprofv3's serial pass times it -- VERDICT round 4: 455 vs 387 us on the dominant kernel)
    keep_overlap = VF.OVERLAP_SMALL_LEVELS
    VF.OVERLAP_SMALL_LEVELS = False
    wl.graphed.eager()                       # the capture emptied the caching allocator's pool: one eager step outside the pass re-populates it
    timer = profiler.KernelTimer()
    ksteps = min(args.steps, 10)
    elapsed_k, _ = timed_steps(wl, ksteps, vdist, dev, timer)
    VF.OVERLAP_SMALL_LEVELS = keep_overlap
    stats = timer.resolve()
    comm_ev = comm_evidence(wl.opt, dev) if world > 1 else None

    extra = {}
    if world == 1 and args.config == "diffeo_fp32" and not args.no_extra_configs:
        # the other BASELINE.json configs that fit one GPU, a few steps each, in the SAME line (labelled; none of them is `value`)
        del wl
        torch.cuda.empty_cache()
        for key, name, eb, esteps in (("dense_bf16", "dense_bf16", 1, 8), ("diffeo_fp32_4_pairs_per_gpu", "diffeo_fp32", 4, 4),
                                      ("semisup_fp32", "semisup_fp32", 1, 8), ("diffeo_fp32_trained_flow", "diffeo_fp32_trained_flow", 1, 8),
                                      ("diffeo_fp32_eager_submission", "diffeo_fp32", 1, 12),
                                      ("diffeo_fp32_bf16x3_engine", "diffeo_fp32", 1, 8),
                                      ("diffeo_fp32_native_engine", "diffeo_fp32", 1, 8)):
            engine = VF.FP32_ENGINE
            try:
                if key == "diffeo_fp32_native_engine":     # the headline workload on the exact-fp32 MFMA kernels of rounds 1-2, for comparison
                    if engine == "native":
                        continue
                    VF.FP32_ENGINE = "native"
                if key == "diffeo_fp32_bf16x3_engine":     # ... and on the three-piece bf16 split of round 3
                    if engine == "split":
                        continue
                    VF.FP32_ENGINE = "split"
                # (the headline step submitted launch by launch from Python instead of as one hipGraph launch: faster by ~2 % on a warm host --
                # the replayed graph's branches overlap less -- and slower on a cold one; skipped when the headline itself ran that way)
                if key == "diffeo_fp32_eager_submission" and os.environ.get("VXM_GRAPH", "1") == "0":
                    continue
                w2 = Workload(vxm, vdist, name, shape, eb, dev, rank, graph=False if key == "diffeo_fp32_eager_submission" else None)
                # warm-up = the timed pattern itself (esteps steps enqueued back to back): the caching allocator only reaches its steady
                # state under the run-ahead of the real loop (blocks held by the side stream's pending events are not reusable yet); with two
                # synchronous warm-up steps a hipMalloc of several hundred ms could land inside the timed region
                timed_steps(w2, esteps, vdist, dev)
                t2, l2 = timed_steps(w2, esteps, vdist, dev)
                host2 = 1e3 * timed_steps.host_enqueue_s / esteps
                sub2 = timed_steps.submission
                VF.OVERLAP_SMALL_LEVELS = False            # per-kernel pass: one stream, as the headline's
                w2.graphed.eager()
                tm = profiler.KernelTimer()
                timed_steps(w2, 2, vdist, dev, tm)
                VF.OVERLAP_SMALL_LEVELS = keep_overlap
                st2 = tm.resolve()
                d2 = max(st2, key=lambda k: st2[k]["ms"])
                roof2 = binding_roofline(d2, st2[d2])
                # counter traffic of the dominant kernel: the committed summaries were collected at ONE pair per GPU on the default engine
                # (fp32 headline) and on the bf16 config; they describe an extra config only when its launches are those launches
                if eb != 1:
                    roof2["traffic"], roof2["traffic_unit"] = None, "the committed counters were collected at 1 pair per GPU: launches of %d pairs move other byte counts" % eb
                else:
                    roof2["traffic"], roof2["traffic_unit"] = hbm_traffic(d2, st2[d2]["launches"] / 2.0, "_bf16" if w2.bf16 else "")
                    roof2["mfma_util"] = mfma_util(d2, "_bf16" if w2.bf16 else "")
                extra[key] = {"value": eb * esteps / t2, "unit": "volume-pairs/s", "ms_per_step": 1e3 * t2 / esteps, "steps": esteps,
                              "dtype": "bf16" if w2.bf16 else "f32", "workload": w2.describe(), "final_loss": l2,
                              "host_enqueue_ms_per_step": host2, "submission": sub2, "roofline": roof2}
                if w2.trained:                             # the HBM-bound kernels in the regime they are in after training, with their own table
                    kt = kernel_table(st2, 2, shape, eb)
                    hb = {k: v for k, v in kt.items() if k.startswith(("warp3d", "vecint", "resize3d", "ncc", "gradloss"))}
                    stv = [k for k in st2 if k.startswith(("warp3d", "vecint"))]
                    b_ = sum(st2[k]["bytes"] for k in stv)
                    ms_ = sum(st2[k]["ms"] for k in stv)
                    extra[key]["kernels_hbm"] = hb
                    extra[key]["spatial_transformer_plus_vecint"] = {"algorithmic_bytes_per_step": b_ / 2, "ms_per_step": ms_ / 2,
                                                                     "gbs": b_ / (ms_ * 1e-3) / 1e9 if ms_ else 0.0,
                                                                     "frac_of_hbm_peak": (b_ / (ms_ * 1e-3) / 1e9) / HBM_PEAK_GBS if ms_ else 0.0}
                del w2
                torch.cuda.empty_cache()
            except Exception as exc:                       # an extra line must never take the headline down with it
                extra[key] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            finally:
                VF.FP32_ENGINE = engine
                VF.OVERLAP_SMALL_LEVELS = keep_overlap

        try:
            extra["register_fp32"] = register_extra(vxm, shape, dev)
        except Exception as exc:
            extra["register_fp32"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        torch.cuda.empty_cache()

    if rank != 0:
        return
    kernels = kernel_table(stats, ksteps, shape, B)
    dom = max(stats, key=lambda k: stats[k]["ms"])
    ds = stats[dom]
    roof = binding_roofline(dom, ds)
    roof["traffic"], roof["traffic_unit"] = hbm_traffic(dom, ds["launches"] / ksteps, "_bf16" if bf16 else "")
    roof["measured_in"] = "per-kernel pass: %d steps with HIP-event bracketing, %.3f ms/step (the `value` pass runs un-instrumented)" % (
        ksteps, 1e3 * elapsed_k / ksteps)
    roof["shader_clock_ghz"] = shader_clock(dom, "_bf16" if bf16 else "")     # under the profiler's counter pass; null without a matching profile
    roof["mfma_util"] = mfma_util(dom, "_bf16" if bf16 else "")               # counter utilisation of the matrix pipe, beside `frac`
    stv = [k for k in stats if k.startswith(("warp3d", "vecint"))]
    st_bytes, st_ms = sum(stats[k]["bytes"] for k in stv), sum(stats[k]["ms"] for k in stv)
    # the same sum with the two ResizeTransform launches that remain (x0.5 forward / backward) -- what the round-5 verdict added up (604 us then)
    st_ms_rs = st_ms + sum(stats[k]["ms"] for k in stats if k.startswith("resize3d"))
    out = {
        "metric": "volume-pairs/sec VxmDense 160x192x224 int_steps=0 MSE train (bf16 activations)" if dense
                  else ("volume-pairs/sec VxmDense 160x192x224 int_steps=7 NCC train (bf16 activations; not the fp32 headline)" if bf16
                        else "volume-pairs/sec VxmDense 160x192x224 int_steps=7 NCC train"),
        "value": world * B * args.steps / elapsed, "unit": "volume-pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
        "config": {"workload": wl_desc(args, shape, B), "global_batch": world * B, "parallelism": "dp%d" % world,
                   "fp32_engine": VF.fp32_engine_note()},
        "roofline": roof, "kernels": kernels, "final_loss": final_loss,
        # SpatialTransformer + VecInt against the HBM roofline (north star: >= 50 %); on the near-zero field of a random-init flow head --
        # extra_configs.diffeo_fp32_trained_flow carries the same figure on a 5-voxel field
        "spatial_transformer_plus_vecint": {"algorithmic_bytes_per_step": st_bytes / ksteps, "ms_per_step": st_ms / ksteps,
                                            "gbs": st_bytes / (st_ms * 1e-3) / 1e9 if st_ms else 0.0,
                                            "frac_of_hbm_peak": (st_bytes / (st_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if st_ms else 0.0,
                                            "ms_per_step_with_resize": st_ms_rs / ksteps,
                                            "note": "round 6: fullsize + the final warp are one kernel, so the bytes are those of the FUSED formulation "
                                                    "(no full-resolution pos_flow: 667 MB per pair, was 719 + 372 of resize traffic); the fraction of a "
                                                    "fused path falls when bytes are removed faster than time -- compare ms_per_step_with_resize across rounds"},
        "host_enqueue_ms_per_step": host_ms,        # rank 0's host time per step of the value pass (hipGraphLaunch, or Python + launches when eager), without the pacing wait
        "submission": submission,                   # value pass: graph replays vs eager steps, caching-allocator activity inside the timed region
    }
    cmu = conv_mfma_util(stats, "_bf16" if bf16 else "")
    if cmu is not None:
        out["conv_mfma_util"] = cmu
    if comm_ev is not None:
        out["comm"] = comm_ev
    if extra:
        out["extra_configs"] = extra
    if world == 1 and not args.no_gpu_baseline and args.config == "diffeo_fp32":
        torch.cuda.empty_cache()
        gb = gpu_baseline(shape, args.int_steps) if args.gpu_baseline else recorded_gpu_baseline()
        if gb is not None:
            out["gpu_baseline"] = gb
            if "value" in gb:
                gb["speedup_of_value"] = out["value"] / gb["value"]
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(shape, args.int_steps, args.cpu_baseline_steps, args.cpu_threads,
                                           "mse" if dense else "ncc", 0.01 if dense else 1.0)
    print(json.dumps(out))


def wl_desc(args, shape, B):
    w = Workload.__new__(Workload)
    w.name, w.B, w.shape, w.int_steps = args.config, B, shape, args.int_steps
    w.bf16, w.dense, w.semi = args.config in ("dense_bf16", "diffeo_bf16"), args.config == "dense_bf16", args.config == "semisup_fp32"
    return w.describe()


if __name__ == "__main__":
    main()
