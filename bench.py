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
    ap.add_argument("--config", choices=["diffeo_fp32", "dense_bf16", "diffeo_bf16"], default="diffeo_fp32",
                    help="diffeo_fp32 = BASELINE.json configs[2], the headline metric (default); dense_bf16 = configs[1]: int_steps=0, "
                         "MSE + 0.01 Grad, bf16 activations / fp32 accumulate under torch.autocast; diffeo_bf16 = the headline network "
                         "and losses with bf16 activations (an extra, labelled line: NOT the headline metric, which is fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", type=int, default=2)               # timed CPU steps after one warm-up, at the FULL shape
    ap.add_argument("--cpu-threads", type=int, default=0)                      # 0: min(32, cores), see cpu_baseline()
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


def hbm_traffic(kernel, launches_per_step, suffix=""):
    """HBM bytes per launch of the region `kernel` from the committed rocprofv3 PMC summary of this same command
    (profiles/*_hbm_counters.json, written by tools/profile_bench.sh: FETCH_SIZE and WRITE_SIZE in separate --pmc passes,
    kernel-trace only).  Units and gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes: both
    counters are in KiB; FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so the read side is
    doubled.  A region label names a kernel family ("k<1>"); its template instances ("k<1, 2, 32>", "k<1, 4, 16>") are
    averaged weighted by their dispatch counts.  The file is REFUSED (traffic null, reason in the note) when it does not
    describe this run: no instance of the kernel in it, or a dispatch count that is not launches_per_step x the profiled
    steps recorded in its "_meta" entry.  Returns (bytes_per_launch or None, note)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_counters%s.json" % suffix)))
    if not files:
        return None, "no profiles/*_hbm_counters%s.json" % suffix
    rel = os.path.relpath(files[-1], ROOT)
    try:
        with open(files[-1]) as f:
            ctr = json.load(f)
    except (OSError, ValueError):
        return None, "%s unreadable" % rel
    kernel_ns = kernel.replace(" ", "")
    stem = kernel_ns[:-1] if kernel_ns.endswith(">") else kernel_ns
    hits = {k: v for k, v in ctr.items() if k.replace(" ", "") == kernel_ns or k.replace(" ", "").startswith(stem + ",")}
    hits = {k: v for k, v in hits.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    if not hits:
        return None, "%s has no counters for %s: stale profile, re-run tools/profile_bench.sh" % (rel, kernel)
    n = sum(v["FETCH_SIZE"]["dispatches"] for v in hits.values())
    steps = (ctr.get("_meta") or {}).get("steps_profiled")
    if steps and abs(n - launches_per_step * steps) > 0.5:
        return None, "%s: %d dispatches of %s in %d profiled steps, this run launches %.1f per step: stale profile" % (
            rel, n, kernel, steps, launches_per_step)
    tot = sum((2.0 * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"]) * v["FETCH_SIZE"]["dispatches"] for v in hits.values())
    return tot / n * 1024.0, "bytes/launch (2*FETCH_SIZE + WRITE_SIZE, KiB counters; rocprofv3 --pmc of this command: %s; instances %s)" % (
        rel, ", ".join(sorted(hits)))


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
    alg = st["nominal"] if is_bf else st["flops"]
    peak = BF16_MFMA_PEAK_TFLOPS if is_bf else FP32_MFMA_PEAK_TFLOPS
    ach = alg / sec / 1e12
    if hbm is not None and st["bytes"] / (HBM_PEAK_GBS * 1e9) > alg / (peak * 1e12):
        hbm["mfma_frac"] = ach / peak
        return hbm
    return dict(per, kernel=name, bound="mfma", achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak,
                algorithmic_per_launch=alg / st["launches"])


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse()
    from voxelmorph_amd import dist as vdist
    # `python bench.py --gpus N` without a torchrun environment: become `python -m torch.distributed.run --nproc-per-node N
    # ... bench.py --gpus N ...` (one rank per GPU, 127.0.0.1 rendezvous); under torchrun this returns at once
    vdist.self_launch(args.gpus, os.path.abspath(__file__), sys.argv[1:])
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
    from voxelmorph_amd.optim import FlatAdam

    dev = torch.device("cuda", local)
    shape = tuple(int(s) for s in args.shape.split(","))
    B = args.batch_per_gpu
    bf16 = args.config in ("dense_bf16", "diffeo_bf16")
    dense = args.config == "dense_bf16"
    if args.int_steps is None:
        args.int_steps = 0 if dense else 7
    lam = 0.01 if dense else 1.0                              # README.md:70: lambda 0.01 with MSE, 1 with NCC
    torch.manual_seed(1234)                                   # identical initial weights on every rank
    model = vxm.networks.VxmDense(shape, int_steps=args.int_steps, int_downsize=2).to(dev)
    opt = FlatAdam(model, lr=1e-4, comm=vdist.native_comm())      # VXM_COMM=rccl: direct libvxm_comm.so all-reduce
    opt.broadcast_params(0)
    torch.manual_seed(1234 + rank)                            # each rank synthesises its own volume pairs in HBM
    src = torch.rand(B, 1, *shape, device=dev)
    trg = torch.rand(B, 1, *shape, device=dev)
    ncc = vxm.losses.MSE().loss if dense else vxm.losses.NCC().loss
    reg = vxm.losses.Grad("l2", loss_mult=2).loss

    def step():
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):      # bf16: blocked-bf16 activations between the convs
            y, pre = model(src, trg)
            loss = ncc(trg, y) + lam * reg(None, pre)
        loss.backward()
        opt.step()                                            # all-reduce (world>1) + fused Adam
        return loss

    for _ in range(args.warmup):
        step()
    # Python's cyclic GC would otherwise run its first full (generation-2) collection somewhere inside the timed
    # region: a one-off ~40 ms host stall (measured) that stops kernel submission.  Collect now and move the survivors
    # to the permanent generation; the GC stays enabled (a training loop does the same after its first steps).
    import gc
    gc.collect()
    gc.freeze()
    timer = profiler.KernelTimer()
    profiler.install(timer)
    vdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    vdist.barrier()
    elapsed = vdist.max_over_ranks(time.perf_counter() - t0, dev)
    profiler.uninstall()
    stats = timer.resolve()
    final_loss = float(loss.detach())

    if rank != 0:
        return
    kernels = {}
    for name, st in stats.items():
        ent = {"launches_per_step": st["launches"] / args.steps, "ms_per_step": st["ms"] / args.steps,
               "avg_launch_ms": st["ms"] / st["launches"]}
        if st["flops"]:
            ent["tflops"] = st["flops"] / (st["ms"] * 1e-3) / 1e12                # FLOPs the kernel executes (MFMA utilisation)
            if st.get("nominal", 0.0) > st["flops"] * 1.001:                      # collapsed-upsample kernels: reference formulation
                ent["nominal_tflops"] = st["nominal"] / (st["ms"] * 1e-3) / 1e12
        if st["bytes"]:
            ent["gbs"] = st["bytes"] / (st["ms"] * 1e-3) / 1e9
        kernels[name] = ent
    dom = max(stats, key=lambda k: stats[k]["ms"])
    ds = stats[dom]
    roof = binding_roofline(dom, ds)
    roof["traffic"], roof["traffic_unit"] = hbm_traffic(dom, ds["launches"] / args.steps, "_bf16" if bf16 else "")
    out = {
        "metric": "volume-pairs/sec VxmDense 160x192x224 int_steps=0 MSE train (bf16 activations)" if dense
                  else ("volume-pairs/sec VxmDense 160x192x224 int_steps=7 NCC train (bf16 activations; not the fp32 headline)" if bf16
                        else "volume-pairs/sec VxmDense 160x192x224 int_steps=7 NCC train"),
        "value": world * B * args.steps / elapsed, "unit": "volume-pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
        "config": {"workload": ("VxmDense 3D %s, int_steps=%d (CVPR dense), MSE + 0.01 Grad(l2,x2), bf16 activations / fp32 accumulate, "
                                "fp32 master weights, Adam lr 1e-4, %d pair(s)/GPU (BASELINE.json configs[1])" if dense else
                                "VxmDense 3D %s, int_steps=%d diffeomorphic (int_downsize=2), NCC(9^3)+Grad(l2,x2), bf16 activations / fp32 "
                                "accumulate in the U-Net (everything else fp32), Adam lr 1e-4, %d pair(s)/GPU" if bf16 else
                                "VxmDense 3D %s, int_steps=%d diffeomorphic (int_downsize=2), NCC(9^3)+Grad(l2,x2), fp32, Adam "
                                "lr 1e-4, %d pair(s)/GPU (BASELINE.json configs[2])") % ("x".join(map(str, shape)), args.int_steps, B),
                   "global_batch": world * B, "parallelism": "dp%d" % world},
        "roofline": roof, "kernels": kernels, "final_loss": final_loss,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(shape, args.int_steps, args.cpu_baseline_steps, args.cpu_threads,
                                           "mse" if dense else "ncc", lam)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
