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
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--shape", type=str, default="160,192,224")
    ap.add_argument("--int-steps", type=int, default=7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-shape", type=str, default="160,192,112")   # half of the volume: ~6 s per CPU step on 32 threads
    return ap.parse_args()


def cpu_baseline(full_shape, sample_shape, int_steps):
    """The oracle (CPU restatement of the reference torch path, oracle/vxm_oracle.py) timed on this host's
    cores on a bounded sample: the SAME network/loss/optimizer on a sub-volume; throughput is scaled by the
    voxel ratio to volume-pairs/s at the full shape.  Reported baseline, not the optimisation target."""
    import numpy as np
    from oracle import vxm_oracle as orc
    # torch's intra-op pool degrades badly beyond one socket's worth of threads (256 threads on the
    # 2x64-core bench host ran 10x slower than 8 threads elsewhere): use at most 32 and report that number
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    rng = np.random.default_rng(1234)
    src = torch.from_numpy(rng.random((1, 1) + sample_shape).astype(np.float32))
    trg = torch.from_numpy(rng.random((1, 1) + sample_shape).astype(np.float32))
    sd = orc.seeded_state_dict(sample_shape, seed=0, flow_std=1e-5)
    params = [v.requires_grad_() for v in sd.values()]
    opt = torch.optim.Adam(params, lr=1e-4)

    def step():
        opt.zero_grad()
        loss, _ = orc.train_step_loss(src, trg, sd, "ncc", 1.0, int_steps=int_steps, int_downsize=2)
        loss.backward()
        opt.step()

    step()                                  # warm-up
    t0 = time.perf_counter()
    step()
    dt = time.perf_counter() - t0
    frac = float(np.prod(sample_shape)) / float(np.prod(full_shape))
    return {
        "value": frac / dt, "unit": "volume-pairs/s", "cores": cores, "kind": "port",
        "sample": "1 warm-up + 1 timed training step (fwd+NCC+Grad+bwd+Adam) of the torch-CPU oracle on a %s "
                  "sub-volume (%.4f of %s voxels), %.2f s/step; pairs/s scaled by the voxel ratio"
                  % ("x".join(map(str, sample_shape)), frac, "x".join(map(str, full_shape)), dt),
        "cpu_model": _cpu_model(),
    }


def hbm_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this same command
    (profiles/*_hbm_counters.json, written by tools/profile_bench.sh: FETCH_SIZE and WRITE_SIZE in separate --pmc
    passes, kernel-trace only).  Units and gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes:
    both counters are in KiB; FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so the
    read side is doubled.  Returns (bytes_per_launch or None, source)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_counters.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            ctr = json.load(f)
    except (OSError, ValueError):
        return None, None
    ent = ctr.get(kernel)
    if ent is None and kernel.endswith(">"):          # region label "k<1>" vs the full template argument list "k<1, 4>"
        stem = kernel[:-1]
        hits = [v for k, v in ctr.items() if k.startswith(stem + ",") or k.startswith(stem + ">")]
        ent = hits[0] if len(hits) == 1 else None
    if not ent or "FETCH_SIZE" not in ent or "WRITE_SIZE" not in ent:
        return None, None
    return (2.0 * ent["FETCH_SIZE"]["mean"] + ent["WRITE_SIZE"]["mean"]) * 1024.0, os.path.relpath(files[-1], ROOT)


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
    rank, local, world = vdist.init_from_env()
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world),
                  file=sys.stderr)
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
    torch.manual_seed(1234)                                   # identical initial weights on every rank
    model = vxm.networks.VxmDense(shape, int_steps=args.int_steps, int_downsize=2).to(dev)
    opt = FlatAdam(model, lr=1e-4, comm=vdist.native_comm())      # VXM_COMM=rccl: direct libvxm_comm.so all-reduce
    opt.broadcast_params(0)
    torch.manual_seed(1234 + rank)                            # each rank synthesises its own volume pairs in HBM
    src = torch.rand(B, 1, *shape, device=dev)
    trg = torch.rand(B, 1, *shape, device=dev)
    ncc = vxm.losses.NCC().loss
    reg = vxm.losses.Grad("l2", loss_mult=2).loss

    def step():
        opt.zero_grad()
        y, pre = model(src, trg)
        loss = ncc(trg, y) + 1.0 * reg(None, pre)
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
    final_loss = float(loss)

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
    if ds["flops"]:
        ach = ds["flops"] / (ds["ms"] * 1e-3) / 1e12
        roof = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                "algorithmic_per_launch": ds["flops"] / ds["launches"], "avg_launch_ms": ds["ms"] / ds["launches"]}
    else:
        ach = ds["bytes"] / (ds["ms"] * 1e-3) / 1e9
        roof = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_per_launch": ds["bytes"] / ds["launches"], "avg_launch_ms": ds["ms"] / ds["launches"]}
    roof["traffic"], src = hbm_traffic(dom)
    if src:
        roof["traffic_unit"] = "bytes/launch (2*FETCH_SIZE + WRITE_SIZE, KiB counters; rocprofv3 --pmc of this command: %s)" % src
    out = {
        "metric": "volume-pairs/sec VxmDense 160x192x224 int_steps=7 NCC train",
        "value": world * B * args.steps / elapsed, "unit": "volume-pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "VxmDense 3D %s, int_steps=%d diffeomorphic (int_downsize=2), NCC(9^3)+Grad(l2,x2), fp32, Adam "
                               "lr 1e-4, %d pair(s)/GPU (BASELINE.json configs[2])" % ("x".join(map(str, shape)), args.int_steps, B),
                   "global_batch": world * B, "parallelism": "dp%d" % world},
        "roofline": roof, "kernels": kernels, "final_loss": final_loss,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(shape, tuple(int(s) for s in args.cpu_baseline_shape.split(",")), args.int_steps)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
