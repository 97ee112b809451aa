"""bf16-activation engine of the U-Net (+ flow head): BASELINE.json configs[1] (VxmDense 160x192x224, int_steps=0, MSE + Grad, bf16).

Same plan, same parameters, same autograd contract as `functional.UnetFn`, but every tensor BETWEEN the convolutions — activations
forward, activation gradients backward — lives in HBM as channel-blocked bf16 `[B][C/8][D][H][W][8]` and the convolutions run on
`v_mfma_f32_16x16x32_bf16` with fp32 accumulation (csrc/conv_bf16.hip).  What stays fp32: the network inputs and the returned
tensor (the flow field the fp32 SpatialTransformer / VecInt / losses consume — coordinates must stay fp32), the
master weights and biases, the parameter gradients (written straight into FlatAdam's bucket) and the optimiser.

It is selected the way a torch user asks for it: `with torch.autocast('cuda', dtype=torch.bfloat16): model(src, trg)`
(or `voxelmorph_amd.torch.functional_bf16.set_activation_dtype('bf16')` process-wide).  Replaces what autocast would send to
MIOpen for `ConvBlock` / flow conv / MaxPool3d / Upsample + cat (voxelmorph/torch/networks.py:83-85,122-144,211,257,290-305).
"""
import ctypes

import os

import torch

from .. import _lib
from .. import profiler as _prof
from .._lib import call, ptr, require_device, stream
from .functional import _claim_sink, _dims, _vol3, _Workspace, _c

_FORCED = None


def set_activation_dtype(name):
    """'bf16' routes every 3-D U-Net through this engine, 'fp32' never does, None (default) follows torch.autocast."""
    global _FORCED
    if name not in (None, "bf16", "fp32"):
        raise ValueError("activation dtype must be 'bf16', 'fp32' or None, got %r" % (name,))
    _FORCED = name


def enabled():
    if _FORCED is not None:
        return _FORCED == "bf16"
    return torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16


# VXM_PACK_CACHE=0: never reuse a packed operator (every launch re-packs, the behaviour of rounds 1-2 on the fp32 path).  For training
# loops that write parameter VALUES behind the version counter (`p.data.copy_()`, `vector_to_parameters`, an EMA swap) and cannot call
# `invalidate_packs`: costs one small pack launch per conv launch.
PACK_CACHE = os.environ.get("VXM_PACK_CACHE", "1") != "0"
_NOCACHE_TICK = [0]


def _ver(t):
    """Cache key of the packed copies of a parameter (and of the hand-made in-place check): its version counter, the address of its
    storage (`p.data = other` re-seats it without touching the counter) and a generation number bumped by `invalidate_packs`.
    Inference tensors (a model built or loaded under torch.inference_mode()) have no version counter; they cannot be modified in
    place by autograd-visible code either.  A write through `.data` INTO the same storage is invisible to all three: such writers
    call `voxelmorph_amd.invalidate_packs(model)` (or run with VXM_PACK_CACHE=0)."""
    gen = t.__dict__.get("_vxm_pack_gen", 0)
    if not PACK_CACHE:
        _NOCACHE_TICK[0] += 1
        gen = ("nocache", _NOCACHE_TICK[0])          # never equal to a stored key
    if t.is_inference():
        return ("inference", t.data_ptr(), gen)
    return (t._version, t.data_ptr(), gen)


def _inplace_ver(t):
    """what the hand-made in-place-modification check of the fused engine compares between forward and backward"""
    gen = t.__dict__.get("_vxm_pack_gen", 0)
    return ("inference", t.data_ptr(), gen) if t.is_inference() else (t._version, t.data_ptr(), gen)


def invalidate_packs(params):
    """Drop the packed (bf16 / split-fp32) operators cached on `params` (an iterable of parameters, or a module).  Needed after
    parameter VALUES were changed without bumping the version counter -- `p.data.copy_(...)`, `vector_to_parameters`, an EMA weight
    swap, collectives into the flat buffer (`FlatAdam.broadcast_params` calls it itself).  Ordinary in-place updates (`p.add_`, every
    torch optimiser, `load_state_dict`, `FlatAdam.step`) bump the counter and need nothing."""
    if hasattr(params, "parameters"):
        params = params.parameters()
    for p in params:
        p.__dict__["_vxm_pack_gen"] = p.__dict__.get("_vxm_pack_gen", 0) + 1


def _blocked(B, C, dims, dev):
    return torch.empty((B, C // 8) + tuple(dims) + (8,), dtype=torch.bfloat16, device=dev)


def _pad16(c):
    return (c + 15) // 16 * 16


def pack_weights(w, ci_lo, ci_n, flip):
    """fp32 `[Cout][Cin][3][3][3]` -> bf16 MFMA-fragment order of the forward operator on input channels [ci_lo, ci_lo+ci_n), or of
    its adjoint onto that range (flip).  Cached per parameter until its version counter moves (i.e. once per optimiser step)."""
    cache = w.__dict__.setdefault("_vxm_bf16_packs", {})
    key = (ci_lo, ci_n, bool(flip))
    hit = cache.get(key)
    if hit is not None and hit[0] == _ver(w) and hit[1].device == w.device:
        return hit[1]
    cout, cin = w.shape[:2]
    inc, outc = (cout, ci_n) if flip else (ci_n, cout)
    wp = torch.empty(_lib.lib().vxm_bf16_conv_packed_bytes(inc, outc), dtype=torch.uint8, device=w.device)
    call("vxm_bf16_conv_pack_weights", ptr(_c(w)), cin, cout, ci_lo, ci_n, 1 if flip else 0, ptr(wp), stream())
    cache[key] = (_ver(w), wp)
    return wp


def prepack(jobs):
    """Pack every stale operator of `jobs` = [(w, ci_lo, ci_n, flip), ...] in ONE launch (`vxm_bf16_conv_pack_weights_batch`); the
    per-layer `pack_weights` calls of the step then hit the cache.  After an optimiser step all ~27 packed copies of a VxmDense
    U-Net are stale at once."""
    stale = []
    for w, ci_lo, ci_n, flip in jobs:
        cache = w.__dict__.setdefault("_vxm_bf16_packs", {})
        key = (ci_lo, ci_n, bool(flip))
        hit = cache.get(key)
        if hit is not None and hit[0] == _ver(w) and hit[1].device == w.device:
            continue
        cout, cin = w.shape[:2]
        inc, outc = (cout, ci_n) if flip else (ci_n, cout)
        nbytes = _lib.lib().vxm_bf16_conv_packed_bytes(inc, outc)
        wp = hit[1] if hit is not None and hit[1].device == w.device and hit[1].numel() == nbytes else \
            torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        stale.append((_c(w), cin, cout, ci_lo, ci_n, bool(flip), wp, cache, key, _ver(w)))
    if not stale:
        return
    table = (_lib.Bf16PackJob * len(stale))()
    for j, (w, cin, cout, ci_lo, ci_n, flip, wp, _, _, _) in enumerate(stale):
        table[j] = _lib.Bf16PackJob(w.data_ptr(), wp.data_ptr(), cin, cout, ci_lo, ci_n, 1 if flip else 0)
    call("vxm_bf16_conv_pack_weights_batch", ctypes.cast(table, ctypes.c_void_p), len(stale), stream())
    for _, _, _, _, _, _, wp, cache, key, ver in stale:
        cache[key] = (ver, wp)


def _pack_jobs(plan, params, cin0, with_backward, input_grads):
    """the operators one pass over `plan` uses: the forward one of every conv and, for a training step, its adjoints per segment"""
    jobs = []
    for op in plan.ops:
        if op["kind"] != "conv":
            continue
        w = params[2 * op["k"]]
        jobs.append((w, 0, w.shape[1], False))
        if not with_backward:
            continue
        s0, _, s1 = op["src"]
        if s0 < plan.n_inputs:
            if input_grads:
                jobs.append((w, 0, cin0, True))
            continue
        jobs.append((w, 0, plan.ch[s0], True))
        if s1 is not None:
            jobs.append((w, plan.ch[s0], plan.ch[s1], True))
    return jobs


_WALK_BACK = [False]       # direction of the last full-resolution conv launch (snake order: see functional.SNAKE)


def conv(x0, c0, up0, x1, c1, wp, bias, y, cout, planar, slope, mask, mask_slope, B, D, H, W):
    nct = 1 if (planar or cout <= 16) else 2
    V = B * D * H * W
    from . import functional as VF
    rev = 0
    if VF.SNAKE and D * H * W >= (1 << 21):          # consecutive full-resolution launches walk their tensors in alternating directions
        _WALK_BACK[0] = not _WALK_BACK[0]
        rev = 2 if _WALK_BACK[0] else 0
    # algorithmic bytes: every operand element read once (the upsampled segment at its own resolution), every result written once
    nbytes = 2.0 * (c0 * (V // 8 if up0 else V) + c1 * V) + (4.0 if planar else 2.0) * cout * V + (2.0 * cout * V if mask is not None else 0.0)
    with _prof.region("k_bf16_conv<%d,%d,%d>" % (nct, 6 if nct == 2 else 8, 1 if planar else 0),
                      flops=2.0 * 27 * (c0 + c1) * (16 * nct * ((cout + 16 * nct - 1) // (16 * nct))) * B * D * H * W,
                      nbytes=nbytes, nominal=2.0 * 27 * (c0 + c1) * cout * B * D * H * W):
        call("vxm_bf16_conv_fwd", ptr(x0), c0, 1 if up0 else 0, ptr(x1), c1, ptr(wp), ptr(bias), ptr(y), cout, (1 if planar else 0) | rev,
             float(slope), ptr(mask), float(mask_slope), B, D, H, W, stream())


def conv_bwd_weight(ws, x0, c0, up0, x1, c1, dz, cdz, gw, gb, B, D, H, W):
    cout_w, cin_w = gw.shape[:2]
    need = _lib.lib().vxm_bf16_conv_bwd_weight_workspace_bytes(c0 + c1, cdz, B, D, H, W)
    buf = ws.get(need)
    with _prof.region("k_bf16_conv_bwd_weight<%d>" % (cdz // 16), flops=2.0 * 28 * (c0 + c1) * cdz * B * D * H * W,
                      nominal=2.0 * 27 * cin_w * cout_w * B * D * H * W):
        call("vxm_bf16_conv_bwd_weight", ptr(x0), c0, 1 if up0 else 0, ptr(x1), c1, ptr(dz), cdz, ptr(gw), cin_w, cout_w, ptr(gb),
             ptr(buf), buf.numel(), B, D, H, W, stream())


class UnetBf16Fn(torch.autograd.Function):
    """`functional.UnetFn` with blocked-bf16 activations.  forward(plan, *inputs fp32, *params fp32) -> fp32 planar tensor."""

    @staticmethod
    def forward(ctx, plan, *tensors):
        inputs = [_c(t) for t in tensors[:plan.n_inputs]]
        params = tensors[plan.n_inputs:]
        require_device(*inputs)
        require_device(*params)
        for t in inputs:
            _vol3(t, "Unet")
        B = inputs[0].shape[0]
        shape3 = tuple(inputs[0].shape[2:])
        nlev = max(plan.lvl.values())
        if any(s % (1 << nlev) for s in shape3):
            raise ValueError("Unet: volume %s must be divisible by %d (MaxPool floors and the skip concat "
                             "would not line up, networks.py:130,138)" % (shape3, 1 << nlev))
        if any(op["kind"] == "cat" for op in plan.ops):
            raise NotImplementedError("bf16 engine: a U-Net that ends on a concat is not built (use the fp32 engine)")
        dev = inputs[0].device
        for i, t in enumerate(inputs):
            if t.shape[1] != plan.ch[i] or t.shape[0] != B or tuple(t.shape[2:]) != shape3:
                raise ValueError("Unet: input %d has shape %s, expected [%d,%d,%s]" % (i, tuple(t.shape), B, plan.ch[i], shape3))
        if len(inputs) > 2:
            raise NotImplementedError("bf16 engine: one or two input tensors")
        if any(ctx.needs_input_grad[1:]):
            # vxm_bf16_conv_bwd_weight holds the taps of at most 32 output channels in a wave's accumulators: refuse HERE, before
            # any gradient sink of the optimiser has been claimed, not half-way through loss.backward()
            wide = [c for (_, c, _) in plan.convs if c > 32]
            if wide:
                raise NotImplementedError("bf16 engine: training needs ConvBlocks of at most 32 output features (got %s); "
                                          "run this network on the fp32 engine (no autocast)" % wide)
        V = shape3[0] * shape3[1] * shape3[2]
        cin0 = sum(plan.ch[i] for i in range(plan.n_inputs))
        xin = _blocked(B, _pad16(cin0), shape3, dev)             # virtual concat of the inputs, blocked, zero-padded to 16 channels
        x1 = inputs[1] if len(inputs) == 2 else None
        call("vxm_bf16_to_blocked", ptr(inputs[0]), plan.ch[0], inputs[0][0].numel(), ptr(x1), plan.ch[1] if x1 is not None else 0,
             x1[0].numel() if x1 is not None else 0, ptr(xin), _pad16(cin0), B, V, stream())
        prepack(_pack_jobs(plan, params, cin0, any(ctx.needs_input_grad[1:]), any(ctx.needs_input_grad[1:1 + plan.n_inputs])))
        T = {}                  # tensor id -> blocked bf16 activation (ids of the inputs map to `xin`)
        out = None
        last = len(plan.ops) - 1
        for n, op in enumerate(plan.ops):
            dst = op["dst"]
            D, H, W = _dims(shape3, plan.lvl[dst])
            if op["kind"] == "conv":
                s0, up0, s1 = op["src"]
                w, b = params[2 * op["k"]], params[2 * op["k"] + 1]
                cout = plan.ch[dst]
                if s0 < plan.n_inputs:
                    x0, c0, x1b, c1 = xin, _pad16(cin0), None, 0
                else:
                    x0, c0 = T[s0], plan.ch[s0]
                    x1b, c1 = (T[s1], plan.ch[s1]) if s1 is not None else (None, 0)
                wp = pack_weights(w, 0, w.shape[1], False)
                planar = n == last and cout <= 4 and op["slope"] == 1.0
                if planar:
                    y = torch.empty((B, cout, D, H, W), dtype=torch.float32, device=dev)
                elif cout % 16:
                    raise NotImplementedError("bf16 engine: feature counts must be multiples of 16, got %d" % cout)
                else:
                    y = _blocked(B, cout, (D, H, W), dev)
                conv(x0, c0, up0, x1b, c1, wp, b, y, cout, planar, op["slope"], None, 1.0, B, D, H, W)
                if planar:
                    out = y
                else:
                    T[dst] = y
            else:               # pool
                src = T[op["src"]]
                sD, sH, sW = _dims(shape3, plan.lvl[op["src"]])
                y = _blocked(B, plan.ch[dst], (D, H, W), dev)
                call("vxm_bf16_maxpool2_fwd", ptr(src), ptr(y), B, plan.ch[dst], sD, sH, sW, stream())
                T[dst] = y
        ctx.planar_out = out is not None
        if out is None:         # a bare U-Net: hand the last activation back as fp32 NCDHW
            D, H, W = _dims(shape3, plan.lvl[plan.out])
            out = torch.empty((B, plan.ch[plan.out], D, H, W), dtype=torch.float32, device=dev)
            call("vxm_bf16_from_blocked", ptr(T[plan.out]), plan.ch[plan.out], ptr(out), plan.ch[plan.out], B, D * H * W, stream())
        ctx.plan, ctx.T, ctx.xin, ctx.params, ctx.shape3, ctx.B, ctx.cin0 = plan, T, xin, params, shape3, B, cin0
        ctx.versions = [_inplace_ver(t) for t in params]
        return out

    @staticmethod
    def backward(ctx, gout):
        plan, params, shape3, B, xin, cin0 = ctx.plan, ctx.params, ctx.shape3, ctx.B, ctx.xin, ctx.cin0
        if ctx.T is None:
            raise RuntimeError("UnetBf16Fn: backward a second time: the saved activations were released by the first pass")
        if [_inplace_ver(t) for t in params] != ctx.versions:
            raise RuntimeError("UnetBf16Fn: a parameter was modified in place between forward and backward "
                               "(e.g. an optimizer step before loss.backward()); gradients would be wrong")
        T, ctx.T, ctx.xin = ctx.T, None, None
        dev = gout.device
        gout = _c(gout)
        ws = _Workspace(dev)
        # weight gradients on the second HIP stream beside the backward-data chain they do not feed (as UnetFn.backward of the fp32 engine,
        # functional.py: the tails of the big launches fill with each other's blocks); VXM_NO_OVERLAP=1: one stream
        from . import functional as VF
        main = torch.cuda.current_stream(dev)
        side = VF._side_stream(dev) if VF.OVERLAP_SMALL_LEVELS else None
        n_in = plan.n_inputs
        grads = [None] * (n_in + len(params))
        DZ, GP, GS = {}, {}, {}

        def lrelu_bwd(g, y, slope):
            if slope == 1.0:
                return g
            dz = torch.empty_like(g)
            call("vxm_bf16_lrelu_bwd", ptr(g), ptr(y), ptr(dz), float(slope), g.numel(), stream())
            return dz

        out_op = plan.ops[plan.producer[plan.out]]
        D, H, W = _dims(shape3, plan.lvl[plan.out])
        cout = plan.ch[plan.out]
        g_blk = _blocked(B, _pad16(cout), (D, H, W), dev)
        call("vxm_bf16_to_blocked", ptr(gout), cout, gout[0].numel(), None, 0, 0, ptr(g_blk), _pad16(cout), B, D * H * W, stream())
        DZ[plan.out] = g_blk if ctx.planar_out else lrelu_bwd(g_blk, T[plan.out], out_op["slope"])

        deferred_dw = None
        try:
            for n in range(len(plan.ops) - 1, -1, -1):
                if deferred_dw is not None:
                    deferred_dw()
                    deferred_dw = None
                op = plan.ops[n]
                dst = op["dst"]
                D, H, W = _dims(shape3, plan.lvl[dst])
                if op["kind"] == "pool":
                    src = op["src"]
                    sD, sH, sW = _dims(shape3, plan.lvl[src])
                    C = plan.ch[src]
                    prod = plan.ops[plan.producer[src]]
                    dz = _blocked(B, C, (sD, sH, sW), dev)
                    call("vxm_bf16_maxpool2_bwd", ptr(T[src]), ptr(GP.pop(dst)), ptr(GS.pop(src, None)), ptr(dz),
                         float(prod["slope"] if prod["kind"] == "conv" else 1.0), B, C, sD, sH, sW, stream())
                    DZ[src] = dz
                    continue
                s0, up0, s1 = op["src"]
                w, b = params[2 * op["k"]], params[2 * op["k"] + 1]
                cout = plan.ch[dst]
                dz = DZ.pop(dst)
                cdz = dz.shape[1] * 8
                feeds_inputs = s0 < n_in
                if feeds_inputs:
                    x0, c0, x1b, c1 = xin, _pad16(cin0), None, 0
                else:
                    x0, c0 = T[s0], plan.ch[s0]
                    x1b, c1 = (T[s1], plan.ch[s1]) if s1 is not None else (None, 0)
                gw_sink, gb_sink = _claim_sink(w), _claim_sink(b)
                gw = gw_sink if gw_sink is not None else torch.empty_like(w)
                gb = gb_sink if gb_sink is not None else torch.empty_like(b)
                if side is not None:
                    ev = torch.cuda.Event()
                    ev.record(main)

                    def launch_dw(ev=ev, x0=x0, c0=c0, up0=up0, x1b=x1b, c1=c1, dz=dz, cdz=cdz, gw=gw, gb=gb, gw_sink=gw_sink, gb_sink=gb_sink, D=D, H=H, W=W):
                        side.wait_event(ev)
                        with torch.cuda.stream(side):
                            conv_bwd_weight(ws, x0, c0, up0, x1b, c1, dz, cdz, gw, gb, B, D, H, W)      # (ws is only ever used on the second stream then)
                        dz.record_stream(side)              # released by the main-stream chain before the second stream may be done
                        for g_, sink in ((gw, gw_sink), (gb, gb_sink)):
                            if sink is None:
                                g_.record_stream(side)
                    # under graph capture the weight gradient is enqueued behind this layer's backward-data launches (functional.DW_ORDER)
                    if VF.DW_ORDER == "after" or (VF.DW_ORDER != "before" and torch.cuda.is_current_stream_capturing()):
                        deferred_dw = launch_dw
                    else:
                        launch_dw()
                else:
                    conv_bwd_weight(ws, x0, c0, up0, x1b, c1, dz, cdz, gw, gb, B, D, H, W)
                grads[n_in + 2 * op["k"]] = None if gw_sink is not None else gw
                grads[n_in + 2 * op["k"] + 1] = None if gb_sink is not None else gb
                if feeds_inputs:
                    if any(ctx.needs_input_grad[1 + i] for i in range(n_in)):
                        gx = _blocked(B, _pad16(cin0), (D, H, W), dev)
                        conv(dz, cdz, False, None, 0, pack_weights(w, 0, cin0, True), None, gx, _pad16(cin0), False, 1.0, None, 1.0, B, D, H, W)
                        full = torch.empty((B, cin0, D, H, W), dtype=torch.float32, device=dev)
                        call("vxm_bf16_from_blocked", ptr(gx), _pad16(cin0), ptr(full), cin0, B, D * H * W, stream())
                        lo = 0
                        for i in range(n_in):
                            grads[i] = full[:, lo:lo + plan.ch[i]]
                            lo += plan.ch[i]
                    continue
                # ---- backward-data per segment (the forward kernel with the adjoint weights of that channel range)
                c0r = plan.ch[s0]
                prod0 = plan.ops[plan.producer[s0]]
                single = len(plan.consumers[s0]) == 1
                if up0:
                    lD, lH, lW = D // 2, H // 2, W // 2
                    if prod0["kind"] != "conv" or not single:
                        raise NotImplementedError("bf16 engine: upsampled tensors come from a decoder ConvBlock with one consumer")
                    dzl = _blocked(B, c0r, (lD, lH, lW), dev)
                    # adjoint conv + adjoint of the upsampling + leaky_relu_backward of the producer in one kernel: the full-resolution
                    # gradient of the upsampled segment (440 MB at the top level) is never written
                    nct = 1 if c0r <= 16 else 2
                    with _prof.region("k_bf16_conv<%d,%d,2>" % (nct, 6 if nct == 2 else 8), flops=2.0 * 27 * cdz * c0r * B * D * H * W,
                                      nbytes=2.0 * B * D * H * W * (cdz + c0r / 4.0), nominal=2.0 * 27 * cdz * c0r * B * D * H * W):
                        call("vxm_bf16_conv_bwd_data_up", ptr(dz), cdz, ptr(pack_weights(w, 0, c0r, True)), ptr(dzl), c0r,
                             ptr(T[s0]) if prod0["slope"] != 1.0 else None, float(prod0["slope"]), B, D, H, W, stream())
                    DZ[s0] = dzl
                elif prod0["kind"] == "conv" and single:
                    gx = _blocked(B, c0r, (D, H, W), dev)
                    fuse = prod0["slope"] != 1.0
                    conv(dz, cdz, False, None, 0, pack_weights(w, 0, c0r, True), None, gx, c0r, False, 1.0, T[s0] if fuse else None,
                         prod0["slope"], B, D, H, W)
                    DZ[s0] = gx
                else:
                    gx = _blocked(B, c0r, (D, H, W), dev)
                    conv(dz, cdz, False, None, 0, pack_weights(w, 0, c0r, True), None, gx, c0r, False, 1.0, None, 1.0, B, D, H, W)
                    if prod0["kind"] == "pool":
                        GP[s0] = gx
                    else:
                        DZ[s0] = lrelu_bwd(gx, T[s0], prod0["slope"])
                if s1 is not None:
                    c1r = plan.ch[s1]
                    gs = _blocked(B, c1r, (D, H, W), dev)
                    conv(dz, cdz, False, None, 0, pack_weights(w, c0r, c1r, True), None, gs, c1r, False, 1.0, None, 1.0, B, D, H, W)
                    if any(plan.ops[m]["kind"] == "pool" for m in plan.consumers[s1]):
                        GS[s1] = gs
                    else:
                        DZ[s1] = lrelu_bwd(gs, T[s1], plan.ops[plan.producer[s1]]["slope"])
            if deferred_dw is not None:
                deferred_dw()
        finally:
            # also on an exception half-way (the GraphedStep fallback goes straight on to zero_grad / Adam): later main-stream work must not race
            # with second-stream launches that are still writing the flat gradient bucket -- as UnetFn.backward
            if side is not None:
                main.wait_stream(side)              # parameter gradients (and the activations the second stream read) are final past this point
        return (None,) + tuple(grads)
