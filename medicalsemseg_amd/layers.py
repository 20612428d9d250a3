"""Layer primitives with explicit forward/backward over the C ABI (``hip.py``).

Activations are channels-last ``[N, D, H, W, C]`` tensors (possibly channel slices of a wider
concat buffer) in the model's compute dtype (bf16 or fp32).  Parameters stay ordinary fp32
``nn.Parameter``s in the torch layouts (state-dict compatible with MONAI / the reference); every
forward repacks them into the MFMA operand images (cached while weights are unchanged).

Each primitive returns what its backward needs in a small tuple; the model-level
``torch.autograd.Function`` (see ``models/unet.py``) strings them together, so one autograd node
covers the whole network and every buffer / accumulation is decided here, not by autograd.
"""
from __future__ import annotations

import os

from typing import Optional

import torch

from . import hip

# bumped by optimisers that update parameters behind torch's back (FlatAdamW): invalidates packed weights
weights_epoch = 0


def bump_weights_epoch():
    global weights_epoch
    weights_epoch += 1


class _PackRegistry:
    """Every packed-weight image built through a PackedCache, grouped by (device, compute dtype).  When any image is
    found stale (its parameter changed: optimizer step, load_state_dict, in-place edit) ALL images of the group are
    rebuilt by ONE msseg_pack_weights_batch launch instead of one launch per image."""

    def __init__(self):
        self.groups = {}

    def add(self, device, dtype, src, dst, job):
        g = self.groups.setdefault((device, dtype), {"jobs": [], "table": None, "keep": []})
        rec = {"src": src, "dst": dst, "job": job, "state": (src._version, weights_epoch), "alive": True}
        g["jobs"].append(rec)
        g["table"] = None
        return rec

    def drop(self, device, dtype, rec):
        g = self.groups.get((device, dtype))
        if g is not None and rec["alive"]:
            rec["alive"] = False
            g["jobs"] = [r for r in g["jobs"] if r["alive"]]
            g["table"] = None

    def _ensure_table(self, g, device):
        jobs = g["jobs"]
        if g["table"] is None:
            arr = (hip.PackJob * len(jobs))(*[r["job"] for r in jobs])
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            if g.get("last") is not None:
                g["keep"].append(g["last"])   # a captured hipGraph may still reference the previous table
            g["table"] = g["last"] = host.to(device)
            g["max_total"] = max(int(r["job"].total) for r in jobs)
            g["sum_total"] = sum(int(r["job"].total) for r in jobs)

    def prepare(self):
        """build every missing job table now (host-to-device copies are not allowed while a stream is capturing)"""
        for (device, _), g in self.groups.items():
            if g["jobs"]:
                self._ensure_table(g, device)

    def refresh_if_stale(self, device=None):
        """eager check for callers that replay graphs captured WITHOUT the repack (inference): one batched repack per
        stale (device, dtype) group, nothing when the parameters did not change"""
        for (dev, dtype), g in self.groups.items():
            if device is not None and dev != device:
                continue
            if any(r["state"] != (r["src"]._version, weights_epoch) for r in g["jobs"]):
                self.refresh(dev, dtype)

    def refresh(self, device, dtype):
        g = self.groups[(device, dtype)]
        jobs = g["jobs"]
        self._ensure_table(g, device)
        hip.pack_weights_batch(g["table"], len(jobs), g["max_total"], g["sum_total"], dtype)
        for r in jobs:
            r["state"] = (r["src"]._version, weights_epoch)


PACK_REGISTRY = _PackRegistry()


class PackedCache:
    """Packed-weight images of one parameter, rebuilt when the parameter changed."""

    def __init__(self):
        self._val = {}

    def __del__(self):
        try:
            for (dtype, _), rec in self._val.items():
                for r in [rec] + rec.get("extra", []):
                    PACK_REGISTRY.drop(r["src"].device, dtype, r)
        except Exception:   # interpreter shutdown
            pass

    def get(self, p: torch.Tensor, dtype, kind: str, builder):
        k = (dtype, kind)
        rec = self._val.get(k)
        if rec is not None and (rec["src"].data_ptr() != p.data_ptr() or rec["src"].device != p.device):
            for r in [rec] + rec.get("extra", []):
                PACK_REGISTRY.drop(r["src"].device, dtype, r)   # the parameter moved: rebuild from scratch
            rec = None
        if rec is None:
            hip.LAST_PACK_JOBS.clear()
            v = builder()
            jobs = list(hip.LAST_PACK_JOBS)     # one image, or the sub-images of a composite one (pack_conv_k3_c48)
            assert jobs and jobs[0][1].data_ptr() == v.data_ptr()
            recs = [PACK_REGISTRY.add(p.device, dtype, src, dst, job) for src, dst, job in jobs]
            rec = recs[0]
            rec["extra"], rec["value"] = recs[1:], v
            self._val[k] = rec
            return v
        if rec["state"] != (rec["src"]._version, weights_epoch):
            PACK_REGISTRY.refresh(p.device, dtype)
        return rec["value"]


class _WgradSide:
    """Weight-gradient kernels on a second HIP stream.

    In the backward pass only the input-gradient chain is serial; every weight gradient depends on tensors that exist
    when its layer is reached and is needed only by the optimizer.  The whole-network backward functions open a
    section (`begin`), the layers then issue their wgrad kernels through `run` on the side stream (after the main
    stream's work so far), and `join` makes the main stream wait for them.  The full-chip wgrad kernels of the
    high-resolution levels then overlap the many small kernels of the deep levels instead of queueing behind them.
    The side stream owns the wgrad workspace (all wgrad launches are serial on it); tensors handed to it are kept
    alive until the join."""

    INLINE, SIDE, DEFER = 0, 1, 2

    def __init__(self):
        self.streams = {}
        self.active = None
        self.mode = self.INLINE
        self.keep = []
        self.deferred = []

    def begin(self, device):
        # Measured on MI355X (round 1, UNet 96^3 B=2): 5.38 ms/step with the side stream vs 5.22 ms without -- the
        # persistent wgrad workgroups take a CU's whole register file, so nothing co-resides and the small kernels only
        # queue behind them.  Kept as an opt-in experiment.
        if not torch.cuda.is_available() or not os.environ.get("MSSEG_WGRAD_STREAM"):
            return
        st = self.streams.get(device)
        if st is None:
            st = self.streams[device] = torch.cuda.Stream(device=device)
        self.active = st
        self.mode = self.INLINE

    def set_mode(self, mode):
        """INLINE: on the calling stream (full-chip layers with nothing small to overlap); SIDE: on the side stream
        right away (small layers); DEFER: collected and issued to the side stream by flush() -- full-chip wgrads that
        should run under the small kernels of the deep levels rather than against the next full-chip dgrad."""
        if self.active is not None:
            if mode == self.INLINE and self.mode != self.INLINE:
                # back on the calling stream: its weight-gradient launches share the slab workspace with the side
                # stream's, so everything issued there must have finished first
                self.flush()
                torch.cuda.current_stream().wait_stream(self.active)
            self.mode = mode

    def _issue(self, fn):
        st = self.active
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            fn()

    def run(self, fn, *tensors):
        if self.active is None or self.mode == self.INLINE:
            return fn()
        self.keep.extend(tensors)
        if self.mode == self.DEFER:
            self.deferred.append(fn)
        else:
            self._issue(fn)

    def flush(self):
        if self.active is not None:
            for fn in self.deferred:
                self._issue(fn)
        self.deferred = []

    def join(self):
        self.flush()
        st, self.active = self.active, None
        self.mode = self.INLINE
        if st is not None:
            torch.cuda.current_stream().wait_stream(st)
        self.keep.clear()


WGRAD_SIDE = _WgradSide()


# set by a network's inference-only forward (nothing kept for a backward: models/unet.py _forward_cl(keep=False)) around its
# layer calls: enables forms that only pay when no backward follows (Conv3.halves_ok: K-split of the 24^3-level convs)
INFERENCE_FORWARD = False


class _Applied:
    """what Conv3.bwd returns in place of the InstanceNorm-backward sums when its finish kernel already ran the receiving
    unit's whole backward (small-grid path): the tensor it returned IS that unit's dy"""

    def __repr__(self):
        return "APPLIED"


APPLIED = _Applied()


def _grad_buf(p: torch.nn.Parameter):
    """(tensor to write the gradient into, accumulate?)

    Parameters whose gradient buffer belongs to an optimiser with a LAZY zero_grad (optim.FlatAdamW: `p._msseg_gowner`): the first
    kernel that produces the parameter's gradient after a zero_grad() OVERWRITES the buffer (accumulate False), later ones in the
    same epoch add.  The zero fill of the flat gradient buffer (310 MB per step for Swin-UNETR-48) and the read of the zeros by
    every first accumulation disappear; parameters no kernel touched in an epoch are zeroed by the optimiser before it reads
    the buffer (FlatAdamW._zero_untouched)."""
    if p.grad is None:
        p.grad = torch.empty_like(p, memory_format=torch.contiguous_format)
        return p.grad, False
    owner = getattr(p, "_msseg_gowner", None)
    if owner is not None and not getattr(p, "_msseg_kgrad", False):
        p._msseg_kgrad = True      # from now on this gradient is kernel-written: the optimiser stops zero-filling its slice
        p._msseg_gepoch = owner._gepoch
        return p.grad, True        # (this epoch it still was)
    if owner is not None and getattr(p, "_msseg_gepoch", -1) != owner._gepoch:
        p._msseg_gepoch = owner._gepoch
        return p.grad, False
    return p.grad, True


def _empty_like_vol(x, C, dtype=None):
    return torch.empty(x.shape[:-1] + (C,), dtype=dtype or x.dtype, device=x.device)


def _norm_grad_bufs(nrm):
    """(dgamma, dbeta, accumulate) buffers of an InstNormAct for the fused InstanceNorm-backward epilogues"""
    if nrm.gamma is not None and nrm.gamma.requires_grad:
        dg, acc = _grad_buf(nrm.gamma)
        db, _ = _grad_buf(nrm.beta)
        return dg, db, acc
    return None, None, False


class Conv3:
    """3x3x3 s1 p1 convolution (+bias).  Few-input-channel layers go through the gather kernel."""

    def __init__(self, weight: torch.nn.Parameter, bias: Optional[torch.nn.Parameter]):
        self.w, self.b = weight, bias
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        self.cache = PackedCache()

    def _gather(self, dtype):
        return self.cin % (8 if dtype == torch.bfloat16 else 4) != 0

    def fwd(self, x, out=None, want_stats=False):
        """returns y, or (y, stats) when want_stats: InstanceNorm statistics of y, fused into the conv epilogue
        where the kernel supports it, otherwise one extra pass."""
        dtype = x.dtype
        if want_stats and out is None and self.halves_ok(tuple(x.shape[:4]), dtype):
            return self.fwd_halves(x)
        y = out if out is not None else _empty_like_vol(x, self.cout)
        stats = None
        if self._gather(dtype):
            wp = self.cache.get(self.w, dtype, "g", lambda: hip.pack_conv_gather(self.w.detach(), dtype))
            if (self.cin == 1 and dtype == torch.bfloat16 and (self.cout % 32 == 0 or self.cout % 48 == 0) and self.cout <= 256
                    and not os.environ.get("MSSEG_NO_STEM")):
                # the one-channel stem: dedicated kernel with the statistics fused
                if want_stats and x.shape[0] <= 8:
                    stats = torch.empty(x.shape[0], self.cout, 2, dtype=torch.float32, device=x.device)
                hip.conv3d_stem(x, wp, self.b, y, self.cout, stats)
            else:
                hip.conv3d_gather(x, wp, self.b, y, self.cin, self.cout, 3, 1, 1)
        else:
            vol = tuple(x.shape[:4])
            wp = self.cache.get(self.w, dtype, ("f", vol), lambda: hip.pack_conv_k3(self.w.detach(), dtype, vol=vol))
            if want_stats and x.shape[0] <= 8:
                stats = torch.empty(x.shape[0], self.cout, 2, dtype=torch.float32, device=x.device)
            hip.conv3d_k3(x, wp, self.b, y, self.cin, self.cout, stats)
        if want_stats:
            return y, (stats if stats is not None else hip.channel_stats(y))
        return y

    def halves_ok(self, vol, dtype) -> int:
        """channels per half if conv(x) on this grid can run as two launches of a ping-pong kernel on the channel halves of x
        (a concat buffer): 96 input channels -> 2 x 48 (Swin-UNETR's decoder convs over cat(up, skip), csrc/conv3d_k3_c48.hip);
        0 otherwise"""
        if dtype != torch.bfloat16 or os.environ.get("MSSEG_NO_SPLIT_CAT"):
            return 0
        if self.cin == 96 and hip.lib().msseg_conv3d_k3_kernel(*vol, 48, self.cout, hip.BF16) == 4:
            return 48
        # inference only (sliding window, 8 windows per launch): the 64- / 128-input-channel convs of the 24^3 level as 2 / 4
        # launches of the 32-channel ping-pong kernel on channel slices of the input (the generic kernel runs them at 250 TFLOP/s:
        # chains of 32-channel stages on 4x4x8 tiles); in training that grid (batch 2) is too small for the ping-pong kernel
        if (INFERENCE_FORWARD and self.cin in (64, 128) and self.cout % 32 == 0 and not os.environ.get("MSSEG_NO_KSPLIT_INFER")
                and hip.lib().msseg_conv3d_k3_kernel(*vol, 32, self.cout, hip.BF16) == 3):
            return 32
        # 64 -> 2 x 32 on the 32-channel ping-pong kernel (BasicUNet's UpCat convs in TRAINING, where the concat buffer exists
        # anyway for the weight gradient) measured neutral: 4.031 vs 4.026 ms per step -- the second launch's read-back of
        # the stored sums costs what the faster kernel gains; the inference forward (fwd_split: no concat buffer) keeps it
        return 0

    def fwd_halves(self, x):
        """y = conv(x) + bias as two launches on the two channel halves of x (the concat buffer): the second adds its sums onto
        the stored result of the first and emits the InstanceNorm statistics.  The generic kernel runs such layers one
        32-channel stage after the other (and pads 96 -> 48 to 3 x 32 input and 2 x 32 output channels); the intermediate
        sum is rounded to bf16 once more than there.  Returns (y, stats)."""
        dtype, vol = x.dtype, tuple(x.shape[:4])
        h = self.halves_ok(vol, dtype)
        y = _empty_like_vol(x, self.cout)
        stats = torch.empty(x.shape[0], self.cout, 2, dtype=torch.float32, device=x.device)
        for i in range(self.cin // h):       # two parts, or 2 / 4 slices of 32 channels (inference, 24^3 level)
            wi = self.w.detach()[:, i * h:(i + 1) * h]
            pi = self.cache.get(wi, dtype, ("fp", i, h, vol), lambda wi=wi: hip.pack_conv_k3(wi, dtype, vol=vol))
            xi = x[..., i * h:(i + 1) * h]
            if i == 0:
                hip.conv3d_k3(xi, pi, self.b, y, h, self.cout)
            else:
                hip.conv3d_k3_accumulate(xi, pi, y, h, self.cout, stats)
        return y, stats

    def split_ok(self, vol, dtype, ca, cb_) -> bool:
        """can conv(cat([a, b])) with a: ca, b: cb_ channels on the grid vol = (N, D, H, W) run as two launches of the
        ping-pong kernel (inference: no concat buffer)?"""
        if (os.environ.get("MSSEG_NO_SPLIT_CAT") or dtype != torch.bfloat16 or ca + cb_ != self.cin or ca != 32 or cb_ != 32
                or self.cout % 32 or vol[0] > 8):
            return False
        return hip.lib().msseg_conv3d_k3_kernel(*vol, 32, self.cout, hip.BF16) == 3

    def fwd_split(self, xa, xb):
        """y = conv(cat([xa, xb], channel)) + bias WITHOUT the concat buffer (inference forward of the decoder's first conv):
        launch 1 = the first 32 input channels (+ bias), launch 2 = the other 32 accumulated onto the stored result, with
        the InstanceNorm statistics of the sums.  The generic kernel takes 64-input-channel layers at 0.8 PFLOP/s, the
        ping-pong kernel 32-channel ones at 1.2: 2 x 305 us against 950 us at 96^3 x 8 windows.  The intermediate sum is
        rounded to bf16 once more than in the one-launch form.  Returns (y, stats)."""
        dtype = xa.dtype
        vol = tuple(xa.shape[:4])
        wa, wb = self.w.detach()[:, :32], self.w.detach()[:, 32:]
        pa = self.cache.get(wa, dtype, ("fa", vol), lambda: hip.pack_conv_k3(wa, dtype, vol=vol))
        pb = self.cache.get(wb, dtype, ("fb", vol), lambda: hip.pack_conv_k3(wb, dtype, vol=vol))
        y = _empty_like_vol(xa, self.cout)
        hip.conv3d_k3(xa, pa, self.b, y, 32, self.cout)
        stats = torch.empty(xa.shape[0], self.cout, 2, dtype=torch.float32, device=xa.device)
        hip.conv3d_k3_accumulate(xb, pb, y, 32, self.cout, stats)
        return y, stats

    def dgrad_accumulate_ok(self, dy) -> bool:
        """can the input gradient be ADDED onto a tensor that already holds another gradient of the input (the accumulate
        epilogue of the ping-pong kernels: 32 or 48 output channels of the forward conv, large grids)?"""
        if dy.dtype != torch.bfloat16 or self.cout not in (32, 48) or dy.shape[0] > 4 or os.environ.get("MSSEG_NO_DGRAD_ACCUM"):
            return False
        vol = tuple(dy.shape[:4])
        return hip.lib().msseg_conv3d_k3_kernel(*vol, self.cout, self.cin, hip.BF16) == (4 if self.cout == 48 else 3)

    def bwd(self, x, dy, need_dx=True, dx_out=None, bias_grad_is_zero=False, next_norm=None, accumulate_dx=False):
        """next_norm = (InstNormAct, yraw, stats, act) of the layer whose activation is this conv's input: its
        InstanceNorm-backward reductions are then fused into the input-gradient kernel; returns (dx, red).
        accumulate_dx (with dx_out, dgrad_accumulate_ok()): dx_out += the input gradient (one rounding of the sum)."""
        dtype = x.dtype
        if self.w.requires_grad:
            g, acc = _grad_buf(self.w)
            if self._gather(dtype):
                WGRAD_SIDE.run(lambda: hip.conv3d_gather_wgrad(x, dy, g, self.cin, self.cout, 3, 1, 1, acc), x, dy)
            else:
                WGRAD_SIDE.run(lambda: hip.conv3d_k3_wgrad(x, dy, g, self.cin, self.cout, acc), x, dy)
        if self.b is not None and self.b.requires_grad:
            g, acc = _grad_buf(self.b)
            if bias_grad_is_zero:
                # a bias feeding InstanceNorm: d loss / d bias == 0 identically (the mean subtraction removes it);
                # dy sums to zero over every (n, c) by construction of the InstanceNorm backward
                # (with a lazily zeroed buffer the fill runs once: nothing else ever writes this gradient)
                if not acc and not getattr(self.b, "_msseg_grad_is_zero", False):
                    g.zero_()
                    self.b._msseg_grad_is_zero = getattr(self.b, "_msseg_gowner", None) is not None
            else:
                hip.channel_sum(dy, g, acc)
        if not need_dx:
            return None
        if self._gather(dtype):
            raise NotImplementedError("input gradient of a few-channel stem conv is never needed on this path")
        vol = tuple(dy.shape[:4])
        if accumulate_dx:
            assert dx_out is not None and next_norm is None and self.dgrad_accumulate_ok(dy)
            wp = self.cache.get(self.w, dtype, ("d", vol),
                                lambda: hip.pack_conv_k3(self.w.detach(), dtype, dgrad=True, vol=vol))
            unused = torch.empty(dy.shape[0], self.cin, 2, dtype=torch.float32, device=dy.device)   # the epilogue's statistics
            hip.conv3d_k3_accumulate(dy, wp, dx_out, self.cout, self.cin, unused)
            return dx_out
        if hip.conv3d_k3_small_ok(dy, self.cout, self.cin):
            # small grid: split-K partials, then ONE finish kernel = the plain sum, or the receiving unit's whole backward
            wp = self.cache.get(self.w, dtype, "ds", lambda: hip.pack_conv_k3(self.w.detach(), dtype, dgrad=True, cb=32))
            part, ng = hip.conv3d_k3_small_partials(dy, wp, self.cout, self.cin)
            dx = dx_out if dx_out is not None else _empty_like_vol(dy, self.cin)
            if next_norm is not None:
                nrm, yraw, stats, act = next_norm
                dg, db, acc = _norm_grad_bufs(nrm)
                hip.conv3d_k3_small_bwd_finish(part, ng, dx, (yraw, stats, nrm.gamma, nrm.beta, nrm.eps, nrm.slope),
                                               dg, db, acc)
                return dx, APPLIED
            return hip.conv3d_k3_small_bwd_finish(part, ng, dx)
        wp = self.cache.get(self.w, dtype, ("d", vol),
                            lambda: hip.pack_conv_k3(self.w.detach(), dtype, dgrad=True, vol=vol))
        dx = dx_out if dx_out is not None else _empty_like_vol(dy, self.cin)
        if next_norm is not None:
            nrm, yraw, stats, act = next_norm
            if dy.shape[0] <= 8 and self.cin % 4 == 0:
                dg = db = None
                acc = False
                if nrm.gamma is not None and nrm.gamma.requires_grad:
                    dg, acc = _grad_buf(nrm.gamma)
                    db, _ = _grad_buf(nrm.beta)
                red = hip.conv3d_k3_dgrad_inbwd(dy, wp, dx, self.cout, self.cin, yraw, act, stats, nrm.slope, nrm.eps,
                                                dg, db, acc)
                return dx, red
            hip.conv3d_k3(dy, wp, None, dx, self.cout, self.cin)
            return dx, None
        hip.conv3d_k3(dy, wp, None, dx, self.cout, self.cin)
        return dx


class Conv1:
    """1x1x1 convolution (+bias).  `cin_pad`: the input tensor carries zero-padded channels up to cin_pad."""

    def __init__(self, weight, bias):
        self.w, self.b = weight, bias
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        self.cache = PackedCache()

    def _gather(self, dtype):
        return self.cin % (8 if dtype == torch.bfloat16 else 4) != 0

    def fwd(self, x, out=None, want_stats=False):
        dtype = x.dtype
        y = out if out is not None else _empty_like_vol(x, self.cout)
        if self._gather(dtype):
            wp = self.cache.get(self.w, dtype, "g", lambda: hip.pack_conv_gather(self.w.detach(), dtype))
            if (want_stats and self.cin == 1 and dtype == torch.bfloat16 and x.dim() == 5 and x.shape[0] <= 8 and hip.ld(y) == self.cout
                    and (self.cout % 32 == 0 or self.cout % 48 == 0) and self.cout <= 256 and not os.environ.get("MSSEG_NO_STEM")):
                # the shortcut conv on the one-channel input: stem kernel (centre tap) with the statistics fused
                stats = torch.empty(x.shape[0], self.cout, 2, dtype=torch.float32, device=x.device)
                hip.conv3d_stem(x, wp, self.b, y, self.cout, stats, k=1)
                return y, stats
            hip.conv3d_gather(x, wp, self.b, y, self.cin, self.cout, 1, 1, 0)
        elif (self.cout <= 4 and self.cin <= 64 and x.shape[-1] == self.cin and self.w.is_contiguous()
              and not os.environ.get("MSSEG_NO_HEAD_KERNEL")):
            # segmentation head (2-4 classes): streaming kernel on the fp32 weight as it is
            hip.conv3d_k1_head(x, self.w.detach(), self.b, y, self.cin, self.cout)
        else:
            wp = self.cache.get(self.w, dtype, "f",
                                lambda: hip.pack_conv_k1(self.w.detach().reshape(self.cout, self.cin), dtype))
            hip.conv3d_k1(x, wp, self.b, y, self.cin, self.cout)
        if want_stats:
            return y, hip.channel_stats(y)
        return y

    # ---- segmentation head fused with the InstanceNorm + LeakyReLU of the unit in front of it ----
    def head_norm_ok(self, yraw) -> bool:
        """the head can read the RAW conv output of the last conv+norm unit and normalise on load (forward) / recompute the
        activation (backward): the unit's activation tensor is then never written or read"""
        epc = 16 // yraw.element_size()
        return (self.cout <= 4 and self.cin <= 64 and self.cin % epc == 0 and yraw.shape[-1] == self.cin
                and hip.ld(yraw) % epc == 0 and yraw.data_ptr() % 16 == 0 and self.w.is_contiguous()
                and not os.environ.get("MSSEG_NO_HEAD_KERNEL") and not os.environ.get("MSSEG_NO_HEAD_FUSE"))

    def fwd_norm(self, yraw, stats, nrm, out):
        return hip.conv3d_k1_head_norm(yraw, stats, nrm.gamma, nrm.beta, nrm.slope, nrm.eps, self.w.detach(), self.b, out,
                                       self.cin, self.cout)

    def bwd_norm(self, dy, dy_channels, nrm, yraw, stats):
        """backward of fwd_norm: returns (da, red) for the unit's `bwd(..., red=red)`; weight / bias gradients land in .grad"""
        dyk = dy[..., :dy_channels]
        dx = torch.empty(yraw.shape, dtype=yraw.dtype, device=yraw.device)
        dg, db, acc = _norm_grad_bufs(nrm)
        if self.b is not None and self.b.requires_grad:
            g, bacc = _grad_buf(self.b)
            hip.channel_sum(dy[..., :self.cout], g, bacc)
        if self.w.requires_grad:
            gw, wacc = _grad_buf(self.w)
            red = hip.conv3d_k1_head_bwd_fused(dyk, self.w.detach(), dx, self.cin, self.cout, yraw, stats, nrm.gamma, nrm.beta,
                                               nrm.slope, nrm.eps, gw, wacc, dg, db, acc)
        else:
            red = hip.conv3d_k1_head_dgrad_inbwd(dyk, self.w.detach(), dx, self.cin, self.cout, yraw, stats, nrm.gamma,
                                                 nrm.beta, nrm.slope, nrm.eps, dg, db, acc)
        return dx, red

    def bwd(self, x, dy, need_dx=True, dy_channels=None, next_norm=None):
        """dy may carry zero-padded channels (dy_channels = padded count, multiple of 8).
        next_norm = (InstNormAct, yraw, stats, act) of the layer whose activation is this conv's input: its
        InstanceNorm-backward sums are then fused into the input-gradient kernel; returns (dx, red)."""
        dtype = x.dtype
        fused_bias = False
        if self.w.requires_grad:
            g, acc = _grad_buf(self.w)
            if self._gather(dtype):
                WGRAD_SIDE.run(lambda: hip.conv3d_gather_wgrad(x, dy[..., :self.cout], g, self.cin, self.cout, 1, 1, 0, acc), x, dy)
            elif hip.linear_wgrad_ok(x, self.cin, self.cout):
                # weight and bias gradient from one pass over the voxels (csrc/linear_wgrad.hip)
                gb, bacc = _grad_buf(self.b) if (self.b is not None and self.b.requires_grad) else (None, False)
                WGRAD_SIDE.run(lambda: hip.linear_wgrad(x, dy[..., :self.cout], g.view(self.cout, self.cin), gb, self.cin,
                                                        self.cout, acc, bacc), x, dy)
                fused_bias = True
            else:
                WGRAD_SIDE.run(lambda: hip.conv3d_k1_wgrad(x, dy[..., :self.cout], g, self.cin, self.cout, acc), x, dy)
        if self.b is not None and self.b.requires_grad and not fused_bias:
            g, acc = _grad_buf(self.b)
            hip.channel_sum(dy[..., :self.cout], g, acc)
        if not need_dx:
            return None
        if self._gather(dtype):
            raise NotImplementedError("input gradient of a few-channel 1x1 conv is never needed on this path")
        dx = _empty_like_vol(dy, self.cin)
        kpad = dy_channels if dy_channels is not None else self.cout
        dyk = dy if dy_channels is None else dy[..., :kpad]
        epc = 16 // dy.element_size()
        if (next_norm is not None and self.cout <= 4 and kpad >= 4 and self.cin <= 64 and self.cin % epc == 0 and hip.ld(dyk) % epc == 0
                and dyk.data_ptr() % 16 == 0 and self.w.is_contiguous() and not os.environ.get("MSSEG_NO_HEAD_KERNEL")):
            # segmentation head: streaming kernel on the fp32 weight, with the receiving layer's backward sums
            nrm, yraw, stats, act = next_norm
            dg, db, acc = _norm_grad_bufs(nrm)
            red = hip.conv3d_k1_head_dgrad_inbwd(dyk, self.w.detach(), dx, self.cin, self.cout, yraw, stats, nrm.gamma,
                                                 nrm.beta, nrm.slope, nrm.eps, dg, db, acc)
            return dx, red
        wp = self.cache.get(self.w, dtype, "d",
                            lambda: hip.pack_conv_k1(self.w.detach().reshape(self.cout, self.cin), dtype, dgrad=True))
        if next_norm is not None:
            nrm, yraw, stats, act = next_norm
            if dy.shape[0] <= 8 and self.cin % 4 == 0:
                dg, db, acc = _norm_grad_bufs(nrm)
                red = hip.conv3d_k1_dgrad_inbwd(dyk, wp, dx, kpad, self.cin, yraw, act, stats, nrm.slope, nrm.eps, dg, db, acc)
                return dx, red
            hip.conv3d_k1(dyk, wp, None, dx, kpad, self.cin)
            return dx, None
        hip.conv3d_k1(dyk, wp, None, dx, kpad, self.cin)
        return dx


class Deconv2:
    """ConvTranspose3d k = s = 2 (+bias): [N,D,H,W,Cin] -> [N,2D,2H,2W,Cout]."""

    def __init__(self, weight, bias):
        self.w, self.b = weight, bias
        self.cin, self.cout = weight.shape[0], weight.shape[1]
        self.cache = PackedCache()

    def fwd(self, x, out=None):
        dtype = x.dtype
        N, D, H, W, _ = x.shape
        y = out if out is not None else torch.empty(N, 2 * D, 2 * H, 2 * W, self.cout, dtype=dtype, device=x.device)
        wp = self.cache.get(self.w, dtype, "f", lambda: hip.pack_deconv(self.w.detach(), dtype))
        hip.deconv_k2s2(x, wp, self.b, y, self.cin, self.cout)
        return y

    def bwd(self, x, dy, need_dx=True, next_norm=None):
        """next_norm = (InstNormAct, yraw, stats, act) of the layer whose activation `x` is: its InstanceNorm-backward sums
        are fused into the input-gradient kernel; returns (dx, red).  The bias gradient comes out of the same pass."""
        dtype = x.dtype
        if self.w.requires_grad:
            g, acc = _grad_buf(self.w)
            WGRAD_SIDE.run(lambda: hip.deconv_k2s2_wgrad(x, dy, g, self.cin, self.cout, acc), x, dy)
        want_db = self.b is not None and self.b.requires_grad
        if not need_dx:
            if want_db:
                g, acc = _grad_buf(self.b)
                hip.channel_sum(dy, g, acc)
            return None
        wp = self.cache.get(self.w, dtype, "d", lambda: hip.pack_deconv(self.w.detach(), dtype, bwd=True))
        dx = torch.empty_like(x, memory_format=torch.contiguous_format)
        db, dbacc = _grad_buf(self.b) if want_db else (None, False)
        if next_norm is not None and hip.deconv_k2s2_small_unit_ok(tuple(x.shape), self.cin, self.cout, dtype):
            # small grid (the deep UpCat levels): K-split input gradient as one fp32 block, then the small-grid finish kernel
            # runs the receiving conv + InstanceNorm + LeakyReLU unit's whole backward -- 2 launches for the generic flat
            # kernel with fused sums, its finalize and the apply pass (csrc/deconv_k2s2_gen.hip, conv3d_k3_small.hip)
            nrm, yraw, stats, act = next_norm
            part = hip.deconv_k2s2_bwd_partials(dy, wp, self.cin, self.cout)
            dg, dbt, nacc = _norm_grad_bufs(nrm)
            hip.conv3d_k3_small_bwd_finish(part, 1, dx, (yraw, stats, nrm.gamma, nrm.beta, nrm.eps, nrm.slope), dg, dbt, nacc)
            if want_db:
                hip.channel_sum(dy, db, dbacc)
            return dx, APPLIED
        nn_ = None
        dg = dbt = None
        nacc = False
        if next_norm is not None:
            nrm, yraw, stats, act = next_norm
            if dy.shape[0] <= 8 and self.cin % 4 == 0:
                dg, dbt, nacc = _norm_grad_bufs(nrm)
                nn_ = (yraw, act, stats, nrm.slope, nrm.eps)
        red = hip.deconv_k2s2_bwd_fused(dy, wp, dx, self.cin, self.cout, nn_, db, dbacc, dg, dbt, nacc)
        return (dx, red) if next_norm is not None else dx


class InstNormAct:
    """InstanceNorm3d(eps 1e-5) [+affine] [+residual] + LeakyReLU(slope); slope == 1 -> no activation."""

    def __init__(self, gamma: Optional[torch.nn.Parameter], beta: Optional[torch.nn.Parameter], slope: float,
                 eps: float = 1e-5):
        self.gamma, self.beta, self.slope, self.eps = gamma, beta, float(slope), float(eps)

    def fwd(self, y_raw, out=None, residual=None, stats=None, pooled=None):
        """pooled: optional [N, D/2, H/2, W/2, C] tensor that receives max_pool3d(result, 2) from the same pass"""
        if stats is None:
            stats = hip.channel_stats(y_raw)
        a = out if out is not None else torch.empty_like(y_raw, memory_format=torch.contiguous_format)
        if pooled is not None:
            if residual is None and hip.instnorm_pool_ok(y_raw, a, pooled) and not os.environ.get("MSSEG_NO_NORM_POOL"):
                hip.instnorm_act_pool_fwd(y_raw, stats, self.gamma, self.beta, a, pooled, self.slope, self.eps)
                return a, stats
            hip.instnorm_act_fwd(y_raw, stats, self.gamma, self.beta, a, self.slope, self.eps, residual)
            hip.maxpool2_fwd(a, pooled)
            return a, stats
        hip.instnorm_act_fwd(y_raw, stats, self.gamma, self.beta, a, self.slope, self.eps, residual)
        return a, stats

    def pool_bwd_reduce(self, y_raw, stats, skip_grad, pooled_grad):
        """Encoder level: the gradient of this layer's output is skip_grad + maxpool-backward(pooled_grad).  One kernel
        forms it (dense) together with this layer's backward sums; returns (da, red) for `bwd(..., red=red)`, or None when
        the shape is not eligible (the caller then runs maxpool2_bwd and the plain backward)."""
        if (os.environ.get("MSSEG_NO_NORM_POOL") or os.environ.get("MSSEG_NO_POOL_BWD_FUSE")
                or not hip.instnorm_pool_ok(y_raw, skip_grad, pooled_grad)):
            return None
        da = torch.empty(y_raw.shape, dtype=y_raw.dtype, device=y_raw.device)
        dg, db, acc = _norm_grad_bufs(self)
        red = hip.instnorm_act_poolbwd_reduce(y_raw, stats, self.gamma, self.beta, skip_grad, pooled_grad, da, self.slope,
                                              self.eps, dg, db, acc)
        return da, red

    def bwd(self, y_raw, stats, a, da, want_dres=False, red=None):
        """red: reductions already produced by the kernel that made `da` (fused path) -> only the apply pass runs;
        red is APPLIED: `da` already IS this unit's dy (the producer's finish kernel ran the whole backward)"""
        if red is APPLIED:
            if want_dres:
                raise RuntimeError("a residual unit cannot take an already-applied gradient")
            return da
        dy = torch.empty(y_raw.shape, dtype=y_raw.dtype, device=y_raw.device)
        dres = torch.empty(y_raw.shape, dtype=y_raw.dtype, device=y_raw.device) if want_dres else None
        # without a residual the sign of the pre-activation is recomputed from y_raw: one tensor read less
        a_in = a if want_dres else None
        if red is not None:
            hip.instnorm_act_bwd_apply(y_raw, stats, self.gamma, a_in, da, red, dy, self.slope, self.eps, dres, self.beta)
            return (dy, dres) if want_dres else dy
        dg = db = None
        acc = False
        if self.gamma is not None and self.gamma.requires_grad:
            dg, acc = _grad_buf(self.gamma)
            db, acc2 = _grad_buf(self.beta)
            assert acc == acc2
        hip.instnorm_act_bwd(y_raw, stats, self.gamma, a_in, da, dy, self.slope, self.eps, dres, dg, db, acc, self.beta)
        return (dy, dres) if want_dres else dy


class ConvNormAct:
    """conv3x3x3 -> InstanceNorm -> LeakyReLU, the unit of BasicUNet's TwoConv."""

    def __init__(self, conv: Conv3, norm: InstNormAct):
        self.conv, self.norm = conv, norm

    def fwd(self, x, out=None, pooled=None):
        cv, nm = self.conv, self.norm
        if (INFERENCE_FORWARD and pooled is None and cv.cin == 1 and x.dtype == torch.bfloat16 and x.shape[0] <= 8
                and (cv.cout % 32 == 0 or cv.cout % 48 == 0) and cv.cout <= 256 and not os.environ.get("MSSEG_NO_STEM")
                and not os.environ.get("MSSEG_NO_STEM_TWICE")):
            # inference, the one-channel stem unit: a statistics-only launch, then the conv again with InstanceNorm + LeakyReLU
            # in its epilogue -- the raw output is never written and the normalisation pass over it (read + write of the
            # largest tensor of the network) disappears; the conv itself reads one channel and is bound by its output write
            wp = cv.cache.get(cv.w, x.dtype, "g", lambda: hip.pack_conv_gather(cv.w.detach(), x.dtype))
            stats = torch.empty(x.shape[0], cv.cout, 2, dtype=torch.float32, device=x.device)
            hip.conv3d_stem(x, wp, cv.b, None, cv.cout, stats)
            a = out if out is not None else _empty_like_vol(x, cv.cout)
            hip.conv3d_stem_norm(x, wp, cv.b, stats, nm.gamma, nm.beta, nm.eps, nm.slope, a, cv.cout)
            return a, (x, None, stats, a)
        if (not cv._gather(x.dtype) and hip.conv3d_k3_small_ok(x, cv.cin, cv.cout)
                and (pooled is None or hip.instnorm_pool_ok(x, x, pooled))):
            # small grid (12^3 / 6^3 levels): split-K partials + one finish kernel for bias, raw output, statistics,
            # normalise + LeakyReLU and the max-pool -- 2 launches instead of conv, finalize, normalise (+ pool)
            wp = cv.cache.get(cv.w, x.dtype, "fs", lambda: hip.pack_conv_k3(cv.w.detach(), x.dtype, cb=32))
            part, ng = hip.conv3d_k3_small_partials(x, wp, cv.cin, cv.cout)
            y = _empty_like_vol(x, cv.cout)
            a = out if out is not None else _empty_like_vol(x, cv.cout)
            stats = torch.empty(x.shape[0], cv.cout, 2, dtype=torch.float32, device=x.device)
            hip.conv3d_k3_small_fwd_finish(part, ng, cv.b, nm.gamma, nm.beta, nm.eps, nm.slope, y, a, pooled, stats)
            return a, (x, y, stats, a)
        y, stats = self.conv.fwd(x, want_stats=True)
        a, stats = self.norm.fwd(y, out, stats=stats, pooled=pooled)
        return a, (x, y, stats, a)

    def bwd(self, saved, da, need_dx=True, red=None, next_saved=None, next_cna=None):
        """red: this layer's IN-backward reductions if the producer of `da` already computed them.
        next_cna/next_saved: the ConvNormAct (and its saved tuple) whose activation is this conv's input; when given
        the result is (dx, red_for_that_layer)."""
        x, y, stats, a = saved
        dy = self.norm.bwd(y, stats, a, da, red=red)
        if next_cna is not None and need_dx:
            nx, ny, nstats, na = next_saved
            return self.conv.bwd(x, dy, True, bias_grad_is_zero=True, next_norm=(next_cna.norm, ny, nstats, na))
        return self.conv.bwd(x, dy, need_dx, bias_grad_is_zero=True)


def maxpool_fwd(x):
    N, D, H, W, C = x.shape
    y = torch.empty(N, D // 2, H // 2, W // 2, C, dtype=x.dtype, device=x.device)
    return hip.maxpool2_fwd(x, y)


class ResBlock:
    """MONAI UnetResBlock (stride 1): lrelu(IN(conv2(lrelu(IN(conv1(x))))) + r), r = IN(conv3_1x1(x)) iff in != out
    else x; convs bias-free, InstanceNorm affine=False, LeakyReLU(0.01)  (SURVEY.md row A3)."""

    def __init__(self, conv1_w, conv2_w, conv3_w=None, slope=0.01):
        self.c1, self.c2 = Conv3(conv1_w, None), Conv3(conv2_w, None)
        self.c3 = Conv1(conv3_w, None) if conv3_w is not None else None
        self.n1, self.n2 = InstNormAct(None, None, slope), InstNormAct(None, None, slope)
        self.n3 = InstNormAct(None, None, 1.0) if conv3_w is not None else None

    @staticmethod
    def _small_unit(conv, nrm, x, residual=None, out=None):
        """conv + InstanceNorm (+ residual) + LeakyReLU on a small grid: split-K partials + ONE finish kernel (raw output,
        statistics, normalise, residual, activation) -- csrc/conv3d_k3_small.hip; returns (activation, raw output, stats)"""
        wp = conv.cache.get(conv.w, x.dtype, "fs", lambda: hip.pack_conv_k3(conv.w.detach(), x.dtype, cb=32))
        part, ng = hip.conv3d_k3_small_partials(x, wp, conv.cin, conv.cout)
        y = _empty_like_vol(x, conv.cout)
        a = out if out is not None else _empty_like_vol(x, conv.cout)
        stats = torch.empty(x.shape[0], conv.cout, 2, dtype=torch.float32, device=x.device)
        hip.conv3d_k3_small_fwd_finish(part, ng, conv.b, nrm.gamma, nrm.beta, nrm.eps, nrm.slope, y, a, None, stats, residual)
        return a, y, stats

    def fwd(self, x, out=None):
        # the 12^3 ... 3^3 levels (Swin-UNETR encoder4 / encoder10 / decoder5 / decoder4): both 3x3x3 convs on the split-K path
        small = (not self.c1._gather(x.dtype) and hip.conv3d_k3_small_ok(x, self.c1.cin, self.c1.cout)
                 and hip.conv3d_k3_small_ok(x, self.c2.cin, self.c2.cout))
        if small:
            a1, y1, s1 = self._small_unit(self.c1, self.n1, x)
        else:
            y1, s1 = self.c1.fwd(x, want_stats=True)
            a1, s1 = self.n1.fwd(y1, stats=s1)
        if self.c3 is not None:
            y3, s3 = self.c3.fwd(x, want_stats=True)
            r, s3 = self.n3.fwd(y3, stats=s3)
        else:
            y3 = s3 = None
            r = x
        if small:
            o, y2, s2 = self._small_unit(self.c2, self.n2, a1, residual=r, out=out)
        else:
            y2, s2 = self.c2.fwd(a1, want_stats=True)
            o, s2 = self.n2.fwd(y2, out, residual=r, stats=s2)
        return o, (x, y1, s1, a1, y2, s2, y3, s3, r, o)

    def bwd(self, saved, do, need_dx=True):
        x, y1, s1, a1, y2, s2, y3, s3, r, o = saved
        dy2, dres = self.n2.bwd(y2, s2, o, do, want_dres=True)
        da1, red1 = self.c2.bwd(a1, dy2, True, next_norm=(self.n1, y1, s1, a1))
        dy1 = self.n1.bwd(y1, s1, a1, da1, red=red1)
        if need_dx and self.c1.dgrad_accumulate_ok(dy1):
            # large grids (Swin-UNETR's 96^3 / 48^3 UnetResBlocks): the gradient of the shortcut first, then the first conv's
            # input gradient is added onto it by that kernel's epilogue -- no separate add pass over the (up to 96-channel)
            # input gradient (3 x 340 MB at 96^3, batch 2)
            if self.c3 is not None:
                dy3 = self.n3.bwd(y3, s3, r, dres)
                dx3 = self.c3.bwd(x, dy3, True)
            else:
                dx3 = dres
            return self.c1.bwd(x, dy1, True, dx_out=dx3, accumulate_dx=True)
        dx = self.c1.bwd(x, dy1, need_dx)
        if self.c3 is not None:
            dy3 = self.n3.bwd(y3, s3, r, dres)
            dx3 = self.c3.bwd(x, dy3, need_dx)
        else:
            dx3 = dres
        if not need_dx:
            return None
        return hip.add(dx, dx3, dx)


class UpBlock:
    """MONAI UnetrUpBlock: up = ConvT(k = s)(x) (bias-free); ResBlock(cat([up, skip]))  (row A4).  The transposed conv
    and the skip producer write straight into the two halves of the concat buffer."""

    def __init__(self, transp_w, conv1_w, conv2_w, conv3_w, slope=0.01):
        self.up = Deconv2(transp_w, None)
        self.res = ResBlock(conv1_w, conv2_w, conv3_w, slope)
        self.cout = transp_w.shape[1]

    def alloc_cat(self, x):
        N, D, H, W, _ = x.shape
        return torch.empty(N, 2 * D, 2 * H, 2 * W, 2 * self.cout, dtype=x.dtype, device=x.device)

    def fwd(self, x, cat):
        """cat[..., cout:] must already hold the skip; returns (out, saved)."""
        self.up.fwd(x, cat[..., :self.cout])
        o, s = self.res.fwd(cat)
        return o, (x, s)

    def bwd(self, saved, do):
        x, s = saved
        dcat = self.res.bwd(s, do, True)
        dx = self.up.bwd(x, dcat[..., :self.cout], True)
        return dx, dcat[..., self.cout:]


def _merge_batch(t):
    """[N, D, H, W, C] (dense or a channel slice of a wider buffer) as ONE sample [1, N*D, H, W, C]: per-channel
    statistics over a channels-last batch are InstanceNorm statistics of that merged volume"""
    N, D, H, W, C = t.shape
    return t.view(1, N * D, H, W, C)


class BatchNormAct:
    """BatchNorm3d + ReLU / LeakyReLU(slope) (slope 1: no activation) of the reference's UNETR conv decoder
    (/root/reference/models/segmentors/unetr.py:28-52) and of the SwinDepth MLP (models/backbones/swindepth.py:42-44).
    Training mode normalises with the batch statistics and updates the running ones (momentum, unbiased variance,
    num_batches_tracked) exactly like torch.nn.BatchNorm3d; eval mode uses the running statistics.  The kernels are the
    InstanceNorm ones on the merged volume (channels-last: a batch is one long sample).

    `group` (a torch.distributed process group, or True for the default one) makes it SyncBatchNorm -- the reference
    converts every BatchNorm under DDP (/root/reference/run_training.py:83): the per-channel sums of the forward and
    the two sums of the backward are all-reduced ([C, 2] floats each), everything else stays local."""

    def __init__(self, bn: torch.nn.BatchNorm3d, slope: float = 0.0, group=None):
        self.bn = bn
        self.inner = InstNormAct(bn.weight, bn.bias, slope, bn.eps)
        self.group = group

    def _all_reduce(self, t):
        import torch.distributed as dist
        dist.all_reduce(t, group=None if self.group is True else self.group)

    def _synced(self):
        import torch.distributed as dist
        if self.group is None or not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(None if self.group is True else self.group) > 1

    def fwd(self, y_raw, out=None, stats=None):
        """stats: optional per-sample (sum, sum of squares) [N, C, 2] from the conv epilogue.  Returns (act, saved stats);
        the saved statistics are scaled so that the kernels' local element count yields the (global) mean / variance."""
        bn = self.bn
        N, D, H, W, C = y_raw.shape
        cnt = N * D * H * W
        if bn.training or not bn.track_running_stats:
            s = stats.sum(0, keepdim=True) if stats is not None else hip.channel_stats(_merge_batch(y_raw))
            total = cnt
            if self._synced():
                pack = torch.cat([s.reshape(-1), torch.full((1,), float(cnt), device=s.device)])
                self._all_reduce(pack)
                total = float(pack[-1])          # host sync: SyncBatchNorm runs eagerly
                s = (pack[:-1] * (cnt / total)).reshape(1, C, 2)
            if bn.training and bn.track_running_stats:
                with torch.no_grad():
                    bn.num_batches_tracked.add_(1)
                    mean = s[0, :, 0] / cnt
                    var = (s[0, :, 1] / cnt - mean * mean).clamp_(min=0) * (total / max(total - 1, 1))
                    if bn.momentum is None:
                        m = 1.0 / bn.num_batches_tracked.to(torch.float32)
                        bn.running_mean.add_((mean - bn.running_mean) * m)
                        bn.running_var.add_((var - bn.running_var) * m)
                    else:
                        bn.running_mean.mul_(1 - bn.momentum).add_(mean, alpha=bn.momentum)
                        bn.running_var.mul_(1 - bn.momentum).add_(var, alpha=bn.momentum)
        else:
            # Eval mode: y = x * scale + shift with scale = gamma / sqrt(running_var + eps), shift = beta - running_mean *
            # scale, formed in double.  The kernel normalises with (mean, var) it derives from sums as E[x^2] - mean^2; handing
            # it sums rebuilt from the running statistics would cancel catastrophically when mean^2 >> var (mean 10, var
            # 1e-2: 1e-3 relative error in rstd).  Instead it gets the sums of a zero-mean, (1 - eps)-variance channel --
            # mean 0 and rstd 1 come out exactly -- and the folded scale / shift as its affine parameters.
            eps = self.inner.eps
            rstd = torch.rsqrt(bn.running_var.double() + eps)
            g = self.inner.gamma.detach().double() if self.inner.gamma is not None else torch.ones_like(rstd)
            bta = self.inner.beta.detach().double() if self.inner.beta is not None else torch.zeros_like(rstd)
            scale = g * rstd
            shift = bta - bn.running_mean.double() * scale
            s = torch.zeros(1, C, 2, dtype=torch.float32, device=y_raw.device)
            s[0, :, 1] = (1.0 - eps) * cnt
            a = out if out is not None else torch.empty(y_raw.shape, dtype=y_raw.dtype, device=y_raw.device)
            hip.instnorm_act_fwd(_merge_batch(y_raw), s, scale.float().contiguous(), shift.float().contiguous(),
                                 _merge_batch(a), self.inner.slope, eps, None)
            return a, s
        a = out if out is not None else torch.empty(y_raw.shape, dtype=y_raw.dtype, device=y_raw.device)
        hip.instnorm_act_fwd(_merge_batch(y_raw), s, self.inner.gamma, self.inner.beta, _merge_batch(a), self.inner.slope,
                             self.inner.eps, None)
        return a, s

    def bwd(self, y_raw, stats, da):
        if not (self.bn.training or not self.bn.track_running_stats):
            raise NotImplementedError("backward through an eval-mode BatchNorm (frozen statistics) is not on this path")
        if not self._synced():
            dy = self.inner.bwd(_merge_batch(y_raw), stats, None, _merge_batch(da))
            return dy.view(y_raw.shape)
        n = self.inner
        ym, dam = _merge_batch(y_raw), _merge_batch(da)
        dg, db, acc = _norm_grad_bufs(n)
        red = hip.instnorm_act_bwd_reduce(ym, stats, n.gamma, None, dam, n.slope, n.eps, dg, db, acc, n.beta)
        cnt = float(y_raw.numel() // y_raw.shape[-1])
        pack = torch.cat([red.reshape(-1), torch.full((1,), cnt, device=red.device)])
        self._all_reduce(pack)
        red = (pack[:-1] * (cnt / pack[-1])).reshape(red.shape).contiguous()
        dy = torch.empty(ym.shape, dtype=ym.dtype, device=ym.device)
        hip.instnorm_act_bwd_apply(ym, stats, n.gamma, None, dam, red, dy, n.slope, n.eps, None, n.beta)
        return dy.view(y_raw.shape)


class ConvBNAct:
    """Conv3d k3 p1 (+bias) -> BatchNorm3d -> ReLU: `Conv3DBlock` of the reference's UNETR decoder."""

    def __init__(self, conv: Conv3, norm: BatchNormAct):
        self.conv, self.norm = conv, norm

    def fwd(self, x, out=None):
        y, st = self.conv.fwd(x, want_stats=True)
        a, s = self.norm.fwd(y, out, stats=st)
        return a, (x, y, s)

    def bwd(self, saved, da, need_dx=True):
        x, y, s = saved
        dy = self.norm.bwd(y, s, da)
        # a bias in front of a training-mode BatchNorm has an identically zero gradient, like in front of InstanceNorm
        return self.conv.bwd(x, dy, need_dx, bias_grad_is_zero=True)
