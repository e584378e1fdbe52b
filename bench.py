#!/usr/bin/env python3
"""Driver benchmark: UNet adapter train step (bs=1/GPU on SDXL; bs=4 on SD1.5), per algorithm, on N MI355X.

One "step" = one pass of the adapter hot path over every adapted layer of the UNet (synthetic activations of the real
shapes: benchmarks/sdxl_shapes.py = 788 layers at 1024x1024, benchmarks/sd15_shapes.py = 278 layers at 512x512 bs 4):
adapter forward (delta) and backward (dx per layer; the factor gradients of the Linear layers in grouped launches of 18-24
layers each, accumulated straight into the flat gradient arena -- DESIGN.md 1) of every layer, the arena zero-fill, the data-parallel mean all-reduce of the adapter gradients (N > 1: RCCL, launched bucket by
bucket between the backward segments so it overlaps the rest of the backward) and a fused AdamW update.  EVERY layer
instance owns its upstream gradient g; the inputs x are laid out as in the UNet: to_q / to_k / to_v of a self-attention and to_k / to_v
of a cross-attention read ONE tensor (round 5: such sibling sets of LoKr nn.Linear layers run as one grouped launch each, through the
same op the modules' sibling sets call behind the reference API; --no-siblings = the round 1-4 layout), every other layer owns its x:
a step reads the ~6 GB a real step reads, from HBM, not from the 256 MB Infinity Cache.  The compute of a step is captured in hipGraphs
and replayed.  `python bench.py --gpus N` without a launcher around it starts its own N ranks (torch.distributed.run on 127.0.0.1).

    python bench.py --gpus N --steps K --warmup W [--algo lokr|locon|loha|ia3|mixed] [--model sdxl|sd15] [--dtype ..]

Prints ONE JSON line on rank 0.  Besides the contract fields: "roofline" (dominant kernel family, live HIP-event
timing, PMC traffic of the same build from profiles/), "cpu_baseline" (reference CPU path, bounded sample),
"reference_rocm_eager" (the reference's own call sequence through stock PyTorch-ROCm on this GPU, eager vs eager and
graph vs graph -- the north-star comparator) and "base_plus_adapter" (the step with the frozen layers' own ops in it).
"""
import argparse
import ctypes
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from benchmarks.sd15_shapes import sd15_unet_layers  # noqa: E402
from benchmarks.sdxl_shapes import layer_rows, sdxl_unet_layers  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: "8.0 TB/s spec")
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA peak (same guide: "~2.5 PF dense")
FACTOR = 8
ALGO_LABEL = {"lokr": "LoKr factor=8", "locon": "LoCon dim=16 conv_dim=8", "loha": "LoHa dim=32", "ia3": "(IA)^3",
              "mixed": "mixed preset (LoCon attn + LoKr FFN + (IA)^3 elsewhere)"}
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--algo", default="lokr", choices=list(ALGO_LABEL))
    ap.add_argument("--preset", default=None, help="alias: --preset mixed == --algo mixed")
    ap.add_argument("--model", default="sdxl", choices=["sdxl", "sd15"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--channels-last", action="store_true", default=True,
                    help="(default since round 3) Conv2d activations in torch.channels_last memory -- what "
                         "`unet.to(memory_format=torch.channels_last)` gives and what DESIGN.md 2 documents as the layout of the native "
                         "path: the NHWC row matrices are free views, no transposes")
    ap.add_argument("--nchw", dest="channels_last", action="store_false",
                    help="Conv2d activations in contiguous NCHW memory: two layout transposes per conv layer and pass on top (A/B leg)")
    ap.add_argument("--overlap-leg", action="store_true",
                    help="development: also time base + adapter with every layer's adapter on a side HIP stream (negative result, HISTORY.md 7.4)")
    ap.add_argument("--no-defer", action="store_true",
                    help="A/B: one LoKr weight-gradient launch per layer instead of the grouped launches")
    ap.add_argument("--rank", type=int, default=0,
                    help="LoKr: low-rank w2 = w2_a @ w2_b of this rank (BASELINE configs[3] 'factor=8 (low)': dim 16); 0 = full-matrix w2 "
                         "(the 'dim=10k-class' headline configuration)")
    ap.add_argument("--no-planes", action="store_true",
                    help="A/B: no pre-packed LoKr operand planes (every workgroup converts its fp32 w2 tile, as in rounds 1-2)")
    ap.add_argument("--no-reference", action="store_true", help="skip the PyTorch-ROCm eager comparator leg")
    ap.add_argument("--no-base", action="store_true", help="skip the base+adapter leg")
    ap.add_argument("--no-per-algo", action="store_true",
                    help="skip the per_algo object (short runs of the other BASELINE configs in child processes)")
    ap.add_argument("--no-own-base", action="store_true",
                    help="base + adapter leg: F.linear and the adapter as two autograd nodes (rounds 1-5) instead of ops.lokr_adapted_linear")
    ap.add_argument("--no-siblings", action="store_true",
                    help="A/B, the round 1-4 layout: every layer instance owns its input and is launched on its own (default: the "
                         "to_q / to_k / to_v layers of a self-attention and to_k / to_v of a cross-attention read ONE tensor, as in the "
                         "UNet, and LoKr runs each such set as one grouped launch -- what the modules do behind the reference API)")
    ap.add_argument("--autocast", action="store_true",
                    help="sd-scripts' mixed_precision=bf16: the layers that read a LayerNorm output (attn1 to_q / to_k / to_v, "
                         "ff.net.0.proj: 280 of the 788) get an fp32 input and the forward runs under torch.autocast(bf16) -- the reference's F.linear runs in "
                         "the autocast dtype (modules/lokr.py:543-566); the native ops cast once per op, a sibling set once per set")
    ap.add_argument("--shared-inputs", action="store_true",
                    help="development: instances of one shape share x / g (cache-resident, the round-1 behaviour)")
    ap.add_argument("--eager", action="store_true", help="time the step without hipGraph capture (Python-driven)")
    ap.add_argument("--segments", type=int, default=0,
                    help="backward graph segments at N > 1 (collectives in between); 0 = cut at the gradient-bucket boundaries, one "
                         "segment per bucket (default), K > 0 = K equal segments (rounds 2-3)")
    ap.add_argument("--collective", default="all_reduce", choices=["all_reduce", "reduce_scatter"],
                    help="per bucket: one in-place all-reduce (default) or an in-place reduce-scatter + all-gather pair")
    ap.add_argument("--layers", default="all", help="'all', 'linear' or 'conv' (development)")
    ap.add_argument("--host-timing", action="store_true", help="development: also report the host time to SUBMIT one step (config.host_submit_ms)")
    ap.add_argument("--collectives-on-side-stream", action="store_true",
                    help="N > 1 with the ProcessGroup-free communicator: bucket collectives on the communicator's own stream between the "
                         "bucket-aligned backward segments (overlap).  Default for the captured step: on the compute stream behind the backward "
                         "graph -- a second stream that waits on the graph-launching stream slows every graph launch (profiles/r05_ws1_*)")
    ap.add_argument("--collectives-inline", action="store_true",
                    help="N > 1: force the bucket collectives onto the compute stream behind the backward graph (default: decided per payload "
                         "by lycoris_amd.grad_sync.overlap_pays)")
    ap.add_argument("--collectives-on-main-stream", action="store_true",
                    help="development: issue the bucket collectives on the compute stream (no side stream, no overlap) -- separates the "
                         "cost of the collectives themselves from the cost of running them beside the backward pass")
    ap.add_argument("--per-tensor-optimizer", action="store_true",
                    help="A/B: AdamW over the individual parameter tensors instead of the flat parameter arena (AdapterGradSync.flat_parameters)")
    ap.add_argument("--force-segments", action="store_true",
                    help="development: cut the backward into --segments graphs at N = 1 too (the N > 1 replay structure without the collectives)")
    ap.add_argument("--rccl-ws1", action="store_true",
                    help="run the N > 1 step on one GPU: nccl (= RCCL) process group of world_size 1, every gradient bucket all-reduced "
                         "in place on the side stream between the backward segment graphs (MASTER_ADDR / MASTER_PORT / RANK from the env)")
    ap.add_argument("--dev-base-only", action="store_true", help="development: run the base + adapter leg alone and exit (for rocprofv3 --kernel-trace)")
    ap.add_argument("--dev-idle-comm", action="store_true", help="development: create the RCCL communicator, never use it (plain N = 1 step)")
    ap.add_argument("--dev-hop-only", action="store_true", help="development: --rccl-ws1 with the collectives themselves skipped (marks, waits, join only)")
    ap.add_argument("--comm-high-priority", action="store_true", help="development: the communicator's stream at high priority (A/B: it slows the whole step 3.5x)")
    ap.add_argument("--comm-torch-stream", action="store_true", help="development: the communicator enqueues on a torch pool stream")
    ap.add_argument("--capture-collectives", action="store_true",
                    help="--backend rccl only: record the bucket collectives INTO one backward hipGraph (the communicator's stream is "
                         "forked into the capture); default: they are enqueued between the replays of the backward segment graphs")
    ap.add_argument("--backend", default="rccl", choices=["rccl", "nccl", "gloo"],
                    help="'rccl' (default): the ProcessGroup-free communicator of lycoris_amd.grad_sync (ncclCommInitRank once, "
                         "collectives as plain stream work; no torch.distributed process group exists); 'nccl': the same collectives "
                         "through c10d's ProcessGroupNCCL (rounds 3-4, A/B); 'gloo' (development) lets N ranks share ONE GPU "
                         "(rank % device_count) to exercise the N > 1 control flow on a single-GPU box")
    ap.add_argument("--pmc-pass", type=int, default=0,
                    help="run N eager compute passes and exit (for rocprofv3 --pmc, which cannot sample inside graph replays)")
    a = ap.parse_args()
    if a.preset == "mixed":
        a.algo = "mixed"
    return a


# ---------------------------------------------------------------------------------------------------------------------
# workload
# ---------------------------------------------------------------------------------------------------------------------
def layer_specs(model, algo):
    """[(spec, algo_of_layer, count)].  (IA)^3 follows the reference's own preset (config.py:185-195: to_k / to_v on
    the output side, ff.net.2 on the input side); the mixed preset is BASELINE.json configs[4]."""
    base = sdxl_unet_layers(1) if model == "sdxl" else sd15_unet_layers(4)
    if algo == "ia3":
        if model != "sdxl":
            raise SystemExit("--algo ia3 is defined on the SDXL shape list")
        mk = lambda n, M, I, O, side, tag: (dict(kind="linear", M=M, I=I, O=O, count=n, tag=tag, side=side), "ia3", n)
        return [mk(120, 1024, 1280, 1280, "out", "attn1 to_k/to_v @1280"), mk(20, 4096, 640, 640, "out", "attn1 to_k/to_v @640"),
                mk(120, 77, 2048, 1280, "out", "attn2 to_k/to_v @1280"), mk(20, 77, 2048, 640, "out", "attn2 to_k/to_v @640"),
                mk(60, 1024, 5120, 1280, "in", "ff.net.2 @1280"), mk(10, 4096, 2560, 640, "in", "ff.net.2 @640")]
    out = []
    for s in base:
        a = algo
        if algo == "mixed":
            a = "locon" if "attn" in s["tag"] else ("lokr" if "ff.net" in s["tag"] else "ia3")
            s = dict(s, side="out")
        out.append((s, a, s["count"]))
    return out


class Inst:
    """One adapted layer instance: its own input x, upstream gradient g and fp32 adapter factors."""

    def __init__(self, spec, algo, dtype, dev, gen, share=None, share_x=None):
        from lycoris_amd import ops
        self.ops, self.spec, self.algo = ops, spec, algo
        M, _, O = layer_rows(spec)
        lin = spec["kind"] == "linear"
        if lin:
            xs, gs = (spec["M"], spec["I"]), (spec["M"], O)
            cin, ksz = spec["I"], ()
        else:
            ho = (spec["H"] + 2 * spec["pad"] - spec["k"]) // spec["stride"] + 1
            xs, gs = (spec["B"], spec["C"], spec["H"], spec["W"]), (spec["B"], O, ho, ho)
            cin, ksz = spec["C"], (spec["k"], spec["k"])
        self.cin, self.ksz, self.O = cin, ksz, O
        self.sibs = None  # [leader, ...]: the instances that read the same tensor (set by build_instances)
        if share is not None:
            self.x, self.g, self.base = share.x, share.g, share.base
        else:
            self.base = None
            if algo == "ia3" and spec.get("side", "out") == "out":
                # out-side (IA)^3 acts on the frozen layer's OUTPUT: x is not read at all
                self.base = torch.randn(*gs, device=dev, dtype=dtype, generator=gen).requires_grad_(True)
                self.x = self.base
            elif share_x is not None:
                self.x = share_x.x  # a sibling projection: the same tensor object as its set's first member
            else:
                # --autocast: a LayerNorm runs in fp32 under torch.autocast and hands its fp32 output to the projections behind it
                xdt = torch.float32 if (AUTOCAST and lin and any(t in spec.get("tag", "") for t in ("attn1 to_q", "ff.net.0"))) else dtype
                self.x = torch.randn(*xs, device=dev, dtype=xdt, generator=gen).requires_grad_(True)
            gshape = xs if (algo == "ia3" and spec.get("side") == "in") else gs
            self.g = torch.randn(*gshape, device=dev, dtype=dtype, generator=gen) / math.sqrt(O)
            if not lin and CHANNELS_LAST:
                self.x = self.x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
                self.g = self.g.contiguous(memory_format=torch.channels_last)
                if self.base is not None:
                    self.base = self.x
        f32 = dict(device=dev, dtype=torch.float32, generator=gen)
        if algo == "lokr" and LOKR_RANK > 0 and LOKR_RANK < max(O // FACTOR, cin // FACTOR) / 2:
            # low-rank w2 (modules/lokr.py:131-136): lokr_w2_a [c, r], lokr_w2_b [r, d * k * k]; the product is formed per call
            kk = ksz[0] * ksz[1] if ksz else 1
            ps = [torch.randn(FACTOR, FACTOR, **f32) * 0.3, torch.randn(O // FACTOR, LOKR_RANK, **f32) * 0.1,
                  torch.randn(LOKR_RANK, (cin // FACTOR) * kk, **f32) * 0.1]
        elif algo == "lokr":  # factor 8, full-matrix w2 (lora_dim >= 10000): w1 [8,8], w2 [O/8, I/8(,k,k)]
            w2 = torch.randn(O // FACTOR, cin // FACTOR, *ksz, **f32) * 0.05
            if ksz:  # conv factor kept in channels_last memory: the implicit-GEMM kernels read / write it in place
                w2 = w2.contiguous(memory_format=torch.channels_last)
            ps = [torch.randn(FACTOR, FACTOR, **f32) * 0.3, w2]
        elif algo == "locon":  # dim 16 / conv_dim 8
            r = 16 if not ksz or ksz == (1, 1) else 8
            down = torch.randn(r, cin, *ksz, **f32) * 0.05
            if ksz:
                down = down.contiguous(memory_format=torch.channels_last)
            ps = [down, torch.randn(O, r, *([1] * len(ksz)), **f32) * 0.05]
        elif algo == "loha":  # dim 32
            r, kk = 32, (ksz[0] * ksz[1] if ksz else 1)
            ps = [torch.randn(O, r, **f32) * 0.1, torch.randn(r, cin * kk, **f32), torch.randn(O, r, **f32) * 0.1,
                  torch.randn(r, cin * kk, **f32)]
        else:  # ia3: one vector over the scaled channel
            C = cin if spec.get("side") == "in" else O
            ps = [torch.randn(*((1, C, 1, 1) if ksz else (C,)), **f32) * 0.1]
            self.bias = torch.randn(O, **f32) * 0.1 if spec.get("side", "out") == "out" else None
        self.params = [torch.nn.Parameter(p) for p in ps]

    def forward(self, base=None):
        """the adapter delta; with `base` (the frozen layer's output) base + delta, fused where the kernels can"""
        ops, s, p = self.ops, self.spec, self.params
        lin = s["kind"] == "linear"
        if self.algo == "lokr" and len(p) == 3 and lin:  # low-rank w2 on nn.Linear: the factors go to the kernels as they are
            if base is not None:
                return ops.lokr_linear_lr(self.x, p[0], p[1], p[2], 1.0, base=base)
            return ops.lokr_linear_lr(self.x, p[0], p[1], p[2], 1.0)
        if self.algo == "lokr" and len(p) == 3:  # Conv2d low-rank w2: planes from the factors where the patch kernels take the layer
            if base is not None:
                return base + self.forward()
            return ops.lokr_conv2d_lr(self.x, p[0], p[1], p[2], 1.0, self.ksz, (s["stride"],) * 2, (s["pad"],) * 2, (1, 1))
        if base is not None:
            if lin and self.algo == "lokr":
                return ops.lokr_linear(self.x, p[0], p[1], 1.0, base=base)
            return base + self.forward()
        if self.algo == "ia3":
            chan = -1 if lin else 1
            if s.get("side", "out") == "out":  # y = base * (1 + w) - bias * w   (rebuild semantics, no GEMM)
                return ops.chan_affine(self.base, p[0], self.bias, 1.0, 1.0, chan)
            return ops.chan_affine(self.x, p[0], None, 0.0, 1.0, chan)  # x * w; the dense op on it is the frozen GEMM
        if lin:
            if self.algo == "lokr":
                return ops.lokr_linear(self.x, p[0], p[1], 1.0)
            if self.algo == "locon":
                return ops.locon_linear(self.x, p[0], p[1], 1.0)
            return ops.loha_linear(self.x, p[0], p[1], p[2], p[3], 1.0)
        st, pd, dl = (s["stride"],) * 2, (s["pad"],) * 2, (1, 1)
        if self.algo == "lokr":
            return ops.lokr_conv2d(self.x, p[0], p[1], 1.0, st, pd, dl)
        if self.algo == "locon":
            return ops.locon_conv2d(self.x, p[0], p[1], 1.0, st, pd, dl)
        return ops.loha_conv2d(self.x, p[0], p[1], p[2], p[3], 1.0, (s["O"], s["C"], s["k"], s["k"]), st, pd, dl)

    # ---- the reference's call sequence on the same tensors (PyTorch-ROCm eager comparator) ----------------------
    def ensure_weight(self):
        if getattr(self, "W", None) is None:
            shape = (self.O, self.cin, *self.ksz)
            self.W = torch.randn(*shape, device=self.x.device, dtype=self.x.dtype) * (1.0 / math.sqrt(self.cin * max(1, len(self.ksz) and self.ksz[0] * self.ksz[1])))
        return self.W

    def base_forward(self, x=None):
        x = self.x if x is None else x
        s = self.spec
        if s["kind"] == "linear":
            return F.linear(x, self.ensure_weight())
        return F.conv2d(x, self.ensure_weight(), None, s["stride"], s["pad"])

    def reference_forward(self):
        """modules/{lokr,locon,loha}.py forward(), rebuild path: dW from the factors in the fp32 parameter dtype, cast to
        the base-weight dtype, `W + dW - W`, dense op (lokr.py:543-566, locon.py:309-332, loha.py:301-322)."""
        W, p = self.ensure_weight(), self.params
        if self.algo == "lokr":
            w2 = p[1] if len(p) == 2 else (p[1] @ p[2]).reshape(p[1].shape[0], self.cin // FACTOR, *self.ksz)
            f1 = p[0].reshape(*p[0].shape, *([1] * (w2.dim() - 2)))
            dW = torch.kron(f1, w2.contiguous())
        elif self.algo == "locon":
            dW = (p[1].reshape(p[1].shape[0], -1) @ p[0].reshape(p[0].shape[0], -1)).reshape(W.shape)
        elif self.algo == "loha":
            dW = ((p[0] @ p[1]) * (p[2] @ p[3])).reshape(W.shape)
        else:
            raise KeyError(self.algo)
        delta_w = (W + dW.to(W.dtype)) - W
        s = self.spec
        if s["kind"] == "linear":
            return F.linear(self.x, delta_w)
        return F.conv2d(self.x, delta_w, None, s["stride"], s["pad"])


CHANNELS_LAST = False
AUTOCAST = False
LOKR_RANK = 0
OVERLAP_LEG = False
SIBLINGS = True
OWN_BASE = True   # base + adapter leg: the frozen nn.Linear inside the LoKr adapter's autograd node (ops.lokr_adapted_linear); --no-own-base: two nodes


def adapter_param_count(args):
    """adapter parameters of the workload from the shape list alone (the DP payload is 4 bytes each: fp32 gradients)"""
    n = 0
    for spec, algo, count in layer_specs(args.model, args.algo):
        if (args.layers == "linear" and spec["kind"] != "linear") or (args.layers == "conv" and spec["kind"] != "conv"):
            continue
        _, _, O = layer_rows(spec)
        lin = spec["kind"] == "linear"
        cin = spec["I"] if lin else spec["C"]
        kk = 1 if lin else spec["k"] * spec["k"]
        if algo == "lokr":
            r = int(args.rank)
            if r > 0 and r < max(O // FACTOR, cin // FACTOR) / 2:
                per = FACTOR * FACTOR + (O // FACTOR) * r + r * (cin // FACTOR) * kk
            else:
                per = FACTOR * FACTOR + (O // FACTOR) * (cin // FACTOR) * kk
        elif algo == "locon":
            r = 16 if kk == 1 else 8
            per = r * cin * kk + O * r
        elif algo == "loha":
            per = 2 * 32 * (O + cin * kk)
        else:
            per = cin if spec.get("side") == "in" else O
        n += per * count
    return n


def build_instances(args, dtype, dev):
    global CHANNELS_LAST, LOKR_RANK, OVERLAP_LEG, SIBLINGS, AUTOCAST, OWN_BASE
    AUTOCAST = bool(getattr(args, "autocast", False))
    OVERLAP_LEG = bool(getattr(args, "overlap_leg", False))
    SIBLINGS = not getattr(args, "no_siblings", False)
    OWN_BASE = not getattr(args, "no_own_base", False)
    CHANNELS_LAST = bool(args.channels_last)
    LOKR_RANK = int(args.rank)
    gen = torch.Generator(device=dev).manual_seed(1234)
    insts = []
    for spec, algo, count in layer_specs(args.model, args.algo):
        if args.layers == "linear" and spec["kind"] != "linear":
            continue
        if args.layers == "conv" and spec["kind"] != "conv":
            continue
        first = None
        sib = int(spec.get("sib", 1)) if (SIBLINGS and spec["kind"] == "linear" and not (algo == "ia3" and spec.get("side", "out") == "out")) else 1
        grp = []
        for k in range(count):
            lead = grp[0] if (sib > 1 and k % sib != 0) else None
            inst = Inst(spec, algo, dtype, dev, gen, share=first if args.shared_inputs else None, share_x=lead)
            first = first or inst
            if sib > 1:
                if lead is None:
                    grp = [inst]
                else:
                    grp.append(inst)
                inst.sibs = grp
            insts.append(inst)
    for pos, it in enumerate(insts):
        it.pos = pos
    return insts


def _groupable(it):
    """a sibling set that goes out as ONE launch: LoKr (full-matrix or low-rank w2) or LoCon on nn.Linear -- ops.lokr_linear_group /
    lokr_linear_lr_group / locon_linear_group, the ops the modules' sibling sets call (lycoris_amd/modules/siblings.py)"""
    return (SIBLINGS and it.sibs is not None and len(it.sibs) > 1 and it.spec["kind"] == "linear"
            and (it.x.dtype != torch.float32 or AUTOCAST)   # (an fp32 activation under a 16-bit autocast: cast once per set by the op)
            and ((it.algo == "lokr" and len(it.params) in (2, 3)) or it.algo == "locon"))


def forward_all(insts, with_base=False):
    """[(output, instance)] in layer order.  A sibling set whose members are all in `insts` is launched by its first member."""
    if AUTOCAST and not torch.is_autocast_enabled("cuda"):
        with torch.autocast("cuda", dtype=insts[0].g.dtype):
            return forward_all(insts, with_base)
    present = {id(it) for it in insts}
    outs, parked = [], {}
    for it in insts:
        if id(it) in parked:
            outs.append((parked.pop(id(it)), it))
            continue
        grp = it.sibs
        if grp is not None and grp[0] is it and _groupable(it) and all(id(m) in present for m in grp):
            if with_base and OWN_BASE and it.algo == "lokr" and len(it.params) == 2:  # frozen layers + adapters of the set: ONE autograd node
                ys = it.ops.lokr_adapted_linear(it.x, [m.ensure_weight() for m in grp], [None] * len(grp), [m.params[0] for m in grp],
                                                [m.params[1] for m in grp], [1.0] * len(grp))
                for m, y in zip(grp[1:], ys[1:]):
                    parked[id(m)] = y
                outs.append((ys[0], it))
                continue
            bases = [m.base_forward() for m in grp] if with_base else None
            if it.algo == "locon":  # (no fused epilogue: base + delta is an elementwise add, as in the modules)
                ys = it.ops.locon_linear_group(it.x, [m.params[0] for m in grp], [m.params[1] for m in grp], [1.0] * len(grp))
                if bases is not None:
                    ys = [b + y for b, y in zip(bases, ys)]
            elif len(it.params) == 2:
                ys = it.ops.lokr_linear_group(it.x, [m.params[0] for m in grp], [m.params[1] for m in grp], [1.0] * len(grp), bases)
            else:  # --rank: low-rank w2 = w2_a @ w2_b
                ys = it.ops.lokr_linear_lr_group(it.x, [m.params[0] for m in grp], [m.params[1] for m in grp], [m.params[2] for m in grp],
                                                 [1.0] * len(grp), bases)
            for m, y in zip(grp[1:], ys[1:]):
                parked[id(m)] = y
            outs.append((ys[0], it))
        elif with_base and OWN_BASE and it.algo == "lokr" and len(it.params) == 2 and it.spec["kind"] == "linear":
            outs.append((it.ops.lokr_adapted_linear(it.x, [it.ensure_weight()], [None], [it.params[0]], [it.params[1]], [1.0])[0], it))
        else:
            outs.append((it.forward(base=it.base_forward()) if with_base else it.forward(), it))
    return outs


def _wrt(items, factors=True):
    """the tensors a backward call differentiates with respect to: each instance's input (once per tensor: siblings share it) and,
    with `factors`, its parameters"""
    seen, out = set(), []
    for it in items:
        for t in [it.x] + (list(it.params) if factors else []):
            if id(t) not in seen:
                seen.add(id(t))
                out.append(t)
    return out


def set_aligned(insts, e):
    """the largest position <= e that does not cut a sibling set (its members are one autograd node)"""
    if 0 < e < len(insts) and insts[e].sibs is not None and _groupable(insts[e]):
        return insts[e].sibs[0].pos
    return e


def backward_range(outs, lo, hi):
    """backward of outs[lo:hi] in reverse order.  autograd.grad instead of .backward(): dx is produced and dropped (a
    real UNet hands it to the previous layer); the factor gradients go straight into the arena (fused accumulation: the kernels
    add them into `.grad` and hand nothing back -- hence allow_unused; the factors are listed because a backward call computes
    only what it is asked for, csrc/torch_ops.cpp)."""
    seg = outs[lo:hi][::-1]
    if seg:  # ONE engine invocation for the whole segment, as loss.backward() is one for a real network
        torch.autograd.grad([y for y, _ in seg], _wrt([it for _, it in seg]), [it.g for _, it in seg], allow_unused=True)


def lib_sha():
    from lycoris_amd import _native
    h = hashlib.sha256()
    with open(_native.lib_path(), "rb") as f:
        h.update(f.read())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it (no WORLD_SIZE in the environment): re-run this very command line
    as N ranks under torch.distributed.run -- one process per GPU, rendezvous on 127.0.0.1 -- and pass rank 0's JSON line through.
    The driver's own form (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) sets WORLD_SIZE and never
    comes here."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev_index = local_rank if args.backend != "gloo" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    comm = None
    if world > 1 or args.rccl_ws1:
        if args.backend == "rccl":  # no process group at all: the 128-byte id travels through the launcher's store
            from lycoris_amd.grad_sync import RcclCommunicator
            # captured steps: overlap (side stream between bucket-aligned backward segments) when the estimated ring time of the
            # exchange exceeds what the second queue costs, inline (compute stream, behind the backward graph) otherwise --
            # grad_sync.overlap_pays; --collectives-on-side-stream / --collectives-inline force either.  Eager steps always overlap
            # (the hooks drive the buckets).  The payload is known before the instances exist: 4 bytes per adapter parameter.
            from lycoris_amd.grad_sync import exchange_estimate_ms, overlap_pays, SIDE_STREAM_HOP_MS, XGMI_ALLREDUCE_BUSBW_GBS
            payload = 4 * adapter_param_count(args)
            want_overlap = overlap_pays(payload, max(world, 2 if args.rccl_ws1 else world))
            if args.collectives_on_side_stream:
                want_overlap = True
            if args.collectives_inline or (args.rccl_ws1 and not args.collectives_on_side_stream):
                want_overlap = False  # (--rccl-ws1 measures ordering cost at one rank: inline unless the A/B flag asks otherwise)
            inline = not (want_overlap or args.capture_collectives or args.eager)
            POLICY.update({"mode": "inline" if inline else ("captured" if args.capture_collectives else "overlap"),
                           "payload_mb": round(payload / 2**20, 1), "est_ring_ms": round(exchange_estimate_ms(payload, world), 3),
                           "hop_ms": SIDE_STREAM_HOP_MS, "est_busbw_gbs": XGMI_ALLREDUCE_BUSBW_GBS,
                           "rule": "overlap (communicator's own stream between bucket-aligned backward segments) when est_ring_ms > hop_ms, "
                                   "inline (compute stream behind the backward graph) otherwise; forced by --collectives-on-side-stream / "
                                   "--collectives-inline", "forced": bool(args.collectives_on_side_stream or args.collectives_inline)})
            kw = dict(high_priority=args.comm_high_priority, stream=torch.cuda.Stream(device=dev) if args.comm_torch_stream else None,
                      on_current_stream=inline)
            comm = RcclCommunicator.from_env(dev, **kw) if world > 1 else RcclCommunicator(0, 1, dev, **kw)
            assert comm.world == world and comm.rank == rank
        elif args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    if args.dev_idle_comm:
        from lycoris_amd.grad_sync import RcclCommunicator
        _KEEP.append(RcclCommunicator(0, 1, dev))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]

    from lycoris_amd.grad_sync import AdapterGradSync

    insts = build_instances(args, dtype, dev)
    all_params = [p for it in insts for p in it.params]
    # --rccl-ws1: the N > 1 step on ONE GPU -- a world_size-1 RCCL group, every bucket really all-reduced (AVG, in place, side
    # stream) between the replays of the backward segment graphs
    sync = AdapterGradSync(all_params, bucket_bytes=32 << 20, always_reduce=bool(args.rccl_ws1), collective=args.collective, comm=comm)
    if args.dev_hop_only:
        sync._comm_reduce = lambda flat: None
    if args.collectives_on_main_stream and comm is None:
        sync.side_stream = None
    sync.attach_fused()  # kernels accumulate into the arena and report to the bucket counters (eager: overlap by hooks)
    from lycoris_amd import ops as _ops
    if args.no_defer:  # A/B: one weight-gradient launch per layer instead of the grouped launches
        _ops.deferred_weight_gradients(False)
    if args.no_planes:
        _ops.lokr_planes_cache(False)
    # fused AdamW over ONE flat leaf per dtype (the parameters are views of it, `.grad` is the gradient arena): the update is one
    # multi-tensor chunk stream instead of a walk over 1 576 small tensors (--per-tensor-optimizer: the round 1-3 form, A/B)
    opt = torch.optim.AdamW(all_params if args.per_tensor_optimizer else sync.flat_parameters(), lr=1e-4, fused=True)
    n_layers = len(insts)
    act_bytes = sum(t.numel() * t.element_size() for t in {id(t): t for it in insts for t in (it.x, it.g)}.values())

    if args.dev_base_only:
        print(json.dumps(base_leg(insts, sync, opt, _ops)), flush=True)
        return
    if args.pmc_pass:
        for _ in range(args.pmc_pass):
            sync.zero_grad()
            backward_range(forward_all(insts), 0, n_layers)
            sync.finish()
        torch.cuda.synchronize()
        return

    # ---- one step, eager (Python-driven; collectives launched from inside the backward by the bucket counters) --------
    def eager_step():
        sync.zero_grad()
        backward_range(forward_all(insts), 0, n_layers)
        sync.finish()
        opt.step()

    # ---- one step, captured: forward graph + K backward segment graphs; between the segments the buckets whose
    #      gradients are complete are all-reduced on the side stream (N > 1), overlapping the later segments ----------
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eager_step()  # warm-up (allocator, lazy init)
        # the second step of a training run refreshes every cached operand plane in one launch and writes the descriptor table of that
        # launch (csrc/torch_ops.cpp refresh_planes_locked); done here so that the capture below holds the pack launch only
        _ops.refresh_lokr_planes(force=True)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    graphs, seg_bounds = [], []
    # capture_error_mode="thread_local" on every capture below: with a process group alive, c10d's watchdog THREAD polls the events of
    # the warm-up step's collectives; in the default "global" mode such a query during a capture is an error and aborts the process
    # (hipErrorStreamCaptureUnsupported: seen once with --collective reduce_scatter, profiles/r04_final_ws1.log) -- a hazard for every
    # N > 1 run, whatever the collective
    if not args.eager:
        sync._sync_enabled = False  # inside a capture nothing may be launched from the callbacks
        pool = torch.cuda.graph_pool_handle()
        g_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_fwd, pool=pool, capture_error_mode="thread_local"):
            for arena in sync.arenas.values():
                arena.zero_()
            # the optimizer changed every factor: repack the cached LoKr operand planes (one grouped launch per 28 factors) --
            # part of the step, captured here so that every replay packs the parameters of ITS step
            _ops.refresh_lokr_planes(force=True)
            outs = forward_all(insts)
        inline_comm = comm is not None and comm.on_current_stream
        multi = (world > 1 or args.force_segments or args.rccl_ws1) and not (inline_comm and not args.force_segments)
        # segment edges (layer positions, descending) and the buckets each segment completes.  Default: cut exactly where a
        # gradient bucket becomes complete -- one segment per bucket (SDXL LoKr: 5), no segment boundary that launches nothing
        # (a sibling set is ONE autograd node: its parameters are complete when the set's first member has been reached, and no
        #  segment edge may fall inside it)
        order = {p: (it.sibs[0].pos if (it.sibs is not None and _groupable(it)) else i) for i, it in enumerate(insts) for p in it.params}
        cuts = sync.bucket_boundaries(order)
        if not multi:
            edges = [0]
        elif args.segments > 0:
            nseg = max(1, min(args.segments, n_layers))
            edges = sorted({set_aligned(insts, round(i * n_layers / nseg)) for i in range(nseg)}, reverse=True)
        else:
            edges = sorted(set(cuts), reverse=True)
        plan, done = [], set()
        for e in edges:
            ready = [i for i, c in enumerate(cuts) if c >= e and i not in done]
            done.update(ready)
            plan.append(ready)
        # (the collectives themselves are NOT captured: recording RCCL's all-reduce into the backward graph on a forked side stream
        # segfaults on this stack -- torch 2.10 + rocm 7.0, world_size-1 group, profiles/r04_ws1_variants.log)
        hi = n_layers
        captured_collectives = bool(args.capture_collectives and comm is not None and multi)
        if captured_collectives:
            # round 5: with the ProcessGroup-free communicator a collective is plain stream work, so the whole backward pass -- the
            # segments AND the bucket collectives behind them -- is ONE graph: wait_current() forks the communicator's stream into the
            # capture behind the segment that completes the bucket, join() brings it back before the capture ends.  The host then
            # submits two graphs per step and never talks to RCCL.
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, pool=pool, capture_error_mode="thread_local"):
                for e, ready in zip(edges, plan):
                    backward_range(outs, e, hi)
                    sync.launch_buckets(ready)
                    hi = e
                comm.join()
            sync._comm_pending = False
            graphs.append(gph)
            plan = [[]]
            multi = False  # nothing is left to do between replays
        else:
            for e in edges:  # backward runs from the last layer to the first
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph, pool=pool, capture_error_mode="thread_local"):
                    backward_range(outs, e, hi)
                graphs.append(gph)
                hi = e
        sync._sync_enabled = True

        seg_done = [None] * len(graphs)

        def step():
            sync._reset_pending()
            g_fwd.replay()
            prev = None
            for k, (gph, ready) in enumerate(zip(graphs, plan)):
                gph.replay()
                if multi:
                    seg_done[k] = sync.mark()
                    if prev is not None:  # the collectives of segment k - 1 are enqueued AFTER segment k has been submitted: the
                        sync.launch_buckets(plan[prev], after=seg_done[prev])  # GPU runs it while the host talks to RCCL
                    prev = k
            if multi and prev is not None:
                sync.launch_buckets(plan[prev], after=seg_done[prev])
            if captured_collectives:
                sync._reset_pending()  # (the collectives and the join are nodes of the backward graph)
            else:
                sync.finish()
            opt.step()
    else:
        step = eager_step

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            if comm is not None:
                comm.barrier()
            else:
                dist.barrier()
        torch.cuda.synchronize()

    c0 = sync.collectives_launched
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1 and comm is not None:
        elapsed = comm.max_over_ranks(elapsed)
    elif world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    ms_per_step = elapsed / args.steps * 1e3
    host_submit_ms = None
    if args.host_timing:  # how long does the HOST need to submit one step (no waiting for the GPU)?  3 steps from an idle queue
        barrier()
        th = time.perf_counter()
        for _ in range(3):
            step()
        host_submit_ms = (time.perf_counter() - th) / 3 * 1e3
        barrier()
    value = world * args.steps / elapsed  # whole job: every rank processes its own batch (weak scaling)

    model_name = "SDXL UNet 1024x1024 bs=1/GPU" if args.model == "sdxl" else "SD1.5 UNet 512x512 bs=4/GPU"
    n_lin = sum(1 for it in insts if it.spec["kind"] == "linear")
    result = {
        # BASELINE.json's metric, verbatim; `value` = the hot path of this repository (adapter fwd + bwd + grad sync + AdamW over every
        # adapted layer), `value_base_plus_adapter` = the same step with the frozen layers' own ops in it
        "metric": "SDXL UNet train steps/sec (bs=1/GPU) per algo at 1/2/4/8 MI355X",
        "metric_detail": f"{'SDXL' if args.model == 'sdxl' else 'SD1.5'} UNet adapter train steps/sec "
                         f"(bs={'1' if args.model == 'sdxl' else '4'}/GPU), " + ALGO_LABEL[args.algo]
                         + "; value = adapter-only step (frozen UNet ops not timed), value_base_plus_adapter = with them",
        "value": round(value, 3), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {
            "workload": f"{model_name}, preset {'full' if args.algo not in ('ia3', 'mixed') else args.algo}: {n_layers} adapted "
                        f"layers ({n_lin} Linear + {n_layers - n_lin} Conv2d), adapter fwd+bwd + grad all-reduce + fused "
                        "AdamW; frozen UNet ops not timed (see base_plus_adapter)",
            "algo": args.algo, "factor": FACTOR if args.algo in ("lokr", "mixed") else None, "layers": n_layers,
            "lokr_w2": (f"low rank {args.rank} (lokr_w2_a @ lokr_w2_b: planes packed from the factors, chain rule in the grouped launch)" if args.rank else "full matrix") if args.algo in ("lokr", "mixed") else None,
            "adapter_params": sum(p.numel() for p in all_params), "dp_payload_mb": round(sync.payload_bytes / 2**20, 1),
            "parallelism": f"dp{world}", **({"parallelism_detail": dict(POLICY)} if POLICY else {}),
            **({"rccl_ranks": comm.count()} if comm is not None else {}), **({"host_submit_ms": round(host_submit_ms, 2)} if host_submit_ms is not None else {}),
            "optimizer": "torch.optim.AdamW(fused=True) over " + ("the individual parameter tensors" if args.per_tensor_optimizer else
                                                                  "one flat parameter arena per dtype (AdapterGradSync.flat_parameters)"),
            "graph": "eager (no capture)" if args.eager else
                     f"hipGraph replay: 1 forward graph + {len(graphs)} backward segment(s)"
                     + ((", bucket collectives (" + args.collective + ") "
                         + ("captured inside the backward graph" if (not args.eager and captured_collectives) else
                            ("issued between the bucket-aligned segments" if args.segments <= 0 else "issued between the segments"))
                         + (" on the communicator's own stream (ProcessGroup-free RCCL, lycoris_amd.grad_sync.RcclCommunicator)"
                            if comm is not None else " on a side stream (c10d " + args.backend + ")")) if ((world > 1 or args.rccl_ws1) and multi) else "")
                     + ((", bucket collectives (" + args.collective + ") on the compute stream behind the backward graph (ProcessGroup-free RCCL, "
                         "lycoris_amd.grad_sync.RcclCommunicator; no second queue: --collectives-on-side-stream for the overlapped form)")
                        if ((world > 1 or args.rccl_ws1) and not args.eager and inline_comm and not multi) else ""),
            "conv_memory_format": "channels_last (documented default, DESIGN.md 2; --nchw for the A/B leg)" if args.channels_last else "contiguous (NCHW)",
            "inputs": "shared per shape (cache-resident)" if args.shared_inputs else
                      (f"every layer instance owns its upstream gradient g; inputs x as in the UNet: the to_q / to_k / to_v layers of a "
                       f"self-attention and to_k / to_v of a cross-attention read ONE tensor ({sum(1 for it in insts if it.sibs is not None)} of "
                       f"{n_layers} layers in {sum(1 for it in insts if it.sibs is not None and it.sibs[0] is it)} sibling sets), every other "
                       f"layer its own: {act_bytes / 1e9:.2f} GB of distinct activations per step" if SIBLINGS else
                       f"distinct x / g per layer instance (round 1-4 layout): {act_bytes / 1e9:.2f} GB read per step"),
            "sibling_sets": ("LoKr nn.Linear sibling sets run as ONE grouped launch each (ops.lokr_linear_group: one forward launch, one autograd "
                             "node, one backward dx launch) -- the op the modules' sibling sets call behind the reference API "
                             "(lycoris_amd/modules/siblings.py)" if (SIBLINGS and any(_groupable(it) for it in insts)) else
                             ("per-layer launches (sets share their input only)" if SIBLINGS else "off (--no-siblings)")),
        },
    }
    extra = rank == 0 and world == 1 and not args.eager
    sync._sync_enabled = False  # the measurement legs below repeat passes without finish(): no bucket bookkeeping there
    if extra and not args.no_roofline and not args.rank:  # (the low-rank leg reports the step only: its Linear kernels are the same)
        result["roofline"] = roofline(insts, args, dtype, dev)
        mb = pmc_mfma(f"{args.algo}/{args.model}/linear")  # north_star: "MFMA utilisation reported against gfx950 peak" (counter pass)
        if mb and result["roofline"] is not None:
            result["roofline"]["mfma_busy"] = mb
    if extra and not args.no_reference and args.algo in ("lokr", "locon", "loha"):
        result["reference_rocm_eager"] = reference_leg(insts, sync, ms_per_step)
    if extra and not args.no_base and args.algo in ("lokr", "locon", "loha"):
        result["base_plus_adapter"] = base_leg(insts, sync, opt, _ops)
        # SURVEY 8d defines the step WITH the frozen layers' forward + dx ops, the arena fill and AdamW: the contract number next
        # to `value`
        result["value_base_plus_adapter"] = round(1e3 / result["base_plus_adapter"]["step_ms"], 3)
    if extra and not args.no_per_algo and args.algo == "lokr" and args.model == "sdxl" and not args.rank and args.layers == "all":
        result["per_algo"] = per_algo_legs()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.algo in ("lokr", "locon", "loha"):
        result["cpu_baseline"] = cpu_baseline(args.algo, args.model)
    # the figures a reader needs from the line's scalars / `roofline` alone (the driver keeps those): eager-vs-eager speedup over the
    # reference's torch call sequence, the contract step with the frozen layers in it, one {ms, frac} pair per BASELINE configuration
    ref, bpa, pa = result.get("reference_rocm_eager"), result.get("base_plus_adapter"), result.get("per_algo")
    if ref:
        result["speedup_eager_vs_eager"] = ref["speedup_eager_vs_eager"]
        result["native_eager_ms"], result["reference_eager_ms"] = ref["native_eager_ms"], ref["reference_eager_ms"]
        result["speedup_native_graph_vs_reference_eager"] = ref["speedup_native_graph_vs_reference_eager"]
    if bpa:
        result["adapter_share"], result["base_plus_adapter_step_ms"] = bpa["adapter_share"], bpa["step_ms"]
    if pa:
        for name, leg in pa.items():
            if "ms_per_step" in leg:
                result[f"ms_{name}"] = leg["ms_per_step"]
                if leg.get("roofline", {}).get("frac") is not None:
                    result[f"frac_{name}"] = leg["roofline"]["frac"]
    if result.get("roofline") is not None:
        hl = {k: result[k] for k in ("speedup_eager_vs_eager", "native_eager_ms", "reference_eager_ms", "value_base_plus_adapter",
                                     "adapter_share") if k in result}
        if pa:
            hl["per_algo"] = {name: {"ms": leg.get("ms_per_step"), "frac": leg.get("roofline", {}).get("frac")} for name, leg in pa.items()}
        result["roofline"]["headline"] = hl
    if args.rccl_ws1:
        steps_run = args.warmup + args.steps
        sync.collectives_launched -= c0
        result["config"]["rccl_ws1"] = {"backend": "rccl (ProcessGroup-free communicator)" if comm is not None else "c10d " + dist.get_backend(),
                                        "buckets": len(sync.buckets),
                                        "all_reduces_per_step": sync.collectives_launched / max(1, steps_run),
                                        "note": "RCCL returns early from an in-place collective of a 1-rank communicator: this run measures "
                                                "the ordering / submission cost of the exchange, not ring time" if comm is not None else None}
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)  # RCCL's version banner sits in the C stdio buffer: the JSON line is the LAST line
    if world > 1:
        barrier()  # ... of the whole job: every rank has written what it had buffered before rank 0 prints
    if rank == 0:
        print(json.dumps(result), flush=True)
    if comm is not None:
        comm.destroy()
    elif world > 1 or args.rccl_ws1:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# measurement legs (rank 0, N = 1)
# ---------------------------------------------------------------------------------------------------------------------
_KEEP = []
POLICY = {}


def _graph_ms(fn, reps=4):
    """ms per call of `fn` replayed from a hipGraph, HIP events on the stream the graph is launched on"""
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()  # warm-up on this stream
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=st, capture_error_mode="thread_local"):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gph.replay()
        best = float("inf")
        for _ in range(3):
            e0.record(st)
            for _ in range(reps):
                gph.replay()
            e1.record(st)
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
    torch.cuda.current_stream().wait_stream(st)
    _KEEP.append(gph)  # tensors created during the capture (saved activations of a forward leg) live in its pool
    return best


def _wall_ms(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def pmc_traffic(family, workload):
    """HBM bytes per launch (and launches per pass) of a kernel family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, gfx950 corrections per MI355X_MICROARCH.md; benchmarks/pmc_summary.py --json writes the file).
    Keyed by WORKLOAD ("algo/model/layers", VERDICT r2 weak #11b: an SD1.5 line used to print the SDXL constant) and only trusted
    when it was collected on THIS build of the library (sha of liblycoris_amd.so recorded in the file): otherwise None + why."""
    try:
        with open(PMC_FILE) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None, "no profiles/pmc_traffic.json"
    note = ""
    if rec.get("lib_sha16") != lib_sha():
        # a later build whose difference from the profiled one is written down in the file (which kernels changed, and why their
        # traffic did not) is accepted WITH that note; anything else is not
        also = rec.get("also_valid_for", {}).get(lib_sha())
        if also is None or workload not in also.get("workloads", []):
            return None, f"profiles/pmc_traffic.json was collected on build {rec.get('lib_sha16')}, this is {lib_sha()}"
        note = f"; collected on build {rec.get('lib_sha16')}, running build {lib_sha()}: {also.get('difference', '')}"
    wl = rec.get("workloads", {}).get(workload)
    if wl is None:
        return None, f"profiles/pmc_traffic.json has no pass for workload {workload}"
    fam = wl.get(family)
    return (fam if fam else None), rec.get("source", "profiles/pmc_traffic.json") + note


def pmc_mfma(workload):
    """MFMA utilisation of the workload's kernels from the committed counter pass (profiles/pmc_mfma.json: SQ_VALU_MFMA_BUSY_CYCLES over
    the dispatch duration, benchmarks/pmc_mfma.py), trusted only when it was collected on THIS build of the library"""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_mfma.json")) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    if rec.get("lib_sha16") != lib_sha():
        return {"note": f"profiles/pmc_mfma.json was collected on build {rec.get('lib_sha16')}, this is {lib_sha()}"}
    wl = rec.get("workloads", {}).get(workload)
    if not wl:
        return None
    top = sorted(wl["kernels"].items(), key=lambda kv: -kv[1]["total_us"])[:6]
    return {"time_weighted_mfma_util": wl["time_weighted_mfma_util"], "source": rec.get("source"),
            "kernels": {k.split("lyc")[-1][:70]: {"avg_us": v["avg_us"], "mfma_util": v["mfma_util"], "launches": v["launches"]} for k, v in top}}


def roofline(insts, args, dtype, dev):
    """Roofline of the dominant kernel family of the step's nn.Linear launches (the Conv2d layers run the same kernels
    through the pixel-row gather), measured live: hipGraphs holding exactly those launches -- forward only, backward only --
    replayed and timed with HIP events on the launch stream.  Every instance reads its own x / g.

    HBM-bound algorithms (LoKr, LoCon, (IA)^3): achieved = algorithmic bytes (fwd x + y, bwd g + x + dx,
    fp32 factors once per pass) / time.  LoHa is MFMA-bound: achieved = useful flops 3 * 2 * M * O * I / time."""
    from lycoris_amd import ops
    algo = args.algo
    lin = [it for it in insts if it.spec["kind"] == "linear" and it.algo == (algo if algo != "mixed" else "lokr")]
    if not lin:
        return None
    esz = torch.empty((), dtype=dtype).element_size()
    core = {"lokr": ops._LokrCore, "locon": ops._LoconCore, "loha": ops._LohaCore}.get(lin[0].algo)
    calls, b_fwd, b_bwd, flops = [], 0, 0, 0
    for it in lin:
        M, I, O = layer_rows(it.spec)
        nfac = 4 * sum(p.numel() for p in it.params)
        if it.algo == "ia3":
            C = it.params[0].numel()
            b_fwd += esz * 2 * M * C + nfac
            b_bwd += esz * 3 * M * C + 2 * nfac
            continue
        fs = [ops._f32c(p) for p in it.params]
        rows = it.x.detach().reshape(-1, I)
        bufs = [torch.zeros_like(f) for f in fs]
        calls.append((it, rows, it.g.reshape(-1, O), fs, bufs))
        # SURVEY 8d per layer: forward x + y, backward g + x + dx, factors once per pass.  A sibling set that goes out as ONE launch
        # reads its shared input once per pass: the x term is counted for the set's first member only (fewer algorithmic bytes)
        own_x = not (_groupable(it) and it.sibs[0] is not it)
        b_fwd += esz * ((M * I if own_x else 0) + M * O) + nfac
        # ... and in the backward the set reads x once and stores ONE summed dx: a non-leading member adds its g only (strict count)
        b_bwd += esz * (M * O + (2 if own_x else 0) * M * I) + 2 * nfac
        flops += 3 * 2 * M * O * I
    saved = {}

    # LoKr: the production launches read pre-packed operand planes (one set per layer, refreshed once per step by grouped
    # pack launches that are part of the timed step but not of this kernel family)
    planes = {}
    use_planes = lin[0].algo == "lokr" and not args.no_planes and dtype != torch.float32
    if use_planes:
        from lycoris_amd import _native as N_
        code_ = N_.dtype_code(dtype)
        for it, rows, g, fs, bufs in calls:
            (a, b), (c, d) = fs[0].shape, fs[1].shape
            nf = int(N_.load().lyc_lokr_planes_bytes(c, d, 1, 0))
            pl = torch.empty(nf + int(N_.load().lyc_lokr_planes_bytes(c, d, 1, 1)), dtype=torch.uint8, device=dev)
            N_.call("lyc_lokr_pack_w2", N_.ptr(fs[1]), d, 1, 0, None, 0, 0, None, 0, 0, 0, 0, c, d, 1, N_.ptr(pl), N_.ptr(pl[nf:]), code_,
                    N_.stream_ptr(dev))
            planes[id(it)] = (pl, pl[nf:])

    # launch units of the LoKr step: a layer, or a sibling set as one grouped launch (what forward_all / the modules issue)
    units, k_of = [], {id(cl[0]): k for k, cl in enumerate(calls)}
    if use_planes:
        import ctypes as _ct
        ys_pre = {}
        for k, (it, rows, g, fs, bufs) in enumerate(calls):
            if _groupable(it) and all(id(m) in k_of for m in it.sibs):
                if it.sibs[0] is it:
                    units.append([k_of[id(m)] for m in it.sibs])
            else:
                units.append([k])
        for u in units:
            if len(u) > 1:
                for k in u:
                    it, rows, g, fs, bufs = calls[k]
                    ys_pre[k] = torch.empty(rows.shape[0], fs[0].shape[0] * fs[1].shape[0], dtype=dtype, device=dev)
        fw_items = {}
        for u in units:
            if len(u) > 1:
                arr = (N_.LinearGroupItem * len(u))()
                for j, k in enumerate(u):
                    it, rows, g, fs, bufs = calls[k]
                    arr[j] = N_.LinearGroupItem(N_.ptr(rows), N_.ptr(fs[0]), N_.ptr(planes[id(it)][0]), None, N_.ptr(ys_pre[k]), None, rows.shape[0], 1.0)
                fw_items[u[0]] = arr
        _KEEP.extend([ys_pre, fw_items])

    def fwd():
        if core is None:  # ia3
            for it in lin:
                saved[id(it)] = it.forward()
            return
        if use_planes:
            for u in units:
                it, rows, g, fs, bufs = calls[u[0]]
                (a, b), (c, d) = fs[0].shape, fs[1].shape
                if len(u) > 1:
                    N_.call("lyc_lokr_linear_fwd_group", _ct.cast(fw_items[u[0]], _ct.c_void_p), len(u), a, b, c, d, code_, N_.stream_ptr(dev))
                    for k in u:
                        saved[id(calls[k][0])] = (ys_pre[k], ())
                else:
                    y = torch.empty(rows.shape[0], a * c, dtype=dtype, device=dev)
                    N_.call("lyc_lokr_linear_fwd_planes", N_.ptr(rows), N_.ptr(fs[0]), N_.ptr(planes[id(it)][0]), None, N_.ptr(y), rows.shape[0],
                            a, b, c, d, 1.0, code_, N_.stream_ptr(dev))
                    saved[id(it)] = (y, ())
            return
        for it, rows, g, fs, bufs in calls:
            if use_planes:
                (a, b), (c, d) = fs[0].shape, fs[1].shape
                y = torch.empty(rows.shape[0], a * c, dtype=dtype, device=dev)
                N_.call("lyc_lokr_linear_fwd_planes", N_.ptr(rows), N_.ptr(fs[0]), N_.ptr(planes[id(it)][0]), None, N_.ptr(y), rows.shape[0],
                        a, b, c, d, 1.0, code_, N_.stream_ptr(dev))
                saved[id(it)] = (y, ())
            else:
                saved[id(it)] = core.fwd(rows, fs, 1.0)

    def bwd():
        if core is None:
            for it in lin:
                torch.autograd.grad(saved[id(it)], [it.x] + it.params, it.g, retain_graph=True, allow_unused=True)  # dw goes into the arena
            return
        for it, rows, g, fs, bufs in calls:
            core.bwd(g, rows, fs, saved[id(it)][1], 1.0, True, [True] * len(fs), False, bufs)

    t_fwd = _graph_ms(fwd)
    t_bwd = _graph_ms(bwd)
    t_ms = t_fwd + t_bwd
    n_l = len(lin)
    launches = {"lokr": 3, "locon": 3, "loha": 7, "ia3": 3}[lin[0].algo]  # kernel launches per layer, fwd + bwd
    out = {"families_ms": {"forward": round(t_fwd, 3), "backward": round(t_bwd, 3)}, "layers": n_l,
           "launches_per_layer": launches, "avg_launch_us": round(t_ms * 1e3 / (launches * n_l), 2)}
    if lin[0].algo == "loha":
        # production backward: dx = g dW per layer + G = g^T x and HadaWeight.backward of ALL layers in one lyc_loha_wgrad_group
        # call (library GEMM per layer, grouped factor-gradient launches)
        import ctypes
        from lycoris_amd import _native as N
        code = N.dtype_code(dtype)
        items = (N.LohaWgradItem * len(calls))()
        gws, dxs = [], []
        for k, (it, rows, g, fs, bufs) in enumerate(calls):
            O, r = fs[0].shape
            I = fs[1].shape[1]
            gws.append(torch.empty(O, I, device=dev))
            dxs.append(torch.empty_like(rows))
            items[k] = N.LohaWgradItem(N.ptr(g), N.ptr(rows), *[N.ptr(f) for f in fs], *[N.ptr(b) for b in bufs], N.ptr(gws[k]),
                                       rows.shape[0], I, O, r, 1.0)
        _KEEP.extend([items, gws, dxs])

        def only_dx():
            for k, (it, rows, g, fs, bufs) in enumerate(calls):
                O, r = fs[0].shape
                N.call("lyc_loha_linear_bwd", N.ptr(g), N.ptr(rows), *[N.ptr(f) for f in fs], N.ptr(saved[id(it)][1][0]), None, N.ptr(dxs[k]),
                       None, None, None, None, rows.shape[0], fs[1].shape[1], O, r, 1.0, code, N.stream_ptr(dev))

        def grouped_wgrad():
            N.call("lyc_loha_wgrad_group", ctypes.cast(items, ctypes.c_void_p), len(calls), code, N.stream_ptr(dev))

        t_dx = _graph_ms(only_dx)
        t_wg = _graph_ms(grouped_wgrad)
        out["families_ms"] = {"forward": round(t_fwd, 3), "backward_dx": round(t_dx, 3), "G_gemm_and_grouped_factor_gradients": round(t_wg, 3),
                              "backward_one_call_per_layer": round(t_bwd, 3)}
        t_ms = t_fwd + t_dx + t_wg
        ach = flops / (t_ms * 1e-3) / 1e12
        tr, src = pmc_traffic("loha_linear", f"loha/{args.model}/linear")
        out.update({"bound": "mfma", "kernel": "LoHa dense contractions y = x dW^T, dx = g dW, G = g^T x of the Linear layers "
                                              "(+ dW rebuild and Hadamard chain rule)",
                    "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                    # HBM bytes of the family (gemm16 + rebuild + factor gradients) over one pass, per layer
                    "traffic": int(tr["bytes_per_pass"] / n_l) if tr else None, "traffic_source": src,
                    "algorithmic_flops_per_layer": int(flops / n_l)})
        return out
    fam = {"lokr": "lokr_kron4", "locon": "locon_linear", "ia3": "ia3"}[lin[0].algo]
    nbytes = b_fwd + b_bwd
    hot = nbytes / (t_ms * 1e-3) / 1e9
    workload = f"{lin[0].algo if algo != 'mixed' else 'mixed'}/{args.model}/linear"
    tr, src = pmc_traffic(fam, workload)
    traffic = int(tr["bytes_per_launch"]) if tr else None
    out.update({"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic, "traffic_source": src, "traffic_workload": workload,
                "hot_path_gbs": round(hot, 1), "hot_path_frac": round(hot / HBM_PEAK_GBS, 4),
                "forward_gbs": round(b_fwd / (t_fwd * 1e-3) / 1e9, 1), "backward_gbs": round(b_bwd / (t_bwd * 1e-3) / 1e9, 1)})
    if lin[0].algo == "lokr":
        # the dominant KERNEL: lyc::kron3_kernel runs the forward and the backward dx (+ dW1 partials) launch of every layer;
        # the weight gradients (dW2, dW1 reduction) of ALL layers run in grouped launches (lyc::kron_dw2s_group_kernel,
        # lyc_lokr_wgrad_group: 24 layers per launch), which re-read g and x.  Three graphs, no subtraction:
        #   forward | backward dx with LYC_DEFER_WGRAD | one lyc_lokr_wgrad_group call over all layers
        import ctypes
        from lycoris_amd import _native as N
        code = N.dtype_code(dtype)
        items = (N.WgradItem * len(calls))()
        wss, dxs = [], []
        for k, (it, rows, g, fs, bufs) in enumerate(calls):
            (a, b), (c, d) = fs[0].shape, fs[1].shape
            nb = max(int(N.load().lyc_lokr_bwd_workspace_bytes(rows.shape[0], a, b, c, d, code)), 16)
            wss.append(torch.empty(nb, dtype=torch.uint8, device=dev))
            dxs.append(torch.empty_like(rows))
            assert N.load().lyc_lokr_wgrad_deferrable(N.ptr(g), N.ptr(rows), rows.shape[0], a, b, c, d, code) == 1
            items[k] = N.WgradItem(N.ptr(g), N.ptr(rows), N.ptr(fs[0]), N.ptr(bufs[0]), N.ptr(bufs[1]), N.ptr(wss[k]),
                                   rows.shape[0], a, b, c, d, 1.0)
        _KEEP.extend([items, wss, dxs])

        bw_items = {}
        if use_planes:
            for u in units:
                if len(u) > 1:
                    arr = (N.LinearGroupItem * len(u))()
                    for j, k in enumerate(u):
                        it, rows, g, fs, bufs = calls[k]
                        arr[j] = N.LinearGroupItem(N.ptr(g), N.ptr(fs[0]), N.ptr(planes[id(it)][1]), N.ptr(rows), N.ptr(dxs[k]), N.ptr(wss[k]), rows.shape[0], 1.0)
                    bw_items[u[0]] = arr
            _KEEP.append(bw_items)

        def only_dx():
            if use_planes:
                for u in units:
                    it, rows, g, fs, bufs = calls[u[0]]
                    (a, b), (c, d) = fs[0].shape, fs[1].shape
                    if len(u) > 1:  # the set's dx as ONE launch that stores the sum of the n results (the shared input's gradient)
                        N.call("lyc_lokr_linear_bwd_group_sum", ctypes.cast(bw_items[u[0]], ctypes.c_void_p), len(u), a, b, c, d, N.ptr(dxs[u[0]]),
                               code, N.stream_ptr(dev))
                    else:
                        k = u[0]
                        N.call("lyc_lokr_linear_bwd_planes", N.ptr(g), N.ptr(rows), N.ptr(fs[0]), N.ptr(planes[id(it)][1]), N.ptr(dxs[k]),
                               N.ptr(bufs[0]), None, N.ptr(wss[k]), rows.shape[0], a, b, c, d, 1.0, code | 0x200, N.stream_ptr(dev))
                return
            for k, (it, rows, g, fs, bufs) in enumerate(calls):
                (a, b), (c, d) = fs[0].shape, fs[1].shape
                if use_planes:
                    N.call("lyc_lokr_linear_bwd_planes", N.ptr(g), N.ptr(rows), N.ptr(fs[0]), N.ptr(planes[id(it)][1]), N.ptr(dxs[k]),
                           N.ptr(bufs[0]), None, N.ptr(wss[k]), rows.shape[0], a, b, c, d, 1.0, code | 0x200, N.stream_ptr(dev))
                else:
                    N.call("lyc_lokr_linear_bwd", N.ptr(g), N.ptr(rows), N.ptr(fs[0]), N.ptr(fs[1]), N.ptr(dxs[k]), N.ptr(bufs[0]),
                           None, N.ptr(wss[k]), rows.shape[0], a, b, c, d, 1.0, code | 0x200, N.stream_ptr(dev))

        tb = int(N.load().lyc_lokr_wgrad_table_bytes(len(calls)))
        wg_table = torch.empty(tb, dtype=torch.uint8, device=dev)
        _KEEP.append(wg_table)

        def grouped_wgrad():  # as csrc/torch_ops.cpp flushes them: problem table in device scratch, one launch per tile class
            N.call("lyc_lokr_wgrad_group_ws", ctypes.cast(items, ctypes.c_void_p), len(calls), code, N.ptr(wg_table), tb, N.stream_ptr(dev))

        def grouped_wgrad_24():  # without the scratch: 24 layers per launch (kernel arguments)
            N.call("lyc_lokr_wgrad_group", ctypes.cast(items, ctypes.c_void_p), len(calls), code, N.stream_ptr(dev))

        def grouped_wgrad_tile_s():  # the round 1-3 tile plan (LYC_WGRAD_TILE_S), A/B leg
            N.call("lyc_lokr_wgrad_group", ctypes.cast(items, ctypes.c_void_p), len(calls), code | 0x400, N.stream_ptr(dev))

        def per_layer_dw2():
            for it, rows, g, fs, bufs in calls:
                (a, b), (c, d) = fs[0].shape, fs[1].shape
                N.call("lyc_lokr_linear_bwd", N.ptr(g), N.ptr(rows), N.ptr(fs[0]), N.ptr(fs[1]), None, None, N.ptr(bufs[1]), None,
                       rows.shape[0], a, b, c, d, 1.0, code, N.stream_ptr(dev))

        t_dx = _graph_ms(only_dx)
        t_wg = _graph_ms(grouped_wgrad)
        t_wg_24 = _graph_ms(grouped_wgrad_24)
        t_wg_s = _graph_ms(grouped_wgrad_tile_s)
        t_dw2 = _graph_ms(per_layer_dw2)
        b_dw2 = sum(esz * (it.spec["M"] * (it.spec["I"] + it.spec["O"])) + 4 * sum(p.numel() for p in it.params) for it in lin)
        b_dx = b_bwd  # g + x (dW1) + dx + factors: the SURVEY 8d backward bytes belong to this launch
        k3_ms, k3_bytes = t_fwd + t_dx, b_fwd + b_dx
        ach = k3_bytes / (k3_ms * 1e-3) / 1e9
        hot = nbytes / ((t_fwd + t_dx + t_wg) * 1e-3) / 1e9
        out["families_ms"] = {"kron_forward": round(t_fwd, 3), "kron_backward_dx_dw1": round(t_dx, 3),
                              "kron_dw2_grouped": round(t_wg, 3), "kron_dw2_grouped_24_per_launch": round(t_wg_24, 3),
                              "kron_dw2_grouped_narrow_tiles_r3": round(t_wg_s, 3),
                              "kron_dw2s_one_launch_per_layer": round(t_dw2, 3),
                              "backward_one_call_per_layer": round(t_bwd, 3)}
        out.update({"operand_planes": bool(use_planes),
                    "kernel": ("lyc::kron4_kernel / lyc::kron4_group_kernel / lyc::kron4_sum_kernel (one body)" if use_planes else "lyc::kron3_kernel") + " (LoKr forward + backward dx / dW1 launches of the "
                              "Linear layers" + (", both operands by LDS-DMA, w2 from pre-packed hi/lo planes" if use_planes else "") + "); the weight "
                              "gradients run grouped (lyc::kron_dw2f_table_kernel: full-width output tiles, one launch per tile class, re-reads g and x): "
                              "families_ms",
                    "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
                    "launches": 2 * (len(units) if use_planes else n_l),
                    "launches_note": (f"{sum(1 for u in units if len(u) > 1)} sibling sets ({sum(len(u) for u in units if len(u) > 1)} layers) as one "
                                      f"lyc::kron4_group_kernel launch each, {sum(1 for u in units if len(u) == 1)} layers on their own; forward + "
                                      "backward dx (a set's dx launch stores the SUM of its n results: lyc::kron4_sum_kernel)") if use_planes else "one per layer and pass",
                    "avg_launch_us": round(k3_ms * 1e3 / (2 * (len(units) if use_planes else n_l)), 2),
                    "algorithmic_bytes_per_launch": int(k3_bytes / (2 * (len(units) if use_planes else n_l))),
                    "hot_path_gbs": round(hot, 1), "hot_path_frac": round(hot / HBM_PEAK_GBS, 4),
                    "backward_gbs": round(b_bwd / ((t_dx + t_wg) * 1e-3) / 1e9, 1),
                    "dw2_grouped_gbs": round(b_dw2 / (t_wg * 1e-3) / 1e9, 1),
                    "dw2s_per_layer_gbs": round(b_dw2 / (t_dw2 * 1e-3) / 1e9, 1)})
        if tr:  # HBM bytes of the family over one pass (every launch of it, the sets' one-pass dx sums included) per forward / dx launch
            out["traffic"] = int(tr["bytes_per_pass"] / out["launches"])
            out["traffic_over_algorithmic"] = round(tr["bytes_per_pass"] / k3_bytes, 3)
        # per shape: forward and dx launches of each distinct Linear shape in their own graphs (<= 12 instances,
        # every instance its own x / g), HIP events.  mfma_model = matrix-core busy fraction from the launch's instruction count
        # (stage 1: 2 v_mfma_16x16x32 per 16x16x32 block, stage 2: 3 v_mfma_16x16x16 per 16x16 tile, + 4 with dW1; 16 cycles each)
        # over 1024 SIMDs at 2.1 GHz -- the measured counters are in profiles/ (r04_pmc_*).
        shapes = {}
        for k, cl in enumerate(calls):
            key = layer_rows(cl[0].spec)
            shapes.setdefault(key, []).append(k)
        table = []
        for (M, I, O), idx in sorted(shapes.items(), key=lambda kv: -len(kv[1])):
            idx = idx[:12]
            sub = [calls[k] for k in idx]

            def f_sub():
                for it, rows, g, fs, bufs in sub:
                    (a, b), (c, d) = fs[0].shape, fs[1].shape
                    if use_planes:
                        y = torch.empty(rows.shape[0], a * c, dtype=dtype, device=dev)
                        N.call("lyc_lokr_linear_fwd_planes", N.ptr(rows), N.ptr(fs[0]), N.ptr(planes[id(it)][0]), None, N.ptr(y), rows.shape[0],
                               a, b, c, d, 1.0, code, N.stream_ptr(dev))
                    else:
                        core.fwd(rows, fs, 1.0)

            def b_sub():
                for k in idx:
                    it, rows, g, fs, bufs = calls[k]
                    (a, b), (c, d) = fs[0].shape, fs[1].shape
                    if use_planes:
                        N.call("lyc_lokr_linear_bwd_planes", N.ptr(g), N.ptr(rows), N.ptr(fs[0]), N.ptr(planes[id(it)][1]), N.ptr(dxs[k]),
                               N.ptr(bufs[0]), None, N.ptr(wss[k]), rows.shape[0], a, b, c, d, 1.0, code | 0x200, N.stream_ptr(dev))
                    else:
                        N.call("lyc_lokr_linear_bwd", N.ptr(g), N.ptr(rows), N.ptr(fs[0]), N.ptr(fs[1]), N.ptr(dxs[k]), N.ptr(bufs[0]),
                               None, N.ptr(wss[k]), rows.shape[0], a, b, c, d, 1.0, code | 0x200, N.stream_ptr(dev))

            tf = _graph_ms(f_sub) * 1e3 / len(idx)
            tb = _graph_ms(b_sub) * 1e3 / len(idx)
            bf_, bb_ = esz * M * (I + O), esz * M * (O + 2 * I)
            rows8, Kf, Nf = M * FACTOR, I // FACTOR, O // FACTOR

            def mfma_frac(K, Nn, us, dw1):
                tiles = -(-rows8 // 16) * -(-Nn // 16)
                n = tiles * (2 * -(-K // 32) + 3 + (4 if dw1 else 0))
                return n * 16 / (1024 * us * 1e-6 * 2.1e9)
            table.append({"M": M, "I": I, "O": O, "count": len(shapes[(M, I, O)]),
                          "fwd_us": round(tf, 2), "fwd_gbs": round(bf_ / tf * 1e-3, 1), "fwd_frac": round(bf_ / tf * 1e-3 / HBM_PEAK_GBS, 4),
                          "fwd_mfma_model": round(mfma_frac(Kf, Nf, tf, False), 3),
                          "bwd_us": round(tb, 2), "bwd_gbs": round(bb_ / tb * 1e-3, 1), "bwd_frac": round(bb_ / tb * 1e-3 / HBM_PEAK_GBS, 4),
                          "bwd_mfma_model": round(mfma_frac(Nf, Kf, tb, True), 3)})
        out["per_shape"] = table
        # Sibling projections in one launch (lyc_lokr_linear_fwd_group / _bwd_group): the to_q / to_k / to_v layers of an
        # attention block (3 of every 6 dim -> dim layers) and to_k / to_v of its cross-attention (the M = 77 layers, in pairs) share
        # their input in the real UNet and are independent of each other.  Measured here as an A/B over the SAME layer instances:
        # per-layer launches (what the module API and the headline step do) against one grouped launch per sibling set.
        if use_planes and args.model == "sdxl":
            sib = {}
            for (M, I, O), gsz, tag in (((1024, 1280, 1280), 3, "attn1_qkv_1280"), ((77, 2048, 1280), 2, "attn2_kv_1280"),
                                        ((4096, 640, 640), 3, "attn1_qkv_640"), ((77, 2048, 640), 2, "attn2_kv_640")):
                idx = shapes.get((M, I, O), [])
                blocks = {(1024, 1280, 1280): 60, (77, 2048, 1280): 60, (4096, 640, 640): 10, (77, 2048, 640): 10}[(M, I, O)]
                nset = min(len(idx) // gsz, 20)
                if nset < 1:
                    continue
                sets = [idx[i * gsz:(i + 1) * gsz] for i in range(nset)]
                a = FACTOR
                c, d = O // a, I // a
                ys = {k: torch.empty(M, O, dtype=dtype, device=dev) for st_ in sets for k in st_}
                fw_items, bw_items = [], []
                for st_ in sets:
                    fw = (N.LinearGroupItem * gsz)()
                    bw = (N.LinearGroupItem * gsz)()
                    for j, k in enumerate(st_):
                        it, rows, g, fs, bufs = calls[k]
                        fw[j] = N.LinearGroupItem(N.ptr(rows), N.ptr(fs[0]), N.ptr(planes[id(it)][0]), None, N.ptr(ys[k]), None, M, 1.0)
                        bw[j] = N.LinearGroupItem(N.ptr(g), N.ptr(fs[0]), N.ptr(planes[id(it)][1]), N.ptr(rows), N.ptr(dxs[k]), N.ptr(wss[k]), M, 1.0)
                    fw_items.append(fw)
                    bw_items.append(bw)
                _KEEP.extend([ys, fw_items, bw_items])

                def f_each():
                    for st_ in sets:
                        for k in st_:
                            it, rows, g, fs, bufs = calls[k]
                            N.call("lyc_lokr_linear_fwd_planes", N.ptr(rows), N.ptr(fs[0]), N.ptr(planes[id(it)][0]), None, N.ptr(ys[k]), M, a, a, c, d,
                                   1.0, code, N.stream_ptr(dev))

                def f_group():
                    for fw in fw_items:
                        N.call("lyc_lokr_linear_fwd_group", ctypes.cast(fw, ctypes.c_void_p), gsz, a, a, c, d, code, N.stream_ptr(dev))

                def b_each():
                    for st_ in sets:
                        for k in st_:
                            it, rows, g, fs, bufs = calls[k]
                            N.call("lyc_lokr_linear_bwd_planes", N.ptr(g), N.ptr(rows), N.ptr(fs[0]), N.ptr(planes[id(it)][1]), N.ptr(dxs[k]),
                                   N.ptr(bufs[0]), None, N.ptr(wss[k]), M, a, a, c, d, 1.0, code | 0x200, N.stream_ptr(dev))

                def b_group():
                    for bw in bw_items:
                        N.call("lyc_lokr_linear_bwd_group", ctypes.cast(bw, ctypes.c_void_p), gsz, a, a, c, d, code, N.stream_ptr(dev))

                t = {nm: _graph_ms(fn) * 1e3 / nset for nm, fn in (("fwd_per_layer", f_each), ("fwd_grouped", f_group), ("dx_per_layer", b_each),
                                                                    ("dx_grouped", b_group))}
                sib[tag] = {"layers_per_group": gsz, "groups_in_unet": blocks, **{k_: round(v, 2) for k_, v in t.items()},
                            "saving_ms_per_step": round(blocks * (t["fwd_per_layer"] - t["fwd_grouped"] + t["dx_per_layer"] - t["dx_grouped"]) * 1e-3, 3)}
            if sib:
                sib["unit"] = "us per sibling group (hipGraph, HIP events)"
                sib["saving_ms_per_step_total"] = round(sum(v["saving_ms_per_step"] for v in sib.values() if isinstance(v, dict)), 3)
                sib["note"] = ("A/B over the same layer instances.  Since round 5 the grouped form IS what `value` runs: the modules find their "
                               "sibling sets behind the reference API (lycoris_amd/modules/siblings.py) and bench.py issues the same op")
                out["sibling_groups"] = sib
        conv = conv_leg(insts, args, dtype)
        if conv:
            out["conv"] = conv
        return out
    if lin[0].algo == "locon":
        # production backward: the fused dx launch per layer (lyc::bneck4_kernel, also writes dt) + the factor gradients of ALL
        # layers in grouped launches (lyc::lowrank_tn_group_kernel, lyc_locon_wgrad_group: 18 layers per launch)
        import ctypes
        from lycoris_amd import _native as N
        code = N.dtype_code(dtype)
        items = (N.LoconWgradItem * len(calls))()
        dts, dxs, ts_pre, ys_pre = [], [], [], []
        for k, (it, rows, g, fs, bufs) in enumerate(calls):
            r, I = fs[0].shape
            O = fs[1].shape[0]
            dts.append(torch.empty(rows.shape[0], r, device=dev))
            dxs.append(torch.empty_like(rows))
            ts_pre.append(torch.empty(rows.shape[0], r, device=dev))
            ys_pre.append(torch.empty(rows.shape[0], O, dtype=dtype, device=dev))
            assert N.load().lyc_locon_wgrad_deferrable(N.ptr(g), N.ptr(rows), rows.shape[0], I, O, r, code) == 1
            items[k] = N.LoconWgradItem(N.ptr(g), N.ptr(rows), N.ptr(ts_pre[k]), N.ptr(dts[k]), N.ptr(bufs[0]), N.ptr(bufs[1]),
                                        rows.shape[0], I, O, r, 1.0)
        # launch units of the step (round 5): a layer, or a sibling set as ONE lyc::bneck4_group_kernel / bneck4_sum_kernel launch each way
        k_of = {id(cl[0]): k for k, cl in enumerate(calls)}
        units = []
        for k, (it, rows, g, fs, bufs) in enumerate(calls):
            if _groupable(it) and all(id(m) in k_of for m in it.sibs):
                if it.sibs[0] is it:
                    units.append([k_of[id(m)] for m in it.sibs])
            else:
                units.append([k])
        fw_items, bw_items = {}, {}
        for u in units:
            if len(u) > 1:
                fa, ba = (N.LoconGroupItem * len(u))(), (N.LoconGroupItem * len(u))()
                for j, k in enumerate(u):
                    it, rows, g, fs, bufs = calls[k]
                    fa[j] = N.LoconGroupItem(N.ptr(rows), N.ptr(fs[0]), N.ptr(fs[1]), N.ptr(ts_pre[k]), N.ptr(ys_pre[k]), rows.shape[0], 1.0)
                    ba[j] = N.LoconGroupItem(N.ptr(g), N.ptr(fs[0]), N.ptr(fs[1]), N.ptr(dts[k]), N.ptr(dxs[k]), rows.shape[0], 1.0)
                fw_items[u[0]], bw_items[u[0]] = fa, ba
        _KEEP.extend([items, dts, dxs, ts_pre, ys_pre, fw_items, bw_items])

        def fwd_units():
            for u in units:
                it, rows, g, fs, bufs = calls[u[0]]
                r, I = fs[0].shape
                O = fs[1].shape[0]
                if len(u) > 1:
                    N.call("lyc_locon_linear_fwd_group", ctypes.cast(fw_items[u[0]], ctypes.c_void_p), len(u), I, O, r, code, N.stream_ptr(dev))
                else:
                    k = u[0]
                    N.call("lyc_locon_linear_fwd", N.ptr(rows), N.ptr(fs[0]), N.ptr(fs[1]), N.ptr(ts_pre[k]), N.ptr(ys_pre[k]), rows.shape[0], I, O, r,
                           1.0, code, N.stream_ptr(dev))

        def only_dx():
            for u in units:
                it, rows, g, fs, bufs = calls[u[0]]
                r, I = fs[0].shape
                O = fs[1].shape[0]
                if len(u) > 1:  # the set's dx as ONE expand stage over its n mid tiles (round 6: lyc::bneck4_sum_kernel) where the plan covers
                    # it; else the n dx launches as one + the one-pass sum (what torch_ops.cpp LoconLinearGroupFn::backward does)
                    rc = N.load().lyc_locon_linear_bwd_group_sum(ctypes.cast(bw_items[u[0]], ctypes.c_void_p), len(u), I, O, r, N.ptr(dxs[u[0]]), code,
                                                                 N.stream_ptr(dev))
                    if rc == 2:  # LYC_ERR_UNSUPPORTED: nothing launched
                        N.call("lyc_locon_linear_bwd_group", ctypes.cast(bw_items[u[0]], ctypes.c_void_p), len(u), I, O, r, code, N.stream_ptr(dev))
                        srcs = (ctypes.c_void_p * len(u))(*[dxs[k].data_ptr() for k in u])
                        N.call("lyc_sum_rows", ctypes.cast(srcs, ctypes.c_void_p), len(u), N.ptr(dxs[u[0]]), dxs[u[0]].numel(), code, N.stream_ptr(dev))
                    else:
                        assert rc == 0, N.load().lyc_last_error()
                else:
                    k = u[0]
                    N.call("lyc_locon_linear_bwd", N.ptr(g), N.ptr(rows), N.ptr(fs[0]), N.ptr(fs[1]), N.ptr(ts_pre[k]), N.ptr(dts[k]),
                           N.ptr(dxs[k]), None, None, rows.shape[0], I, O, r, 1.0, code, N.stream_ptr(dev))

        t_fwd = _graph_ms(fwd_units)

        def grouped_wgrad():
            N.call("lyc_locon_wgrad_group", ctypes.cast(items, ctypes.c_void_p), len(calls), code, N.stream_ptr(dev))

        t_dx = _graph_ms(only_dx)
        t_wg = _graph_ms(grouped_wgrad)
        hot = nbytes / ((t_fwd + t_dx + t_wg) * 1e-3) / 1e9
        b2 = b_fwd + b_bwd  # the two bneck launches of a layer read x / g and write y / dx: the SURVEY 8d bytes of the layer
        k_ms = t_fwd + t_dx
        ach = b2 / (k_ms * 1e-3) / 1e9
        out["families_ms"] = {"bneck_forward": round(t_fwd, 3), "bneck_backward_dx": round(t_dx, 3), "lowrank_tn_grouped": round(t_wg, 3),
                              "backward_one_call_per_layer": round(t_bwd, 3)}
        n_launch = 2 * len(units)
        out.update({"kernel": "lyc::bneck4_kernel / lyc::bneck4_group_kernel / lyc::bneck4_sum_kernel (LoCon forward + backward-dx launches of the Linear "
                              "layers, every operand by LDS-DMA into wave-private rings; a sibling set = one launch each way, its dx the SUM of the n "
                              "results formed in registers); the factor gradients run grouped "
                              "(lyc::lowrank_tn_group_kernel, 18 layers per launch, re-reads g and x): families_ms",
                    "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "launches": n_launch,
                    "launches_note": f"{sum(1 for u in units if len(u) > 1)} sibling sets ({sum(len(u) for u in units if len(u) > 1)} layers) as one launch "
                                     f"each, {sum(1 for u in units if len(u) == 1)} layers on their own",
                    "avg_launch_us": round(k_ms * 1e3 / n_launch, 2), "algorithmic_bytes_per_launch": int(b2 / n_launch),
                    "hot_path_gbs": round(hot, 1), "hot_path_frac": round(hot / HBM_PEAK_GBS, 4),
                    "backward_gbs": round(b_bwd / ((t_dx + t_wg) * 1e-3) / 1e9, 1)})
        return out
    out.update({"kernel": "lyc::chan_scale_kernel / lyc::chan_reduce_kernel",
                "achieved": round(hot, 1), "frac": round(hot / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_launch": int(nbytes / (launches * n_l))})
    return out


def conv_leg(insts, args, dtype):
    """roofline.conv: the Conv2d layers of the LoKr step as their own kernel family (the family furthest below
    its roofline had no leg and no counters).  Algorithmic bytes per layer e * (3 |x| + 2 |y|) (forward: x, y; backward: g, x, dx);
    time = forward graph + backward graph (dx launches + the grouped weight-gradient launches), op level, HIP events."""
    conv = [it for it in insts if it.spec["kind"] == "conv" and it.algo == "lokr"]
    if not conv:
        return None
    esz = torch.empty((), dtype=dtype).element_size()
    nbytes = sum(esz * (3 * it.x.numel() + 2 * it.g.numel()) for it in conv)
    held = {}

    def fwd():
        held["o"] = forward_all(conv)

    def both():
        outs = forward_all(conv)
        backward_range(outs, 0, len(outs))

    t_f = _graph_ms(fwd)
    t_fb = _graph_ms(both)
    t_b = max(t_fb - t_f, 0.0)
    ach = nbytes / (t_fb * 1e-3) / 1e9
    workload = f"lokr/{args.model}/conv"
    tr, src = pmc_traffic("lokr_conv", workload)
    out = {"layers": len(conv), "kernel": "lyc::kconv_kernel (LDS source patch + packed operand planes; forward, backward dx / dW1) + the "
                                           "grouped weight-gradient launches (lyc::kron_dw2f_group_kernel, Conv2d form); round 6: 8 waves + software-pipelined k loop; 1x1 convs run "
                                           "the Linear kernels (lyc::kron4_kernel, grouped weight gradients)",
           "families_ms": {"forward": round(t_f, 3), "backward_dx_and_grouped_dw2": round(t_b, 3)},
           "algorithmic_bytes_per_step": int(nbytes), "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(ach / HBM_PEAK_GBS, 4), "memory_format": "channels_last" if args.channels_last else "contiguous (NCHW: + two row transposes per pass)",
           "traffic": int(tr["bytes_per_pass"]) if tr else None, "traffic_source": src, "traffic_workload": workload}
    if tr:
        out["traffic_over_algorithmic"] = round(tr["bytes_per_pass"] / nbytes, 2)
    mb = pmc_mfma(workload)  # the family is matrix-core / latency bound rather than HBM bound: its MFMA-busy share beside the HBM figure
    if mb:
        out["mfma_busy"] = {k: v for k, v in mb.items() if k != "kernels"} if "time_weighted_mfma_util" in mb else mb
    return out


def reference_leg(insts, sync, native_graph_ms):
    """The north-star comparator: the reference's rebuild call sequence (Inst.reference_forward) through stock
    PyTorch-ROCm on this GPU over the same layer instances, like for like: eager vs eager (Python-driven, what an
    sd-scripts user gets without whole-step capture) and hipGraph vs hipGraph (kernel time)."""
    def ref_pass():
        outs = [(it.reference_forward(), it) for it in insts][::-1]
        torch.autograd.grad([y for y, _ in outs], _wrt([it for _, it in outs]), [it.g for _, it in outs])

    def nat_pass():
        backward_range(forward_all(insts), 0, len(insts))

    sync._sync_enabled = False
    try:
        ref_eager = _wall_ms(ref_pass)
        nat_eager = _wall_ms(nat_pass)
        ref_graph = _graph_ms(ref_pass, reps=2)
        nat_graph = _graph_ms(nat_pass, reps=2)
    finally:
        sync._sync_enabled = True
        for it in insts:
            it.W = None
        torch.cuda.empty_cache()
    return {"what": "adapter fwd+bwd of all layers, reference torch call sequence (kron / matmul -> W + dW - W -> "
                    "F.linear / F.conv2d -> autograd) vs native ops, same tensors, same GPU",
            "reference_eager_ms": round(ref_eager, 2), "native_eager_ms": round(nat_eager, 2),
            "reference_graph_ms": round(ref_graph, 2), "native_graph_ms": round(nat_graph, 2),
            "speedup_eager_vs_eager": round(ref_eager / nat_eager, 2),
            "speedup_graph_vs_graph": round(ref_graph / nat_graph, 2),
            # the north-star sentence read literally: the reference as its users run it (eager) against this path as bench.py runs it
            "speedup_native_graph_vs_reference_eager": round(ref_eager / nat_graph, 2)}


def per_algo_legs():
    """BASELINE.json's other single-GPU configurations, driver-run: LoCon SDXL, LoCon SD1.5 bs 4 (configs[1]), LoHa SDXL (configs[2]),
    (IA)^3, mixed preset fp16 (configs[4]) -- each a short run of this script in a child process (its own 6 GB of activations),
    reduced to ms / step and the roofline of its dominant kernel family."""
    import subprocess
    legs = [("locon_sdxl", ["--algo", "locon"]), ("locon_sd15_bs4", ["--algo", "locon", "--model", "sd15"]),
            ("loha_sdxl", ["--algo", "loha"]), ("ia3_sdxl", ["--algo", "ia3"]), ("mixed_sdxl_fp16", ["--algo", "mixed", "--dtype", "fp16"]),
            # sd-scripts' mixed_precision=bf16: fp32 LayerNorm outputs into to_q / to_k / to_v and the GEGLU projection, torch.autocast(bf16)
            ("lokr_sdxl_autocast", ["--algo", "lokr", "--autocast", "--no-roofline"])]
    out = {}
    for name, extra in legs:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-reference",
               "--no-base", "--no-per-algo"] + extra
        try:
            t0 = time.perf_counter()
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = [ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")][-1]
            d = json.loads(line)
            rl = d.get("roofline") or {}
            out[name] = {"ms_per_step": d["ms_per_step"], "steps_per_s": d["value"], "dtype": d["dtype"], "layers": d["config"]["layers"],
                         "roofline": {k: rl.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
                                                                "families_ms") if k in rl},
                         "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:  # a failing leg must not take the headline line down
            out[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def base_leg(insts, sync, opt=None, ops_=None):
    """the step with the frozen layers' own ops in it (rocBLAS / MIOpen forward + dx-only backward), next to
    the adapter-only number: base alone, and base + adapter (out = base + delta, one autograd.grad for both)."""
    def one_backward(outs, factors=True):  # ONE engine invocation for all layers, as loss.backward() is one for a real network
        outs = outs[::-1]
        torch.autograd.grad([y for y, _ in outs], _wrt([it for _, it in outs], factors), [it.g for _, it in outs], allow_unused=True)

    def base_pass():
        one_backward([(it.base_forward(), it) for it in insts], factors=False)

    def both_pass():
        one_backward(forward_all(insts, with_base=True))

    side = torch.cuda.Stream()

    def both_overlap_pass():
        """the same work with the adapter of every layer on a SIDE stream (lycoris_amd.stream_overlap): forked when the layer's input is
        ready on the main stream, joined before `base + delta` -- the dependency structure of a real network, layer by layer.  The
        frozen GEMM of a bs-1 layer leaves most of the 256 CUs idle (20-80 workgroups); the adapter's latency-bound launches run
        beside it.  autograd runs each backward node on the stream of its forward, so the backward forks and joins the same way."""
        cur = torch.cuda.current_stream()
        outs = []
        for it in insts:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                delta = it.forward()
            base = it.base_forward()
            cur.wait_stream(side)
            delta.record_stream(cur)
            outs.append((base + delta, it))
        one_backward(outs)
        cur.wait_stream(side)

    sync._sync_enabled = False
    t_ov = None
    try:
        t_base = _graph_ms(base_pass, reps=2)
        t_both = _graph_ms(both_pass, reps=2)
        if OVERLAP_LEG:  # --overlap-leg: measured SLOWER (91.9 vs 79.0 ms, round 3): every fork / join is a cross-queue dependency in the graph
            try:
                t_ov = _graph_ms(both_overlap_pass, reps=2)
            except Exception as e:  # a capture that cannot fork / join on this stack must not take the bench line down
                print(f"[bench] overlapped base + adapter leg failed: {type(e).__name__}: {e}", file=sys.stderr)
    finally:
        sync._sync_enabled = True
        for it in insts:
            it.W = None
        torch.cuda.empty_cache()
    # the rest of the step: arena fill, plane refresh, fused AdamW (eager launches, HIP events on the current stream)
    t_rest = 0.0
    if opt is not None:
        def rest():
            for arena in sync.arenas.values():
                arena.zero_()
            ops_.refresh_lokr_planes(force=True)
            opt.step()
        rest()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            rest()
        e1.record()
        e1.synchronize()
        t_rest = e0.elapsed_time(e1) / 3
    out = {"base_only_ms": round(t_base, 2), "base_plus_adapter_ms": round(t_both, 2),
           "fill_refresh_adamw_ms": round(t_rest, 2), "step_ms": round(t_both + t_rest, 2),
           "adapter_share": round(max(t_both - t_base, 0.0) / t_both, 3),
           "what": "frozen F.linear / F.conv2d forward + input-gradient backward of the same layers (random bf16 weights) with the "
                   "adapter forward / backward in the same hipGraph; step_ms adds the gradient-arena fill, the operand-plane refresh "
                   "and fused AdamW over the adapter parameters"}
    if t_ov is not None:
        out["base_plus_adapter_overlapped_ms"] = round(t_ov, 2)
        out["overlap"] = "adapter launches of every layer on a side HIP stream, forked / joined per layer (captured in the same hipGraph)"
    return out


def cpu_baseline(algo, model):
    """The reference's CPU path (rebuild: dW -> dense op -> autograd), restated in oracle/torch_cpu.py (the Python reference cannot
    travel to the GPU box), timed on the host cores of this box on a BOUNDED sample (~25 s): one instance of every distinct Linear
    shape, fp32 and bf16, best of 3 after a warm-up; then EVERY distinct Conv2d shape once in fp32, smallest first, while the budget
    lasts (no per-MAC charge for shapes that were timed).  The step time is the sum over the shape counts; a conv
    shape the budget did not reach is charged at the per-MAC cost of the largest conv that was timed with the same window, and says so."""
    from oracle import torch_cpu
    cores = min(os.cpu_count() or 1, 64)  # beyond ~64 threads the small GEMMs only oversubscribe
    torch.set_num_threads(cores)
    specs = sdxl_unet_layers(1) if model == "sdxl" else sd15_unet_layers(4)

    def factors(spec):
        if spec["kind"] == "linear":
            cin, ksz, O = spec["I"], (), spec["O"]
        else:
            cin, ksz, O = spec["C"], (spec["k"], spec["k"]), spec["O"]
        if algo == "lokr":
            return [torch.randn(FACTOR, FACTOR) * 0.3, torch.randn(O // FACTOR, cin // FACTOR, *ksz) * 0.05]
        if algo == "locon":
            r = 16 if not ksz or ksz == (1, 1) else 8
            return [torch.randn(r, cin, *ksz) * 0.05, torch.randn(O, r, *([1] * len(ksz))) * 0.05]
        kk = ksz[0] * ksz[1] if ksz else 1
        return [torch.randn(O, 32) * 0.1, torch.randn(32, cin * kk), torch.randn(O, 32) * 0.1, torch.randn(32, cin * kk)]

    t_start = time.perf_counter()
    budget = 26.0
    key_of = lambda sp: (sp["kind"], *layer_rows(sp), sp.get("k", 1), sp.get("stride", 1))
    distinct = {}
    for sp in specs:
        distinct.setdefault(key_of(sp), [sp, 0])[1] += sp["count"]
    lin = [v for k, v in distinct.items() if k[0] == "linear"]
    conv = sorted([v for k, v in distinct.items() if k[0] != "linear"], key=lambda v: np_prod(layer_rows(v[0])))
    t32, t16 = {}, {}
    r16 = [0.0, 0.0]
    for sp, cnt in lin:
        k = key_of(sp)
        t32[k] = torch_cpu.time_layer(algo, sp, factors, torch.float32, reps=3)
        if t32[k] < 0.4 and time.perf_counter() - t_start < budget * 0.45:
            t16[k] = torch_cpu.time_layer(algo, sp, factors, torch.bfloat16, reps=3)
            r16[0] += t16[k]
            r16[1] += t32[k]
    timed_conv, charged = 0, []
    per_mac = {}  # window size -> (macs, seconds) of the largest conv timed
    for sp, cnt in conv:
        k = key_of(sp)
        macs = np_prod(layer_rows(sp))
        if time.perf_counter() - t_start < budget:
            t32[k] = torch_cpu.time_layer(algo, sp, factors, torch.float32, reps=1)
            timed_conv += 1
            per_mac[sp["k"]] = (macs, t32[k])
        else:
            ref = per_mac.get(sp["k"]) or max(per_mac.values(), default=(1, 0.0))
            t32[k] = macs * ref[1] / ref[0]
            charged.append(f"{sp['C']}->{sp['O']}@{sp['H']} k{sp['k']}s{sp['stride']}")
    tot32 = sum(t32[key_of(sp)] * cnt for sp, cnt in lin + conv)
    ratio16 = r16[0] / r16[1] if r16[1] > 0 else None
    return {"value": round(1.0 / tot32, 5), "unit": "steps/s", "cores": cores, "kind": "port",
            "bf16_over_fp32_time_ratio": round(ratio16, 3) if ratio16 else None,
            "bf16_value": round(1.0 / (tot32 * ratio16), 5) if ratio16 else None,
            "sample": f"oracle/torch_cpu.py (reference rebuild path restated in torch CPU ops): {len(lin)} distinct Linear shapes (fp32 and "
                      f"bf16, best of 3 after a warm-up) and {timed_conv} of {len(conv)} distinct Conv2d shapes (fp32, once, smallest first) "
                      f"timed in {time.perf_counter() - t_start:.1f}s; step time = sum over the shape counts"
                      + (f"; NOT reached by the budget, charged per MAC of the largest timed conv with the same window: {', '.join(charged)}"
                         if charged else "")}


def np_prod(t):
    r = 1
    for v in t:
        r *= int(v)
    return r


if __name__ == "__main__":
    main()
