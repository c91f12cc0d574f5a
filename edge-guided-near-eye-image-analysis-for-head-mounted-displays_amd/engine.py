"""Host-side execution engine: NHWC slices, packed conv layers and replayable launch plans.

PyTorch is used for device memory and streams only.  A *plan* is built once per
(network, batch, height, width, mode): it owns every activation buffer and a flat list of
prepared C-ABI calls (ctypes descriptors with raw device pointers), so running the network is a
loop of ``fn(desc, stream)`` calls with no tensor allocation, no torch.cat and no host sync.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ACT_LEAKY, ACT_NONE, ACT_RELU  # noqa: F401


import os

HALO_ENABLED = os.environ.get("EGNE_HALO", "1") != "0"
F16X3_ENABLED = os.environ.get("EGNE_F16X3", "1") != "0"      # split-f16 MFMA for layers that ask for it (BDCN)
HALO_F16_MAX_COUTP = int(os.environ.get("EGNE_HALO_F16_MAX_COUTP", "256"))
LATTICE_ENABLED = os.environ.get("EGNE_LATTICE", "1") != "0"   # dilated MSBlock groups as lattice-halo launches
LAYER_BYTES = {}          # layer name -> algorithmic bytes of its launch(es) (input slices + stored output), for bench.py --layers
TDPOOL_FUSED = os.environ.get("EGNE_TDPOOL_FUSED", "1") != "0"     # Transition_down: pooling folded into the 1x1's operand load
HALO_POOL = os.environ.get("EGNE_HALO_POOL", "1") != "0"       # ... and the halo kernel writes pool2 behind conv2_2
POOL_FUSED = os.environ.get("EGNE_POOL_FUSED", "1") != "0"     # conv1_2 writes pool1 from its epilogue (conv3x3_rs_f16.hip)
RW_MIN_W = int(os.environ.get("EGNE_RW_MIN_W", "120"))         # streamed-weights form: narrowest map
RW_MAX_COUTP = int(os.environ.get("EGNE_RW_MAX_COUTP", "32"))   # ... widest output
RW_MAX_CP = int(os.environ.get("EGNE_RW_MAX_CP", "512"))        # ... and widest input slice
RW_ENABLED = os.environ.get("EGNE_RW", "1") != "0"             # resident-weights form of the role-split 3x3 (no consumer loads)
RS_ENABLED = os.environ.get("EGNE_RS", "1") != "0"             # role-split (producer / consumer waves) 3x3 kernel for narrow inputs
RS_MIN_W = int(os.environ.get("EGNE_RS_MIN_W", "60"))
# plain-f16 plans (f16_products = 1): narrowest map of the role-split / streamed-weights forms.  With one product per multiply the halo kernel's
# load -> convert -> barrier -> multiply sequence per 32-channel chunk is pure latency (256 -> 32 at 60x80: 163 us against 144 on the streamed-weights
# kernel), and a resident-weights producer writes `o` in split-pair storage, which puts the dilated group on the ring-of-rows / strip kernels
# (121 -> 41 us at 60x80, 47 -> 33 us at 30x40)
RS_MIN_W_F16 = int(os.environ.get("EGNE_RS_MIN_W_F16", "30"))
RW_MIN_W_F16 = int(os.environ.get("EGNE_RW_MIN_W_F16", "30"))
NARROW_F32 = os.environ.get("EGNE_NARROW_F32", "1") != "0"          # <= 4 output channels: exact-fp32 vector-ALU kernel (conv_narrow_f32.hip)
TAIL16_HALO = os.environ.get("EGNE_TAIL16_HALO", "1") != "0"      # 33..48-channel slices: halo kernel (skips the zero half k-step) instead of role-split
MSDIL_ENABLED = os.environ.get("EGNE_MSDIL", "1") != "0"       # dilated MSBlock groups as one launch (sum in registers)
LATTICE_MIN_W = int(os.environ.get("EGNE_LATTICE_MIN_W", "20"))
S1X1_ENABLED = os.environ.get("EGNE_S1X1", "1") != "0"
MS1X1_ENABLED = os.environ.get("EGNE_MS1X1", "1") != "0"
C4H_MODE = os.environ.get("EGNE_C4H", "wide")     # "wide" | "all" | "off"
MS1X1_MIN_PIX = int(os.environ.get("EGNE_MS1X1_MIN_PIX", "30000"))
BIG_ENABLED = os.environ.get("EGNE_BIG", "1") != "0"
SMALL_ENABLED = os.environ.get("EGNE_SMALL", "1") != "0"      # small-problem form of the flat split-f16 kernel (64-wide tiles, split-K)
BIG_MIN_COUT = int(os.environ.get("EGNE_BIG_MIN_COUT", "256"))
BIG_MIN_CIN = int(os.environ.get("EGNE_BIG_MIN_CIN", "64"))
BIG_SPLIT_TAIL = os.environ.get("EGNE_BIG_SPLIT_TAIL", "1") != "0"
BIG_CUS = 256
BIG1_ENABLED = os.environ.get("EGNE_BIG1", "1") != "0"    # plain-f16 plans (f16_products = 1): the four-stage form of the deep trunk kernel (conv_f16_big1.hip)
WINDOW_NAME = os.environ.get("EGNE_FIT_WINDOW", "enc.b3.conv1")      # launch in front of which pending WINDOW_HOOKS are released ("none": never)
WINDOW_HOOKS = []     # callables waiting for a plan's launch window (Plan.window_at): run on the host where the plan queues that launch
EVENT_KINDS = None   # bench.py: restrict the per-launch HIP events of Plan.run(events) to these kernel families
S1X1_MIN_PIX = int(os.environ.get("EGNE_S1X1_MIN_PIX", "100000"))
HALO_F16_ENABLED = os.environ.get("EGNE_HALO_F16", "1") != "0"
ESF_SPLIT = os.environ.get("EGNE_ESF_SPLIT", "1") != "0"      # split-f16 kernel for the single-slice convs of ESF-Net EVAL plans
#   (measured: logits error vs the reference unchanged, 1.4e-4 vs 1.6e-4 with exact fp32); training plans take the same products for their
#   3x3 forward convolutions, data and weight gradients under TRAIN_SPLIT (pre-scales from device-side maxima)
ZERO_SKIP = os.environ.get("EGNE_ZERO_SKIP", "1") != "0"           # no zero pass for gradient buffers whose accesses are all covered by full-batch stores
FIRST_WRITER = os.environ.get("EGNE_FIRST_WRITER", "1") != "0"     # data gradients: the first writer of a gradient slice stores instead of accumulating
# bias gradient of the dense blocks' activation-free 1x1 'a' layers from the border sums of the following 3x3's output gradient
# (egne_pair_bias_bwd) instead of a pass over the 3x3's full-resolution data gradient
PAIR_BIAS = os.environ.get("EGNE_PAIR_BIAS", "1") != "0"
ZERO_AHEAD = os.environ.get("EGNE_ZERO_AHEAD", "1") != "0"            # gradient twins zeroed behind the previous backward, on their own stream
PAIR_BIAS_SIDE = os.environ.get("EGNE_PAIR_BIAS_SIDE", "1") != "0"    # its two small launches on the plan's second stream
MERGE_DGRAD = os.environ.get("EGNE_MERGE_DGRAD", "1") != "0"       # one data-gradient launch for adjacent raw slices of a 1x1
WGRAD_SPLIT = os.environ.get("EGNE_WGRAD_SPLIT", "1") != "0"       # training plans: 3x3 weight gradients on split-f16 products (wgrad_halo.hip)
# weight gradients on the plan's second stream (only the optimiser reads them): bf16 storage 66.5 -> 64.7-65.4 ms per B=64 step, fp32
# storage 111.5 -> 107.4; round 2 had measured no gain (433.6 vs 434.0 frames/s: the kernels next to them then filled every CU's LDS)
WGRAD_SIDE_STREAM = os.environ.get("EGNE_WGRAD_SIDE", "1") != "0"
WSCALE_EVERY = int(os.environ.get("EGNE_WSCALE_EVERY", "16"))   # training plans: steps between re-measuring max |w| of a split-f16 pack (one host sync each)
TRAIN_SPLIT = os.environ.get("EGNE_TRAIN_SPLIT", "1") != "0"     # training plans: split-f16 (22-bit products) 3x3 forward convolutions and data gradients, pre-scales taken on the device
F16X3_ASCALE = float(os.environ.get("EGNE_F16X3_ASCALE", "16"))   # pre-scale of inputs that are normalised on load (|z| <= sqrt(H*W))
STATS_FUSED = os.environ.get("EGNE_STATS_FUSED", "1") != "0"      # InstanceNorm statistics from the producing conv's epilogue
# ... in bf16-storage training plans too (bf16 3x3 kernel; BatchNorm batch statistics likewise): the statistics passes over `out` of every
# down block and the head's pre-BatchNorm tensor go (7 GB of HBM traffic per B=256 step).  Round 5 measured 0.3-1 % FEWER frames/s with it and
# left it off; with round 6's 3x3 kernel the A/B is level (1552 / 1540 without, 1544 / 1547 with, scratch/r06_env_ab.sh) and the traffic
# decides: ON (EGNE_STATS_FUSED_BF16=0: separate passes; bit-identical either way, tests/test_gpu_bf16.py)
STATS_FUSED_BF16 = os.environ.get("EGNE_STATS_FUSED_BF16", "1") != "0"
FUSE_1X1 = os.environ.get("EGNE_FUSE_1X1", "1") != "0"            # 1x1 + its consuming 3x3 as one launch (inference plans)
FUSE_1X1_MIN_W = int(os.environ.get("EGNE_FUSE_1X1_MIN_W", "60"))
FUSE_C4 = os.environ.get("EGNE_FUSE_C4", "1") != "0"              # convBlock head (3x3 on <= 4 channels + 3x3) as one launch
C1V = os.environ.get("EGNE_C1V", "1") != "0"                      # ... with the one-channel first convolution on the vector ALU (exact fp32)
CALIBRATE = os.environ.get("EGNE_CALIBRATE", "1") != "0"          # per-layer pre-scale of RAW inputs from their measured max (Plan.run)
RECAL_EVERY = int(os.environ.get("EGNE_RECAL_EVERY", "1024"))     # inference plans: runs between two calibrations (0: first run only).  The scales
#   leave 32x of head-room over the calibration batch and are re-measured on a schedule (one short sync per split launch, ~0.1 % of the
#   runs in between).  A batch beyond the head-room turns f16 operands into inf: every split-f16 epilogue tests what it stores and sets
#   the plan's sticky overflow word (round 5, OVF_CHECK); Plan.overflowed() / check_overflow() read it, the next run refuses to go on
# every convolution the planner plans is also put to the C-side chooser (egne_conv2d_auto_kind, csrc/dispatch.hip) and the two must agree
# (tests/test_gpu_nets.py turns it on for the plans of both networks at B = 64 and B = 2 and for the training plans)
CHECK_DISPATCH = os.environ.get("EGNE_CHECK_DISPATCH", "0") != "0"
DISPATCH_LOG = []          # (name, planner's kind, C side's kind) of every convolution checked
OVF_CHECK = os.environ.get("EGNE_OVF_CHECK", "1") != "0"          # inference plans: split-f16 epilogues report non-finite results (egne_conv_desc.ovf_flag, Plan.overflowed)
SMALLCIN_ENABLED = os.environ.get("EGNE_SMALLCIN", "1") != "0"   # first layers: taps folded into K (conv3x3_c4_kernel)
HALO_MIN_W = int(os.environ.get("EGNE_HALO_MIN_W", "30"))
HALO_TALL = os.environ.get("EGNE_SHALO_TALL", "1") != "0"           # conv_halo_f16.hip walks a map transposed when that takes fewer 8 x 32 tiles
HALO_F16_MIN_W = int(os.environ.get("EGNE_HALO_F16_MIN_W", "30"))      # (30x40 maps: 253 -> 194 us for 120 -> 128 channels against the flat kernel)
HALO_F16_MIN_W_NARROW = int(os.environ.get("EGNE_HALO_F16_MIN_W_NARROW", "30"))   # Cout <= 64: the flat kernel's 256x32 tiles starve the chip
HALO_MAX_COUTP = int(os.environ.get("EGNE_HALO_MAX_COUTP", "128"))
# bf16-storage plans: the per-member data gradients of a 1x1 over a would-be torch.cat as ONE launch that reads gz once
# (egne_conv1x1_bf16_multi_fwd), and -- where such a launch is the last writer of a gradient slice -- the activation mask and bias sums
# of the layer that slice belongs to in its epilogue instead of a pass of their own (egne_act_bwd_bias)
NORM_FUSE = os.environ.get("EGNE_NORM_FUSE", "1") != "0"         # ... and the InstanceNorm backward of a block's input / output inside that masking pass (egne_act_norm_bwd)
PREFIX_ACC = os.environ.get("EGNE_PREFIX_ACC", "1") != "0"       # ... first encoder-side writer of a skip tensor's gradient: accumulates onto the decoder's half, stores the other (no zero pass)
BN_ACT_FUSE = os.environ.get("EGNE_BN_ACT_FUSE", "1") != "0"     # ... and a training-mode BatchNorm's backward together with its producer's masking pass (egne_bn_act_bwd)
MULTI_DGRAD = os.environ.get("EGNE_MULTI_DGRAD", "1") != "0"
MASK_ON_WRITE = os.environ.get("EGNE_MASK_ON_WRITE", "1") != "0"
MULTI_SINGLE = os.environ.get("EGNE_MULTI_SINGLE", "1") != "0"
MASK3_ON_WRITE = os.environ.get("EGNE_MASK3_ON_WRITE", "1") != "0"     # ... and on a bf16 3x3 data gradient that is a slice's last writer
BF16_FAST1X1 = os.environ.get("EGNE_BF16_FAST1X1", "1") != "0"     # ... and the 1x1 convolutions over raw slices on the streaming bf16-MFMA kernel
BF16_DGRAD_PACK = os.environ.get("EGNE_BF16_DGRAD_PACK", "1") != "0"   # bf16-storage plans: data-gradient fragments packed straight from the forward weight
BF16_NARROW = os.environ.get("EGNE_BF16_NARROW", "1") != "0"       # bf16-storage plans: k x k convolutions onto <= 8 channels on the LDS-halo kernel (conv_narrow_bf16.hip)
BF16_FAST3X3 = os.environ.get("EGNE_BF16_FAST3X3", "1") != "0"     # bf16-storage plans: 3x3 convolutions and their data gradients on bf16 MFMAs (0: exact-fp32 implicit GEMM)


def pad8(c):
    return (int(c) + 7) // 8 * 8


def pad32(c):
    return (int(c) + 31) // 32 * 32


def _ptr(t, elem_off=0):
    return t.data_ptr() + 4 * int(elem_off)


class Piece:
    """A channel slice [off, off+Cp) of an NHWC fp32 buffer (optionally starting at sample n0)."""

    __slots__ = ("buf", "off", "C", "Cp", "n0", "scale", "shift", "act_in", "nograd", "presplit", "norm_fuse", "f16s")

    def __init__(self, buf, off, C_, Cp=None, n0=0):
        self.buf, self.off, self.C, self.Cp, self.n0 = buf, int(off), int(C_), int(Cp or pad8(C_)), int(n0)
        self.scale = self.shift = None
        self.act_in = ACT_NONE
        self.nograd = False
        self.presplit = None    # a SplitScale: the slice is held in split-pair storage (egne_conv_desc.out_split), see Plan.conv
        self.f16s = None        # a SplitScale: the slice is held as F16 halves of x * value (egne_conv_desc.out_split = 2; Plan.buf16)
        # training plans: the InstanceNorm backward of this tensor's normalised readers joins its gradient where its PRODUCING convolution
        # masks it (egne_act_norm_bwd; set by the plan builder on tensors whose producer is a convolution of the plan, Plan._pending)
        self.norm_fuse = False
        assert self.off % 4 == 0 and self.Cp % 8 == 0 and self.off + self.Cp <= buf.shape[-1]

    @property
    def stride(self):
        return self.buf.shape[-1]

    @property
    def ptr(self):
        hw = self.buf.shape[1] * self.buf.shape[2]
        return self.buf.data_ptr() + self.buf.element_size() * (self.n0 * hw * self.stride)

    def with_norm(self, scale, shift, act_in=ACT_NONE):
        p = Piece(self.buf, self.off, self.C, self.Cp, self.n0)
        p.scale, p.shift, p.act_in, p.nograd, p.norm_fuse = scale, shift, act_in, self.nograd, self.norm_fuse
        return p

    def samples(self, n0):
        return Piece(self.buf, self.off, self.C, self.Cp, n0)


class SplitScale:
    """Scale of a slice in SPLIT-PAIR storage (include/egne_hip.h, egne_conv_desc.out_split): its producer writes hi = f16(x s),
    lo = f16(x s - hi) instead of x, its one consumer copies the halves into its operand image.  ``value`` is set when the plan is
    calibrated, from a bound on the producer's output (max |in| * max_co sum |w| + max |b|): a bound that is loose by 2^k costs
    k bits of the range in which elements still split into two normal halves (2^-14 of the maximum when it is tight) and nothing
    of the 22 bits of the large elements."""

    def __init__(self):
        self.value = F16X3_ASCALE


class NeedsFp32Storage(Exception):
    """Plan.conv: a slice held as f16 (Piece.f16s) met a kernel that neither writes nor reads that storage -- the plan builder falls
    back to fp32 tensors (bdcn_new.BDCN._build)."""


F16_STORAGE = os.environ.get("EGNE_F16_STORAGE", "1") != "0"   # plain-f16 plans (f16_products = 1): conv1_1 / conv1_2 / pool1 of the edge network as f16 tensors
F16_STORAGE_DEEP = os.environ.get("EGNE_F16_STORAGE_DEEP", "1") != "0"   # ... and conv3_1 .. conv5_3, pool3 / pool4 (the deep trunk kernel stages both operands by LDS-DMA)
PRESPLIT = os.environ.get("EGNE_PRESPLIT", "1") != "0"    # MSBlock: `o` written in split-pair storage by its producer (resident-weights 3x3)
# Position p of a 32-channel block in split-pair storage holds channel 16 * ((p >> 2) & 1) + 4 * (p >> 3) + (p & 3): the producer's
# lanes end with channels {4 kg .. 4 kg + 3} and {16 + 4 kg .. 16 + 4 kg + 3} of a pixel (transposed 16x16x32 product) and store them as
# ONE 16-byte piece per plane; the consumer's weights are packed in the same order (ConvLayer.k_perm).
SPLIT_PAIR_PERM = torch.tensor([16 * ((p >> 2) & 1) + 4 * (p >> 3) + (p & 3) for p in range(32)])


class ConvLayer:
    """A convolution whose weights live in torch Parameters (OIHW) and are packed for the kernel.

    ``in_layout`` lists the (C, Cp) of the input slices in concat order; ``weights`` / ``biases``
    hold one tensor per group (three for the fused dilated branch of an MSBlock).
    """

    def __init__(self, weights, biases, in_layout, stride=1, pad=(0, 0), dils=(1,), act=ACT_NONE, pad_mode=0,
                 kernel_hw=None, cout_pad=None):
        self.weights = list(weights)
        self.biases = list(biases) if biases is not None else None
        self.in_layout = [(int(c), int(cp)) for c, cp in in_layout]
        w0 = self.weights[0]
        self.Cout = w0.shape[0]
        self.Cin = sum(c for c, _ in self.in_layout)
        self.kh, self.kw = kernel_hw if kernel_hw else (w0.shape[2], w0.shape[3])
        assert w0.numel() == self.Cout * self.Cin * self.kh * self.kw, (tuple(w0.shape), self.Cin, self.kh, self.kw)
        self.stride, self.pad, self.dils, self.act, self.pad_mode = stride, pad, tuple(dils), act, pad_mode
        self.Ktot = sum(cp for _, cp in self.in_layout)
        self.CoutP = pad32(self.Cout)
        self.Cout_store = cout_pad if cout_pad else pad8(self.Cout)
        self.G = len(self.weights)
        self.wp = None
        self.wf = None          # fragment-order pack for the LDS-halo 3x3 kernel
        self.split = False      # allow the split-f16 (f16x3) kernel for this layer (frozen nets only)
        self.need_split = False
        self.need_sfrag = False  # fragment-order f16 pack for the split-f16 halo kernel
        self.need_c4h = False    # fragment pack of the streaming split-f16 first-layer kernel
        self.c4hi = self.c4lo = None
        self.need_m1 = False     # [CoutP][Ktot] hi/lo pack (slices padded to 32) for the LDS-staged multi-slice 1x1 kernel
        self.m1hi = self.m1lo = None
        self.need_big = False    # LDS-image pack for the deep 256-wide split-f16 kernel
        self.wimg = None
        self.need_big1 = False   # hi-only LDS-image pack of the plain-f16 form of that kernel (conv_f16_big1.hip; plans with f16_products = 1)
        self.wimg1 = None
        self.need_bfrag = False  # bf16 fragment pack of the bf16-storage 3x3 kernel (conv3x3_bf16.hip)
        self.bfrag = None
        self.split1 = False      # allow the streaming split-f16 kernel for this 1x1 layer (frozen nets only)
        self.need_s1 = False
        self.s1hi = self.s1lo = None
        self.whi = self.wlo = None
        self.fhi = self.flo = None
        self.w_scale = 1.0
        self.need_flat = False
        self.need_frag = False
        self.bp = None
        self._versions = None
        self.post = None  # (scale, shift) tensors [CoutP] for a folded eval-mode BatchNorm
        self.k_perm = None  # input-channel order of the split-f16 packs (SPLIT_PAIR_PERM when the input is held in split-pair storage)

    def _kinv(self, dev):
        k, c0 = [], 0
        for c, cp in self.in_layout:
            k += list(range(c0, c0 + c)) + [-1] * (cp - c)
            c0 += c
        return torch.tensor(k, dtype=torch.int32, device=dev)

    def ensure_packed(self, dev):
        vers = tuple(w._version for w in self.weights) + tuple(
            (b._version if b is not None else -1) for b in (self.biases or [])) + (id(getattr(self, "k_perm", None)),)
        have = (getattr(self, "w40", None) is not None or not getattr(self, "need_c4", False)) and ((self.wp is not None or not self.need_flat) and (self.wf is not None or not self.need_frag)
                and (self.whi is not None or not self.need_split) and (self.fhi is not None or not self.need_sfrag)
                and (self.s1hi is not None or not self.need_s1) and (self.wimg is not None or not self.need_big) and (self.wimg1 is not None or not self.need_big1)
                and (self.m1hi is not None or not self.need_m1) and (self.c4hi is not None or not self.need_c4h)
                and (self.bfrag is not None or not self.need_bfrag)
                and (getattr(self, "b1frag", None) is not None or not getattr(self, "need_b1", False)))
        if self.bp is not None and have and vers == self._versions and self.bp.device == dev:
            return False
        L = _lib.lib()
        T = self.kh * self.kw
        n = self.G * T * self.CoutP * self.Ktot
        if self.bp is None or self.bp.device != dev:
            self.bp = torch.zeros(self.G * self.CoutP, dtype=torch.float32, device=dev)
            self.kinv = self._kinv(dev)
            self.wp = self.wf = None
        if self.need_flat and self.wp is None:
            self.wp = torch.empty(n, dtype=torch.float32, device=dev)
        if self.need_frag and self.wf is None:
            self.wf = torch.empty(n, dtype=torch.float32, device=dev)
        st = _lib.stream_ptr()
        for g, w in enumerate(self.weights):
            wd = w.detach()
            assert wd.is_cuda and wd.dtype == torch.float32
            wd = wd.contiguous()
            for need, buf, fn in ((self.need_flat, self.wp, L.egne_pack_conv_weight),
                                  (self.need_frag, self.wf, L.egne_pack_conv_weight_frag)):
                if need:
                    _lib.check(fn(wd.data_ptr(), self.Cout, self.Cin, self.kh, self.kw, self.kinv.data_ptr(),
                                  self.CoutP, self.Ktot, _ptr(buf, g * T * self.CoutP * self.Ktot), st), "pack_conv_weight")
            if self.biases is not None and self.biases[g] is not None:
                self.bp[g * self.CoutP: g * self.CoutP + self.Cout].copy_(self.biases[g].detach())
        if getattr(self, "need_c4", False):
            # [CoutP][40] fp32, column tap*4 + c (taps folded into K for the Cin <= 4 first layers); tiny host-side pack
            n32 = 32 if self.Cout_store <= 32 else 64
            if getattr(self, "w40", None) is None:
                self.w40 = torch.zeros(n32, 40, dtype=torch.float32, device=dev)
            wd = self.weights[0].detach()
            self.w40.zero_()
            self.w40[:self.Cout, :36].view(self.Cout, 9, 4)[:, :, :self.Cin].copy_(wd.permute(0, 2, 3, 1).reshape(self.Cout, 9, self.Cin))
        if self.need_s1:
            # streaming 1x1 kernel: K slots of 16-channel groups per slice, slot (h, j) <-> channel 16g + (j<4 ? 4h+j : 8+4h+j-4)
            import math
            wd = self.weights[0].detach().contiguous()
            mx = float(wd.abs().max())
            self.w_scale1 = 2.0 ** math.floor(math.log2(2048.0 / mx)) if mx > 0 else 1.0
            if self.s1hi is None:
                kmap, c0 = [], 0
                for c, cp in self.in_layout:
                    for g in range((cp + 15) // 16):
                        for h in range(2):
                            for j in range(8):
                                ch = 16 * g + (4 * h + j if j < 4 else 8 + 4 * h + j - 4)
                                kmap.append(c0 + ch if ch < c else -1)
                    c0 += c
                self.s1_G = len(kmap) // 16
                self.s1_kmap = torch.tensor(kmap, dtype=torch.int32, device=dev)
                self.s1hi = torch.empty(self.s1_G * self.CoutP * 16, dtype=torch.float16, device=dev)
                self.s1lo = torch.empty_like(self.s1hi)
            _lib.check(L.egne_pack_conv1x1_weight_f16(wd.data_ptr(), self.Cout, self.Cin, self.s1_kmap.data_ptr(), self.s1_G, self.CoutP,
                                                      self.w_scale1, self.s1hi.data_ptr(), self.s1lo.data_ptr(), st), "pack_conv1x1_f16")
        if self.need_c4h:
            import math
            wd = self.weights[0].detach().contiguous()
            mx = float(wd.abs().max())
            self.w_scale_c4 = 2.0 ** math.floor(math.log2(2048.0 / mx)) if mx > 0 else 1.0
            self.c4_coutp = 32 if self.Cout_store <= 32 else 64
            if self.c4hi is None:
                self.c4hi = torch.empty(3 * self.c4_coutp * 16, dtype=torch.float16, device=dev)
                self.c4lo = torch.empty_like(self.c4hi)
            _lib.check(L.egne_pack_conv3x3_c4_weight_f16(wd.data_ptr(), self.Cout, self.Cin, self.c4_coutp, self.w_scale_c4,
                                                         self.c4hi.data_ptr(), self.c4lo.data_ptr(), st), "pack_conv3x3_c4_f16")
        if self.need_m1:
            import math
            wd = self.weights[0].detach().contiguous()
            mx = float(wd.abs().max())
            self.w_scale_m1 = 2.0 ** math.floor(math.log2(2048.0 / mx)) if mx > 0 else 1.0
            if self.m1hi is None:
                kmap, c0 = [], 0
                for c, cp in self.in_layout:
                    kmap += list(range(c0, c0 + c)) + [-1] * (pad32(cp) - c)
                    c0 += c
                self.m1_ktot = len(kmap)
                self.m1_coutp = (self.Cout + 63) // 64 * 64
                self.m1_kmap = torch.tensor(kmap, dtype=torch.int32, device=dev)
                self.m1hi = torch.empty(self.m1_coutp * self.m1_ktot, dtype=torch.float16, device=dev)
                self.m1lo = torch.empty_like(self.m1hi)
            _lib.check(L.egne_pack_conv1x1_weight_f16x2_map(wd.data_ptr(), self.Cout, self.Cin, self.m1_kmap.data_ptr(), self.m1_coutp,
                                                            self.m1_ktot, self.w_scale_m1, self.m1hi.data_ptr(), self.m1lo.data_ptr(), st),
                       "pack_conv1x1_f16x2_map")
        if self.need_big:
            import math
            wd = self.weights[0].detach().contiguous()
            mx = float(wd.abs().max())
            self.w_scale_big = 2.0 ** math.floor(math.log2(2048.0 / mx)) if mx > 0 else 1.0
            bn = 256 if self.Cout % 256 == 0 else 128
            self.big_bn, self.big_coutp = bn, (self.Cout + bn - 1) // bn * bn
            kts = pad32(self.Ktot)
            if self.wimg is None:
                self.wimg = torch.empty(self.big_coutp * kts * T * 2, dtype=torch.float16, device=dev)
            _lib.check(L.egne_pack_conv_weight_f16img(wd.data_ptr(), self.Cout, self.Cin, self.kh, self.kw, bn, kts, self.w_scale_big,
                                                      self.wimg.data_ptr(), st), "pack_f16img")
            if self.need_big1:
                if self.wimg1 is None:
                    self.wimg1 = torch.empty(self.big_coutp * kts * T, dtype=torch.float16, device=dev)
                _lib.check(L.egne_pack_conv_weight_f16img1(wd.data_ptr(), self.Cout, self.Cin, self.kh, self.kw, bn, kts, self.w_scale_big,
                                                           self.wimg1.data_ptr(), st), "pack_f16img1")
        if getattr(self, "need_b1", False):
            _pack_b1(self, dev, st)
        if self.need_bfrag:
            # weights rounded to bf16 in MFMA-fragment order (fp32 master weights stay in the Parameter)
            wd = self.weights[0].detach().contiguous()
            kts = pad32(self.Ktot)
            if self.bfrag is None:
                self.bfrag = torch.empty(T * self.CoutP * kts, dtype=torch.bfloat16, device=dev)
            _lib.check(L.egne_pack_conv_weight_bf16frag(wd.data_ptr(), self.Cout, self.Cin, self.kh, self.kw, self.CoutP, kts,
                                                        self.bfrag.data_ptr(), st), "pack_bf16frag")
        if self.need_split or self.need_sfrag:
            # one power-of-two scale for all groups that puts max|w| in [1024, 2048): hi and lo halves stay f16-normal
            import math
            ws = [w.detach().contiguous() for w in self.weights]
            if getattr(self, "k_perm", None) is not None:       # the operand image holds the input channels in another order
                ws = [w[:, self.k_perm.to(w.device)].contiguous() for w in ws]
            age = getattr(self, "_ws_age", 0)
            if not getattr(self, "stale_scale_ok", False) or age % WSCALE_EVERY == 0:
                mx = max(float(w.abs().max()) for w in ws)          # host sync, at (re)pack time only
                self.w_scale = 2.0 ** math.floor(math.log2(2048.0 / mx)) if mx > 0 else 1.0
            self._ws_age = age + 1       # (training plans re-pack every step: the scale has 32x of headroom and is re-measured every WSCALE_EVERY steps)
            cps = self.split_coutp()
            kts = pad32(self.Ktot)           # single slice: logical channels first, zero columns up to a multiple of 32
            per = T * cps * kts
            if self.need_split:
                if self.whi is None:
                    self.whi = torch.empty(self.G * per, dtype=torch.float16, device=dev)
                    self.wlo = torch.empty(self.G * per, dtype=torch.float16, device=dev)
                for g, wd in enumerate(ws):
                    _lib.check(L.egne_pack_conv_weight_f16x2(wd.data_ptr(), self.Cout, self.Cin, self.kh, self.kw, cps, kts,
                                                             self.w_scale, self.whi.data_ptr() + 2 * g * per,
                                                             self.wlo.data_ptr() + 2 * g * per, st), "pack_f16x2")
            if self.need_sfrag:
                perf = T * self.sfrag_coutp() * kts
                if self.fhi is None:
                    self.fhi = torch.empty(self.G * perf, dtype=torch.float16, device=dev)
                    self.flo = torch.empty(self.G * perf, dtype=torch.float16, device=dev)
                for g, wd in enumerate(ws):
                    _lib.check(L.egne_pack_conv_weight_f16frag(wd.data_ptr(), self.Cout, self.Cin, self.kh, self.kw, self.sfrag_coutp(), kts,
                                                               self.w_scale, self.fhi.data_ptr() + 2 * g * perf,
                                                               self.flo.data_ptr() + 2 * g * perf, st), "pack_f16frag")
        self._versions = vers
        return True

    def sfrag_coutp(self):
        """Row count of the fragment-order f16 pack: above 64 channels a multiple of 64, so that the halo kernel runs its
        64-wide shape (96 channels: two 64-wide tiles instead of three 32-wide ones that each re-stage the halo)."""
        return self.CoutP if self.CoutP <= 64 else (self.CoutP + 63) // 64 * 64

    def split_coutp(self):
        """Row count of the f16 pack: 128-padded for wide layers (128x128 tile), 32-padded otherwise (256x32 tile)."""
        return (self.Cout + 127) // 128 * 128 if (self.Cout > 64 and self.G == 1) else self.CoutP

    def out_hw(self, H, W):
        d = self.dils[0]
        ho = (H + 2 * self.pad[0] * d - d * (self.kh - 1) - 1) // self.stride + 1
        wo = (W + 2 * self.pad[1] * d - d * (self.kw - 1) - 1) // self.stride + 1
        return ho, wo


def _pack_b1(layer, dev, st):
    """bf16 fragments of a 1x1 layer for the streaming bf16 kernel (conv1x1_bf16.hip), derived from the layer's fp32 flat pack
    ``wp`` [CoutP][Ktot] (forward weights or a data-gradient pack alike); every slice padded to whole 16-channel k-steps."""
    L = _lib.lib()
    d = _lib.ConvDesc()
    d.nseg, d.CoutP, d.Ktot = len(layer.in_layout), layer.CoutP, layer.Ktot
    for i, (_, cp) in enumerate(layer.in_layout):
        d.seg[i].Cp = cp
    if getattr(layer, "b1frag", None) is None:
        n = int(L.egne_conv1x1_bf16_pack_elems(C.byref(d)))
        assert n > 0, "too many k-steps for the streaming 1x1 kernel"
        layer.b1frag = torch.empty(n, dtype=torch.bfloat16, device=dev)
        kofs, o = [], 0
        for _, cp in layer.in_layout:
            kofs.append(o)
            o += cp
        layer.b1info = torch.tensor(kofs + [cp for _, cp in layer.in_layout], dtype=torch.int32, device=dev)
    _lib.check(L.egne_pack_conv1x1_bf16(C.byref(d), layer.wp.data_ptr(), layer.b1info.data_ptr(), layer.b1frag.data_ptr(), st), "pack_conv1x1_bf16")


class PlanarPiece(Piece):
    """An NCHW tensor [N][C][H][W] with C <= 4 read in place by the kernels that take it (first layers: conv3x3_c4_f16.hip; the
    fused convBlock head with C = 1): pixel pitch 1 float, no NHWC staging copy.  Presents the layout (C, 8) of the padded
    slice it replaces."""

    def __init__(self, buf):
        assert buf.dim() == 4 and buf.shape[1] <= 4 and buf.is_contiguous()
        self.buf, self.off, self.C, self.Cp, self.n0 = buf, 0, int(buf.shape[1]), 8, 0
        self.scale = self.shift = None
        self.act_in = ACT_NONE
        self.nograd = True

    @property
    def stride(self):
        return 1

    @property
    def ptr(self):
        return self.buf.data_ptr()

    def with_norm(self, *a, **k):
        raise RuntimeError("a planar one-channel input cannot carry a fused normalisation")


class DgradLayer(ConvLayer):
    """Data gradient of ``fwd`` w.r.t. its input slice ``idx`` as a forward convolution: the weights
    are the flipped / transposed pack of egne_pack_conv_weight_dgrad, the input is the gradient
    w.r.t. the pre-activation output (Cout_store channels)."""

    def __init__(self, fwd, idx, span=1):
        assert fwd.stride == 1 and fwd.pad_mode == 0 and fwd.G == 1, "dgrad: stride-1 zero-padded convs only"
        self.fwd, self.idx = fwd, idx
        self.ci0 = sum(c for c, _ in fwd.in_layout[:idx])
        # ``span`` > 1: adjacent un-padded slices of one buffer taken as ONE output slice (gz is then read once for all of them)
        assert span == 1 or all(c == cp for c, cp in fwd.in_layout[idx:idx + span])
        C_, Cp_ = sum(c for c, _ in fwd.in_layout[idx:idx + span]), sum(cp for _, cp in fwd.in_layout[idx:idx + span])
        self.weights, self.biases = fwd.weights, None
        self.in_layout = [(fwd.Cout, fwd.Cout_store)]
        self.Cout, self.Cin = C_, fwd.Cout
        self.kh, self.kw = fwd.kh, fwd.kw
        self.stride, self.pad_mode, self.act = 1, 0, ACT_NONE
        self.pad = (fwd.kh - 1 - fwd.pad[0], fwd.kw - 1 - fwd.pad[1])
        self.dils = fwd.dils
        self.Ktot, self.CoutP, self.Cout_store, self.G = fwd.Cout_store, pad32(C_), Cp_, 1
        self.wp = self.wf = self.bp = None
        self.need_flat = self.need_frag = False
        self.split = self.need_split = self.need_sfrag = False
        self.split1 = self.need_s1 = False
        self.s1hi = self.s1lo = None
        self.need_big, self.wimg = False, None
        self.need_big1, self.wimg1 = False, None
        self.need_m1, self.m1hi, self.m1lo = False, None, None
        self.need_c4h, self.c4hi, self.c4lo = False, None, None
        self.need_bfrag, self.bfrag = False, None
        self.whi = self.wlo = self.fhi = self.flo = None
        self.w_scale = 1.0
        self._versions, self.post = None, None

    def ensure_packed(self, dev):
        w = self.weights[0]
        vers = (w._version, w.data_ptr())
        have = ((self.wp is not None or not self.need_flat) and (self.wf is not None or not self.need_frag)
                and (getattr(self, "b1frag", None) is not None or not getattr(self, "need_b1", False)))
        if have and vers == self._versions:
            return False
        L = _lib.lib()
        n = self.kh * self.kw * self.CoutP * self.Ktot
        st = _lib.stream_ptr()
        wd = w.detach().contiguous()
        if self.need_flat:
            if self.wp is None:
                self.wp = torch.empty(n, dtype=torch.float32, device=dev)
            _lib.check(L.egne_pack_conv_weight_dgrad(wd.data_ptr(), self.fwd.Cout, self.fwd.Cin, self.kh, self.kw, self.ci0,
                                                     self.Cout, self.CoutP, self.Ktot, 0, self.wp.data_ptr(), st), "pack_dgrad")
        if self.need_frag:
            if self.wf is None:
                self.wf = torch.empty(n, dtype=torch.float32, device=dev)
            _lib.check(L.egne_pack_conv_weight_dgrad(wd.data_ptr(), self.fwd.Cout, self.fwd.Cin, self.kh, self.kw, self.ci0,
                                                     self.Cout, self.CoutP, self.Ktot, 1, self.wf.data_ptr(), st), "pack_dgrad")
        if getattr(self, "need_b1", False):
            _pack_b1(self, dev, st)
        self._versions = vers
        return True


class TransposedLayer(ConvLayer):
    """Data gradient of a reflect-padded and / or stride-2 convolution w.r.t. its PADDED input, as an ordinary
    zero-padded stride-1 convolution over the output gradient with weights derived from the forward ones:

    * stride 1, k x k:  W'[ci][co][j][i] = w[co][ci][k-1-j][k-1-i], pad k-1 -> dense [B,H+2P,W+2P,Cin]
    * stride 2, 4 x 4, P = 1 (the StyleEncoder's down-sampling blocks, utils.py:96-103): the padded position
      py = 2m + a receives gy[m - t] * w[ky = a + 2t], t in {0,1} -- one 2x2 convolution per phase (a, b); the four
      phases are the channel blocks of ONE launch: W'[(a,b,ci)][co][j][i] = w[co][ci][a+2(1-j)][b+2(1-i)], pad 1,
      output [B,Ho+1,Wo+1,4*Cp] which egne_reflect_pad_bwd un-shuffles and folds.

    ``refresh`` recomputes the derived tensor (a handful of torch ops) whenever the forward weight changed."""

    def __init__(self, fwd, idx, dev):
        assert fwd.G == 1 and len(fwd.in_layout) == 1 and idx == 0
        w = fwd.weights[0]
        C_, Cp_ = fwd.in_layout[0]
        self.fwd, self.phase = fwd, 0
        if fwd.stride == 1:
            shape, khw, pad, cout_pad = (C_, fwd.Cout, fwd.kh, fwd.kw), (fwd.kh, fwd.kw), (fwd.kh - 1, fwd.kw - 1), Cp_
        elif fwd.stride == 2 and (fwd.kh, fwd.kw) == (4, 4) and fwd.pad == (1, 1):
            assert C_ == Cp_, "phase-packed dgrad: input slice must not be channel padded"
            shape, khw, pad, cout_pad, self.phase = (4 * C_, fwd.Cout, 2, 2), (2, 2), (1, 1), 4 * Cp_, 1
        elif fwd.stride == 2 and (fwd.kh, fwd.kw) == (2, 2) and fwd.pad == (0, 0):
            # non-overlapping windows (models/deepvog_pytorch.py:22): input pixel (2m + a, 2n + b) only hears from output (m, n) through
            # tap (a, b) -- a 1x1 convolution per phase, the four phases as channel blocks: W'[(a,b,ci)][co] = w[co][ci][a][b]
            assert C_ == Cp_, "phase-packed dgrad: input slice must not be channel padded"
            shape, khw, pad, cout_pad, self.phase = (4 * C_, fwd.Cout, 1, 1), (1, 1), (0, 0), 4 * Cp_, 2
        else:
            raise NotImplementedError("dgrad of a %dx%d stride-%d convolution" % (fwd.kh, fwd.kw, fwd.stride))
        self.derived = torch.zeros(shape, dtype=torch.float32, device=dev)
        super().__init__([self.derived], None, [(fwd.Cout, fwd.Cout_store)], stride=1, pad=pad, kernel_hw=khw, cout_pad=cout_pad)
        self.guard = VersionGuard([w], self.refresh)

    def refresh(self):
        w = self.fwd.weights[0].detach()
        if self.phase == 2:
            Co, Ci = w.shape[:2]
            self.derived.copy_(w.permute(2, 3, 1, 0).reshape(4 * Ci, Co, 1, 1))
        elif self.phase:
            Co, Ci = w.shape[:2]
            v = w.view(Co, Ci, 2, 2, 2, 2).flip(2).flip(4)            # [co][ci][j][a][i][b], j = 1 - t
            self.derived.copy_(v.permute(3, 5, 1, 0, 2, 4).reshape(4 * Ci, Co, 2, 2))
        else:
            self.derived.copy_(w.flip(2).flip(3).transpose(0, 1))


class SplitDgradLayer(ConvLayer):
    """Data gradient of a stride-1 zero-padded k x k convolution w.r.t. input slice ``idx`` as an ORDINARY convolution over the
    output gradient -- W'[ci][co][j][i] = w[co][ci0 + ci][k-1-j][k-1-i], pad k-1-p -- so that every split-f16 forward kernel
    (role-split, resident-weights, halo, flat) serves the backward pass too.  ``guard`` re-derives the tensor when the forward
    weight changed (one strided copy); the usual packing reads it."""

    def __init__(self, fwd, idx, dev):
        assert fwd.stride == 1 and fwd.pad_mode == 0 and fwd.G == 1
        C_, Cp_ = fwd.in_layout[idx]
        self.fwd, self.c0, self.cn = fwd, sum(c for c, _ in fwd.in_layout[:idx]), C_
        self.derived = torch.zeros((C_, fwd.Cout, fwd.kh, fwd.kw), dtype=torch.float32, device=dev)
        super().__init__([self.derived], None, [(fwd.Cout, fwd.Cout_store)], stride=1, dils=fwd.dils,
                         pad=(fwd.kh - 1 - fwd.pad[0], fwd.kw - 1 - fwd.pad[1]), kernel_hw=(fwd.kh, fwd.kw), cout_pad=Cp_)
        self.split = True
        self.guard = VersionGuard([fwd.weights[0]], self.refresh)

    def refresh(self):
        w = self.fwd.weights[0].detach()
        self.derived.copy_(w[:, self.c0:self.c0 + self.cn].flip(2).flip(3).transpose(0, 1))     # (copy_ bumps the version: re-packed)


class BfDgradLayer(ConvLayer):
    """Data gradient of a stride-1 zero-padded k x k convolution w.r.t. input slice ``idx`` for bf16-storage plans: the ordinary
    convolution over gz that conv3x3_bf16.hip runs, its bf16 fragments packed STRAIGHT from the forward weight
    (egne_pack_conv_weight_bf16frag_dgrad: flip and transpose are index arithmetic of the pack kernel) -- one launch per layer and
    step where SplitDgradLayer cost two flips, a strided copy and the pack."""

    def __init__(self, fwd, idx):
        assert fwd.stride == 1 and fwd.pad_mode == 0 and fwd.G == 1
        C_, Cp_ = fwd.in_layout[idx]
        self.fwd, self.c0, self.cn = fwd, sum(c for c, _ in fwd.in_layout[:idx]), C_
        w = fwd.weights[0]
        self.weights, self.biases = [w], None
        self.in_layout = [(fwd.Cout, fwd.Cout_store)]
        self.Cout, self.Cin = C_, fwd.Cout
        self.kh, self.kw = fwd.kh, fwd.kw
        self.stride, self.pad_mode, self.act, self.dils = 1, 0, ACT_NONE, fwd.dils
        self.pad = (fwd.kh - 1 - fwd.pad[0], fwd.kw - 1 - fwd.pad[1])
        self.Ktot, self.CoutP, self.Cout_store, self.G = fwd.Cout_store, pad32(C_), Cp_, 1
        self.bp = self.bfrag = self.post = None
        self.need_bfrag = self.need_flat = False
        self.split = self.split1 = False
        self._versions = None

    def ensure_packed(self, dev):
        w = self.weights[0]
        vers = (w._version, w.data_ptr())
        if self.bfrag is not None and vers == self._versions:
            return False
        assert self.need_bfrag, "BfDgradLayer serves the bf16 3x3 kernel only"
        kts = pad32(self.Ktot)
        if self.bfrag is None:
            self.bfrag = torch.empty(self.kh * self.kw * self.CoutP * kts, dtype=torch.bfloat16, device=dev)
        wd = w.detach().contiguous()
        _lib.check(_lib.lib().egne_pack_conv_weight_bf16frag_dgrad(wd.data_ptr(), self.fwd.Cout, self.fwd.Cin, self.kh, self.kw, self.c0, self.cn,
                                                                   self.CoutP, kts, self.bfrag.data_ptr(), _lib.stream_ptr()), "pack_bf16frag_dgrad")
        self._versions = vers
        return True


class Plan:
    """Buffers + prepared launches for one network at one shape."""

    def __init__(self, device, train=False, dtype=torch.float32):
        """``dtype``: storage type of every NHWC activation / activation-gradient buffer of the plan (``buf``): fp32, or bf16 for
        training plans that keep them in HBM at half the bytes (BASELINE.json configs[2..4]).  Arithmetic, statistics, tables
        (``vec``), master weights and parameter gradients are fp32 either way."""
        assert dtype in (torch.float32, torch.bfloat16), dtype
        self.device = device
        self.dtype, self.bf16 = dtype, dtype == torch.bfloat16
        self.esz = 2 if self.bf16 else 4
        self.train = train  # record a tape of backward emitters while the forward plan is built
        self.tape = []
        self.gtwins = {}    # id(forward buffer) -> gradient buffer of the same shape
        self.calls = []     # (fn, desc_or_args tuple, name)
        self.keep = []      # tensors / descriptors that must outlive the plan's calls
        self.layers = []    # ConvLayers to (re)pack before running
        self.pre = []       # python callables run before the launches (BN folding etc.)
        self.meta = []      # per call: (kernel family, algorithmic FLOPs) for bench.py's roofline
        self.wscale_refs = []   # (call index, argument index, layer, attribute): weight-pack scales baked into launch arguments
        self.post_cal = {}  # call index -> (descriptor, f16 output Piece, its SplitScale, pixels): scale of an f16 output re-measured behind the calibrating launch
        self.cal = {}       # call index -> (index of the a_scale argument, raw input Pieces, pixels): split-f16 pre-scale calibration
        self.calibrated = False
        self.dyn_scales = bool(train) and TRAIN_SPLIT and not self.bf16     # split-f16 pre-scales taken on the device (egne_conv_desc.dyn_scale)
        self.dynbuf, self.ndyn = None, 0
        self._zstream, self._zevent, self._zero_pending = None, None, False      # zero_grads_ahead
        self._ztable = None                           # zero_grads: (twin addresses, device table, blocks)
        self._pair_links = {}                         # 1x1 -> 3x3 pairs whose bias gradients share one reduction (_bw_conv)
        self._touching, self._touched = False, {}     # build_backward: channel ranges of gradient twins already handed out
        self.side_calls, self.side_stream = {}, None   # call index -> event: launches on the plan's second stream (weight gradients; the edge network's MSBlocks)
        self.side_default = False                      # _add: launches emitted while this is set go to the second stream
        self.join_before = set()                       # call indices in front of which the main stream waits for the second one
        self.window_at = None                          # call index at which pending WINDOW_HOOKS are released (None: never)
        self.serial_timing = False                     # run(events): one stream when launches are timed (overlapping kernels stretch each other's durations)
        # f16 overflow of calibrated pre-scales (inference plans): device word set by the split-f16 epilogues, pinned host mirror copied
        # behind every run, the event that says the copy has landed
        self.ovf = self.ovf_host = self.ovf_event = None
        self.overflow_events = 0
        self._want_partials, self.last_partials = False, None      # _conv_bf16: the caller wants the statistics partials of the next 3x3 launch
        self._mask_cands3 = {}                         # as _mask_cands, for slices whose last writer is a bf16 3x3 data gradient (egne_conv_desc.mask_y)
        self._last_b3_desc = None
        self._premasked = {}                           # (buffer id, first channel) -> samples whose output gradient already is the masked gz, bias sums taken (esf_engine._train_bn)
        self._pending = {}                             # (buffer id, first channel, first sample) -> normalisation-backward addends waiting for the tensor's producer (_bw_conv)
        self._mask_cands = {}                          # (buffer id, first channel, channels, samples) -> multi-destination launch that wrote the slice last (_bw_conv)
        self.f16_products = 0                          # egne_conv_desc.f16_products of the plan's split-f16 launches (1: plain f16 operands; BDCN.f16_products)
        self.tail_at, self.tail_hook = None, None      # backward plans: call index where every non-encoder parameter gradient is final, and what to call there (parallel.GradOverlap.tail_ready)
        self._absmax_of, self._dyn_hint = {}, None   # published max |x| words: (buffer, slice, samples) -> (word, call index); forced word
        self.L = _StorageLib(_lib.lib(), self.bf16)

    # ---- memory ------------------------------------------------------------------------------
    def buf(self, B, H, W, Ctot):
        t = torch.zeros((B, H, W, int(Ctot)), dtype=self.dtype, device=self.device)
        self.keep.append(t)
        return t

    def buf16(self, B, H, W, Ctot):
        """An NHWC activation buffer of f16 elements (Piece.f16s: egne_conv_desc.out_split = 2 of its producer)."""
        t = torch.zeros((B, H, W, int(Ctot)), dtype=torch.float16, device=self.device)
        self.keep.append(t)
        return t

    def vec(self, *shape, dtype=torch.float32):
        t = torch.zeros(shape, dtype=dtype, device=self.device)
        self.keep.append(t)
        return t

    # ---- gradients (training plans) ------------------------------------------------------------
    def gbuf(self, buf, _whole=True):
        t = self.gtwins.get(id(buf))
        if t is None:
            t = torch.zeros_like(buf)
            self.gtwins[id(buf)] = t
            self.keep.append(buf)
        if _whole and self._touching:
            self._touched.setdefault(id(buf), []).append([0, int(buf.shape[-1]), False, 0, int(buf.shape[0])])
        return t

    def gp(self, piece, B=None):
        """Gradient twin of ``piece``.  While the backward plan is built the access is recorded as [first channel, end channel,
        stored?, first sample, end sample): the samples [n0, n0 + B) when the caller says how many it touches, else the whole buffer."""
        if self._touching:
            n0, n1 = (piece.n0, piece.n0 + int(B)) if B is not None else (0, int(piece.buf.shape[0]))
            self._touched.setdefault(id(piece.buf), []).append([piece.off, piece.off + piece.Cp, False, n0, n1])
        return Piece(self.gbuf(piece.buf, False), piece.off, piece.C, piece.Cp, piece.n0)

    def mark_stored(self, piece, B, n0=None):
        """The access just recorded for ``piece`` (the last gp call) is a STORE of the samples [n0, n0 + B) (n0: piece.n0): reads of
        these channels and samples later in the backward pass see this pass' values, whatever the twin held before (zero_grads can
        skip the buffer if that holds for all of its accesses)."""
        if self._touching:
            ent = self._touched[id(piece.buf)][-1]
            assert ent[0] == piece.off and ent[1] == piece.off + piece.Cp
            n0 = piece.n0 if n0 is None else n0
            ent[2], ent[3], ent[4] = True, n0, n0 + int(B)

    def first_touch(self, buf, off, Cp, n0=0, B=None):
        """While the backward plan is built: True if no emitter before this one has asked for any of the channels [off, off+Cp) of the
        samples [n0, n0+B) (B = None: all) of ``buf``'s gradient twin (gp / gbuf are the only ways to reach a twin, and emitters run
        in execution order).  The twin was zeroed before the backward pass (or needs no zeroing at all if every access is covered by
        stores), so the first writer may STORE instead of accumulate (no read of the slice)."""
        m0, m1 = (n0, n0 + int(B)) if B is not None else (0, int(buf.shape[0]))
        return FIRST_WRITER and self._touching and all(b <= off or a >= off + Cp or e1 <= m0 or e0 >= m1
                                                       for a, b, _, e0, e1 in self._touched.get(id(buf), ()))

    def touched_prefix(self, buf, off, Cp, n0, B):
        """How many leading samples k of [n0, n0+B) earlier emitters have touched in the channels [off, off+Cp), if the touched samples
        are exactly the prefix [n0, n0+k) -- the decoder's skip gradients reach the image half of an encoder tensor only, so the first
        encoder-side writer accumulates onto k = B/2 samples and STORES the rest.  B when no such prefix exists (accumulate all)."""
        if not (FIRST_WRITER and PREFIX_ACC and self._touching):
            return int(B)
        segs = sorted((max(e0, n0), min(e1, n0 + B)) for a, b, _, e0, e1 in self._touched.get(id(buf), ())
                      if a < off + Cp and b > off and e0 < n0 + B and e1 > n0)
        if not segs:
            return 0
        end = n0
        for s0, s1 in segs:
            if s0 > end:
                return int(B)          # a gap: not a prefix
            end = max(end, s1)
        return int(end - n0)

    def norm_fusable(self, piece):
        """True if the InstanceNorm backward of ``piece``'s normalised readers is deferred to its producer's masking pass."""
        return NORM_FUSE and self.bf16 and self.train and bool(getattr(piece, "norm_fuse", False))

    def defer_norm_bwd(self, piece, scale, shift, B, H, W, a1=None, gq=None, act_q=ACT_NONE):
        """Register an upstream gradient of IN(piece) (``a1``: full resolution; ``gq``: of the 2x2-pooled act_q(IN(piece))) for the fused
        backward that ``piece``'s producing convolution emits (_bw_conv): every reader of one normalised tensor shares its
        (scale, shift), and the normalisation's backward is linear in the upstream gradient."""
        ent = self._pending.setdefault((id(piece.buf), piece.off, piece.n0), dict(scale=scale, shift=shift, B=B, H=H, W=W, a1=None, gq=None, act_q=ACT_NONE))
        assert ent["scale"] is scale and ent["shift"] is shift and (ent["B"], ent["H"], ent["W"]) == (B, H, W), "readers of one normalised tensor share its statistics"
        if a1 is not None:
            assert ent["a1"] is None
            ent["a1"] = a1
        if gq is not None:
            assert ent["gq"] is None
            ent["gq"], ent["act_q"] = gq, act_q

    def flush_deferred_norm(self, bw, piece, name):
        """A producer of ``piece`` that is not a convolution of the plan (the head's BatchNorm in front of block 0): the deferred
        InstanceNorm backward of its normalised readers as a pass of its own, still ONE statistics + ONE apply pass for both readers
        (egne_act_norm_bwd without mask and bias sums) -- called by that producer's backward emitter before it reads the gradient."""
        pend = self._pending.pop((id(piece.buf), piece.off, piece.n0), None)
        if pend is None:
            return
        L = self.L
        B, H, W = pend["B"], pend["H"], pend["W"]
        g = self.gp(Piece(piece.buf, piece.off, piece.C, piece.Cp, piece.n0))
        sums = bw.vec(B * piece.Cp * 2)
        wsn = bw.vec((int(L.egne_norm_bwd_workspace_bytes(B, H * W, piece.Cp, 1)) + 7) // 8, dtype=torch.float64)
        wsb = bw.vec((int(L.egne_act_bwd_bias_workspace_bytes(B * H * W, piece.Cp)) + 7) // 8, dtype=torch.float64)
        a1, gq = pend["a1"], pend["gq"]
        bw.raw(L.egne_act_norm_bwd, (g.ptr, g.stride, g.off, piece.ptr, piece.stride, piece.off, ACT_NONE, pend["scale"].data_ptr(), pend["shift"].data_ptr(),
                                     a1.ptr if a1 is not None else None, a1.stride if a1 is not None else 0, a1.off if a1 is not None else 0,
                                     gq.ptr if gq is not None else None, gq.stride if gq is not None else 0, gq.off if gq is not None else 0,
                                     pend["act_q"], piece.Cp, B, H, W, sums.data_ptr(), wsn.data_ptr(), None, 0, wsb.data_ptr(), B), name + ".norm_bwd_fused")

    def build_backward(self):
        """Replay the tape in reverse into a second plan that shares this plan's gradient buffers."""
        bw = Plan(self.device, dtype=self.dtype)
        bw.fwd = self
        bw.dyn_scales = self.dyn_scales
        bw.serial_timing = True           # run(events): per-launch timing on one stream (the side-stream launches would stretch their neighbours)
        self._touching, self._touched = True, {}
        for emit in reversed(self.tape):
            emit(bw)
        self._touching = False
        assert not self._pending, "deferred normalisation gradients without a producer to take them: %s" % list(self._pending)
        # gradient twins that need no zero pass: every access is a full-batch store or touches only channels stored earlier
        self._zero_free = set()
        if ZERO_SKIP:
            for bid, ents in self._touched.items():
                # per run of samples between two boundaries of the recorded accesses: the channel-range test
                cuts = sorted({e for ent in ents for e in ent[3:5]})
                ok = True
                for s0, s1 in zip(cuts, cuts[1:]):
                    cov = []
                    for a, b_, st, e0, e1 in ents:
                        if e1 <= s0 or e0 >= s1:
                            continue
                        if st:
                            cov.append((a, b_))
                            continue
                        need = [(a, b_)]
                        for ca, cb in cov:       # subtract the covered ranges
                            need = [r for x0, x1 in need for r in ((x0, min(x1, ca)), (max(x0, cb), x1)) if r[0] < r[1]]
                        if need:
                            ok = False
                            break
                    if not ok:
                        break
                if ok:
                    self._zero_free.add(bid)
        self.bw = bw
        return bw

    def zero_grads(self):
        free = getattr(self, "_zero_free", ())
        ts = [t for bid, t in self.gtwins.items() if bid not in free]
        if not ts:
            return
        key = tuple(t.data_ptr() for t in ts)
        if self._ztable is None or self._ztable[0] != key:       # one launch for all of them (egne_zero_many)
            rows, blk = [], 0
            for t in ts:
                nb = t.numel() * t.element_size()
                assert t.is_contiguous() and t.data_ptr() % 16 == 0 and nb % 16 == 0
                rows.append((t.data_ptr(), nb, blk))
                blk += (nb + 65535) // 65536
            self._ztable = (key, torch.tensor(rows, dtype=torch.int64).to(self.device), blk)
        _, tab, blk = self._ztable
        _lib.check(self.L.egne_zero_many(tab.data_ptr(), len(ts), blk, _lib.stream_ptr()))

    def zero_grads_ahead(self):
        """Zero the gradient twins for the NEXT backward pass on a stream of their own, behind everything queued so far: nothing
        reads a twin between two backward passes, so the 0.7 ms of fills overlap the optimiser step and the next forward (the
        frozen edge network is MFMA bound) instead of opening the next backward."""
        if not ZERO_AHEAD:
            return
        if self._zstream is None:
            self._zstream, self._zevent = torch.cuda.Stream(device=self.device), torch.cuda.Event()
            # no join follows these fills until the next backward: tell the allocator, so that a plan dropped in between cannot have
            # a twin's memory handed to a new tensor while its fill is still queued
            for t in self.gtwins.values():
                t.record_stream(self._zstream)
        self._zstream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._zstream):
            self.zero_grads()
        if self._ztable is not None:
            self._ztable[1].record_stream(self._zstream)
        self._zevent.record(self._zstream)
        self._zero_pending = True

    def zero_grads_join(self):
        """Before a backward pass: wait for the fills queued by zero_grads_ahead, or zero now if there were none."""
        if self._zero_pending:
            torch.cuda.current_stream().wait_event(self._zevent)
            self._zero_pending = False
        else:
            self.zero_grads()

    # ---- f16 overflow of the calibrated pre-scales ---------------------------------------------------------
    def ovf_ptr(self):
        """Device address of the plan's sticky overflow word for a split-f16 launch descriptor (egne_conv_desc.ovf_flag), or None:
        only plans whose pre-scales are CALIBRATED need it (training plans take theirs on the device, bf16 plans have none)."""
        if not OVF_CHECK or self.train or self.bf16 or self.dyn_scales or getattr(self, "fwd", None) is not None:
            return None
        if self.ovf is None:
            self.ovf = self.vec(1, dtype=torch.int32)
            self.ovf_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self.ovf_event = torch.cuda.Event()
        return self.ovf.data_ptr()

    OVF_MIRROR_EVERY = 16

    def _ovf_publish(self):
        """Behind the launches of a run.  The word itself stays on the device: ``overflowed()`` reads it where the caller
        synchronises anyway.  For callers that never ask, every 16th run queues a 4-byte copy to pinned host memory, which the start
        of a later run looks at without waiting (replays of a hipGraph capture bypass this code: their callers ask) -- a copy and
        an event behind EVERY run cost the pipelined B=64 inference loop 1.2 % (measured; the kernels' own tests cost 0.2 %)."""
        if self.ovf is None:
            return
        self._ovf_runs = getattr(self, "_ovf_runs", 0) + 1
        if self._ovf_runs % self.OVF_MIRROR_EVERY or torch.cuda.is_current_stream_capturing():
            return
        self.ovf_host.copy_(self.ovf, non_blocking=True)
        self.ovf_event.record()
        self._ovf_recorded = True

    def _ovf_reset(self):
        self.ovf.zero_()
        self.ovf_host.zero_()
        self.calibrated = False          # the next run measures the maxima of ITS batch and takes its scales from them
        self.overflow_events += 1

    def overflowed(self, wait=True):
        """True if a run since the last call produced non-finite values in a split-f16 epilogue: an activation left the f16 range of
        its calibrated pre-scale (a batch beyond 32x the calibration maxima) -- or the inputs were not finite.  The outputs of
        that run are INVALID.  Clears the word and marks the plan for re-calibration, so the caller's answer is simply to run
        the batch again.  ``wait``: block until the last run's word has reached the host (callers ask where they synchronise
        anyway: test.py / evaluate.py when they read the masks); without it only a completed run is judged."""
        if self.ovf is None:
            return False
        if wait:
            # (the caller synchronises here anyway: a 4-byte read of the device word behind everything queued on this stream)
            if int(self.ovf.item()) == 0:
                return False
        else:
            if not getattr(self, "_ovf_recorded", False) or not self.ovf_event.query() or int(self.ovf_host[0]) == 0:
                return False
        self._ovf_reset()
        return True

    def check_overflow(self):
        """``overflowed()`` answered in place: re-calibrate on the batch that is still in the plan's input buffers and run it again.
        Returns True if that happened (the caller re-reads the outputs); raises if the second run is not finite either."""
        if not self.overflowed():
            return False
        self.run()
        if self.overflowed():
            raise RuntimeError("non-finite values in a split-f16 convolution after re-calibration: the inputs themselves are not finite")
        return True

    # ---- launches ----------------------------------------------------------------------------
    DYN_SLOTS = 512

    def _new_slot(self):
        if self.dynbuf is None:
            self.dynbuf = self.vec(self.DYN_SLOTS, dtype=torch.int32)
            self.pre.append(self.dynbuf.zero_)
        assert self.ndyn < self.DYN_SLOTS, "too many device-scaled launches in one plan"
        self.ndyn += 1
        return self.dynbuf.data_ptr() + 4 * (self.ndyn - 1)

    def _publish_absmax(self, piece, B):
        """A producer writes max |x| of the slice it stores into the returned word (cleared at the start of every run)."""
        ptr = self._new_slot()
        self._absmax_of[(id(piece.buf), piece.off, piece.Cp, piece.n0, B)] = (ptr, len(self.calls))
        return ptr

    def _dyn_slot(self, d, pieces, B, npix, name):
        """Training plans: the split-f16 pre-scale comes from max |x| of the raw input slices, on the device (the kernel derives
        it from one word of ``dynbuf``): the word its producer published (act_bwd_bias, the fp32 implicit GEMM's epilogue) if
        there is one, else measured by egne_absmax right before the launch."""
        if self._dyn_hint is not None:
            d.dyn_scale = self._dyn_hint
            return
        if len(pieces) == 1:
            pc = pieces[0]
            ent = self._absmax_of.get((id(pc.buf), pc.off, pc.Cp, pc.n0, B))
            if ent is not None and len(self.calls) - ent[1] <= 4:
                d.dyn_scale = ent[0]
                return
        ptr = self._new_slot()
        for pc in pieces:
            self._add(self.L.egne_absmax, (pc.ptr, pc.stride, pc.off, pc.Cp, npix, ptr), name + ".absmax", kind="absmax")
        d.dyn_scale = ptr

    def _add(self, fn, args, name, flops=0.0, kind=None, cal=None, side=False, ws=()):
        """``ws``: (argument index, layer, attribute) for every weight-pack scale passed BY VALUE in ``args``: ensure_packed may
        re-measure max |w| and repack with another power of two, and Plan.run then rewrites these arguments (_refresh_wscales)."""
        for ai, layer, attr in ws:
            assert args[ai] == getattr(layer, attr), (name, ai, attr)
            self.wscale_refs.append((len(self.calls), ai, layer, attr))
        if cal is not None and CALIBRATE:
            self.cal[len(self.calls)] = cal
        if side or self.side_default:      # launched on the plan's second stream behind an event of the main one (Plan.run); joined at the end of the run
            self.side_calls[len(self.calls)] = None
        if name == WINDOW_NAME and self.window_at is None and not self.train:
            self.window_at = len(self.calls)       # WINDOW_HOOKS are released in front of this launch (_open_window)
        self.calls.append((fn, args, name))
        self.meta.append((kind or name.split(".")[0], flops))

    def _stats_ws(self, d, B, nchunk):
        """Partial-sum workspace for InstanceNorm statistics written from a convolution's epilogue."""
        ws = self.vec(B * nchunk * int(d.Cout_store) * 2, dtype=torch.float64)
        d.stats_ws, d.stats_nchunk = ws.data_ptr(), nchunk
        return ws

    def _stats_finish(self, ws, d, B, HW, nchunk, name, eps=1e-5):
        Cs = int(d.Cout_store)
        scale, shift = self.vec(B, Cs), self.vec(B, Cs)
        self._add(self.L.egne_norm_stats_finish, (ws.data_ptr(), Cs, B, nchunk, HW, eps, scale.data_ptr(), shift.data_ptr()),
                  name + ".stats", kind="norm_stats")
        return scale, shift

    def msdil_ok(self, layer, piece, H, W):
        """True if pl.conv will run this grouped dilated MSBlock convolution on the one-launch kernel (msblock_dil_f16.hip)."""
        return (F16X3_ENABLED and layer.split and MSDIL_ENABLED and layer.G == 3 and layer.kh == 3 and layer.kw == 3
                and layer.pad == (1, 1) and layer.dils == (4, 8, 12) and layer.CoutP == 32 and piece.Cp == 32 and piece.scale is None
                and layer.act == ACT_RELU and layer.post is None and H * W * piece.stride < 2 ** 29)

    def stream1x1_ok(self, layer, pieces, B, H, W, up_add=False):
        """True if ``conv`` runs this 1x1 over raw slices on the streaming split-f16 kernel (weights in LDS, operands straight from
        HBM: conv1x1_f16.hip) -- the kernel that can add an up-sampled half-resolution tensor in its epilogue (``up_add``)."""
        return (F16X3_ENABLED and S1X1_ENABLED and not self.bf16 and not self.dyn_scales and layer.split1 and layer.kh == 1 and layer.kw == 1
                and layer.stride == 1 and layer.G == 1 and layer.pad == (0, 0) and layer.post is None
                and all(pc.scale is None for pc in pieces) and (layer.CoutP == 32 or layer.CoutP % 64 == 0)
                and sum((pc.Cp + 15) // 16 for pc in pieces) * (1 if layer.CoutP == 32 else 2) * 2048
                + (4 * 18 * (32 * (1 if layer.CoutP == 32 else 2) + 8) * 4 if up_add else 0) <= 80 * 1024       # (+ the addend patch of each wave)
                and B * H * W >= S1X1_MIN_PIX)

    def _check_dispatch(self, n0, layer, pieces, dst, B, H, W, residual, name, stats, scores, pool, up_add):
        """EGNE_CHECK_DISPATCH=1: describe the layer to egne_conv2d_auto_kind and demand the kind this planner just recorded (the first
        launch the convolution added to Plan.meta), the frames it handed to the flat kernel behind the deep trunk kernel and what its
        epilogue took along."""
        conv_kinds = [k for k, _ in self.meta[n0:] if k.startswith("conv")]
        if not conv_kinds:
            return
        mine = conv_kinds[0]
        q = _lib.ConvQuery()
        q.dtype = 1 if self.bf16 else 0
        q.B, q.H, q.W = B, H, W
        q.kh, q.kw, q.stride, q.pad_h, q.pad_w, q.pad_mode, q.ngroups = layer.kh, layer.kw, layer.stride, layer.pad[0], layer.pad[1], layer.pad_mode, layer.G
        for g in range(_lib.MAXGROUP):
            q.dil[g] = layer.dils[g] if g < layer.G else 1
        q.nseg = len(pieces)
        for i, p in enumerate(pieces):
            q.seg_C[i], q.seg_Cp[i], q.seg_ch_off[i], q.seg_pix_stride[i] = p.C, p.Cp, p.off, p.stride
            q.seg_affine[i] = int(p.scale is not None)
        p0 = pieces[0]
        q.seg_planar, q.seg_presplit = int(isinstance(p0, PlanarPiece)), int(getattr(p0, "presplit", None) is not None)
        q.Cout, q.Cout_store = layer.Cout, layer.Cout_store
        q.dst_Cp, q.dst_ch_off, q.dst_pix_stride = dst.Cp, dst.off, dst.stride
        q.act, q.has_post, q.has_residual = layer.act, int(layer.post is not None), int(residual is not None)
        if residual is not None:
            q.res_pix_stride, q.res_ch_off = residual.stride, residual.off
        q.split, q.split1, q.split_c4 = int(bool(layer.split)), int(bool(layer.split1)), int(bool(getattr(layer, "split_c4", False)))
        q.train, q.dyn_scales = int(bool(self.train)), int(bool(self.dyn_scales))
        q.is_dgrad = int(isinstance(layer, (DgradLayer, SplitDgradLayer, BfDgradLayer)))
        q.want_stats = int(bool(stats) and STATS_FUSED)
        q.want_scores = int(scores is not None)
        if pool is not None and POOL_FUSED:
            q.want_pool, q.pool_Cp, q.pool_pix_stride = 1, pool.Cp, pool.stride
        q.up_add = int(up_add is not None)
        q.narrow_bf16_ok = int(mine == "conv_bf16:narrow")
        q.f16_products = int(self.f16_products)
        q.f16_storage = int(getattr(p0, "f16s", None) is not None or dst.f16s is not None)
        ch = _lib.ConvChoice()
        _lib.check(self.L.egne_conv2d_auto_kind(C.byref(q), C.byref(ch)), "conv2d_auto_kind")
        theirs = ch.name.decode()
        DISPATCH_LOG.append((name, mine, theirs))
        assert theirs == mine, "%s: the planner chose %s, egne_conv2d_auto_kind %s" % (name, mine, theirs)
        if mine == "conv_f16x3:big":
            tails = [k for k, _ in self.meta[n0:] if k == "conv_f16x3:flat"]
            assert bool(ch.tail_frames) == bool(tails), "%s: frame tail %d against %s" % (name, ch.tail_frames, tails)
        if mine in ("conv_f16x3:rw", "conv_f16x3:rs", "conv_f16x3:halo") and not self.bf16:
            assert bool(ch.fused_pool) == bool(self.last_pooled), "%s: pooled output %d against %s" % (name, ch.fused_pool, self.last_pooled)

    def conv(self, layer, pieces, dst, B, H, W, residual=None, name="conv", stats=False, scores=None, pool=None, up_add=None):
        """``pool``: a Piece for the 2x2 / stride-2 ceil-mode max pooling of the result; ``self.last_pooled`` tells the caller
        whether the convolution kernel wrote it (otherwise the caller runs maxpool2).  ``up_add`` = (P, ph, pw): a half-resolution
        tensor whose bilinear x2 up-sampling is added to the result of a streaming 1x1 (callers check ``stream1x1_ok`` first)."""
        self._pool_req, self.last_pooled, self.last_presplit, self._up_add = pool, False, False, up_add
        n0 = len(self.meta)
        r = self._conv_impl(layer, pieces, dst, B, H, W, residual, name, stats, scores)
        if CHECK_DISPATCH:
            self._check_dispatch(n0, layer, pieces, dst, B, H, W, residual, name, stats, scores, pool, up_add)
        self._pool_req = self._up_add = None
        # algorithmic bytes of the layer: every input slice read once, the output written once -- in the storage the plan holds them in (a plain-f16
        # plan keeps some tensors as f16: Piece.f16s; of a split-pair slice it reads and writes the hi plane only)
        def esz_of(p):
            if getattr(p, "f16s", None) is not None:
                return 2.0
            if self.f16_products == 1 and getattr(p, "presplit", None) is not None:
                return 2.0
            return float(self.esz)
        LAYER_BYTES[name] = B * (H * W * sum(p.Cp * esz_of(p) for p in pieces) + r[0] * r[1] * (min(layer.Cout_store, dst.Cp) * esz_of(dst) + (residual.Cp * esz_of(residual) if residual is not None else 0)))
        return r

    def _conv_impl(self, layer, pieces, dst, B, H, W, residual=None, name="conv", stats=False, scores=None):
        """pieces: input Pieces in concat order; dst: output Piece.  Returns (Ho, Wo); with ``stats`` also leaves the
        per-sample InstanceNorm (scale, shift) of the OUTPUT in ``self.last_stats`` -- from partial sums written by the
        kernel's epilogue where the kernel can do that, from a separate statistics pass otherwise."""
        assert len(pieces) == len(layer.in_layout) and len(pieces) <= _lib.MAXSEG, name
        for p, (c, cp) in zip(pieces, layer.in_layout):
            assert p.Cp == cp and p.C == c, (name, p.C, p.Cp, c, cp)
        if self.bf16:
            return self._conv_bf16(layer, pieces, dst, B, H, W, residual, name, stats, scores)
        Ho, Wo = layer.out_hw(H, W)
        # <= 4 output channels over a narrow raw slice (ESF-Net's logits layer, 32 -> 3 at full resolution): exact fp32 on the vector
        # ALU, weights as scalar operands (conv_narrow_f32.hip) -- the matrix kernels compute a 32-wide output block for it
        if (NARROW_F32 and not self.train and not self.dyn_scales and layer.kh == 3 and layer.kw == 3 and layer.stride == 1 and layer.G == 1
                and layer.pad == (1, 1) and layer.pad_mode == 0 and layer.dils[0] == 1 and layer.Cout <= 4 and len(pieces) == 1
                and not isinstance(pieces[0], PlanarPiece) and pieces[0].scale is None and pieces[0].presplit is None
                and 32 <= pieces[0].Cp <= 64 and residual is None and not stats and scores is None
                and getattr(self, "_pool_req", None) is None and layer.act in (ACT_NONE, ACT_RELU, ACT_LEAKY)
                and H * W * pieces[0].stride < 2 ** 29):
            layer.need_flat = True
            if layer not in self.layers:
                self.layers.append(layer)
            layer.ensure_packed(self.device)
            d = _lib.ConvDesc()
            d.ovf_flag, d.f16_products = self.ovf_ptr(), self.f16_products
            d.B, d.H, d.W, d.Ho, d.Wo = B, H, W, Ho, Wo
            d.kh, d.kw, d.stride, d.pad_h, d.pad_w, d.pad_mode, d.ngroups = 3, 3, 1, 1, 1, 0, 1
            for g in range(_lib.MAXGROUP):
                d.dil[g] = 1
            d.nseg = 1
            p0 = pieces[0]
            sg = d.seg[0]
            sg.ptr, sg.pix_stride, sg.ch_off, sg.Cp, sg.act_in = p0.ptr, p0.stride, p0.off, p0.Cp, p0.act_in
            cp32 = (p0.Cp + 31) // 32 * 32
            wn = self.vec(9 * cp32 * 4)          # [tap][channel][4 outputs]: re-packed when the weight changes

            def repack(layer=layer, wn=wn):
                w = layer.weights[0].detach().to(torch.float32).contiguous()
                _lib.check(self.L.egne_pack_conv3x3_narrow_weight(w.data_ptr(), layer.Cout, layer.Cin, wn.data_ptr(), _lib.stream_ptr()), "pack_conv3x3_narrow")
            self.pre.append(VersionGuard([layer.weights[0]], repack))
            d.Ktot, d.CoutP, d.w = cp32, 4, wn.data_ptr()
            d.bias = layer.bp.data_ptr() if layer.biases is not None else None
            d.act = layer.act
            if layer.post is not None:
                d.post_scale, d.post_shift = layer.post[0].data_ptr(), layer.post[1].data_ptr()
            d.out, d.out_pix_stride, d.out_ch_off = dst.ptr, dst.stride, dst.off
            # the kernel computes four outputs (two packed multiply-adds per operand: no more instructions than three) and stores the
            # slice's padding channels as zeros, like every other convolution of the library (the pack's rows, the bias and the post
            # affine beyond Cout are zero)
            d.Cout_store = min(layer.Cout_store, dst.Cp)
            assert int(self.L.egne_conv3x3_narrow_supported(C.byref(d))) == 1 and p0.act_in == ACT_NONE, name
            self.keep.append(d)
            self._add(self.L.egne_conv3x3_narrow_fwd, (C.byref(d),), name, flops=2.0 * B * H * W * layer.Cout * layer.Cin * 9, kind="conv3x3_narrow")
            return Ho, Wo
        halo = (HALO_ENABLED and layer.kh == 3 and layer.kw == 3 and layer.stride == 1 and layer.G == 1
                and layer.pad == (1, 1) and layer.pad_mode == 0 and len(pieces) == 1 and layer.dils[0] <= 2
                and W >= HALO_MIN_W and layer.CoutP <= HALO_MAX_COUTP and H * W * pieces[0].stride < 2 ** 31)
        smallcin = (SMALLCIN_ENABLED and layer.kh == 3 and layer.kw == 3 and layer.stride == 1 and layer.G == 1
                    and layer.pad == (1, 1) and layer.pad_mode == 0 and len(pieces) == 1 and layer.dils[0] == 1
                    and layer.Cin <= 4 and pad8(layer.Cout) <= 64 and pieces[0].scale is None and residual is None
                    and not isinstance(layer, (DgradLayer, SplitDgradLayer)))      # (a first-writer data gradient has no residual either)
        # frozen nets: streaming split-f16 form (both forms run at the HBM write rate; measured 12 % faster for 64 output
        # channels, 7 % slower for 32)
        c4h = (smallcin and F16X3_ENABLED and (layer.split or getattr(layer, "split_c4", False))
               and (C4H_MODE == "all" or (C4H_MODE == "wide" and pad8(layer.Cout) > 32)))
        split = (F16X3_ENABLED and layer.split and layer.stride == 1 and layer.pad_mode == 0 and len(pieces) == 1
                 and pieces[0].Cp >= 32)
        # narrow 3x3 layers on wide images: split-f16 arithmetic AND the LDS halo (input fetched once for 9 taps)
        shalo = (split and HALO_F16_ENABLED and layer.kh == 3 and layer.kw == 3 and layer.G == 1 and layer.pad == (1, 1)
                 and layer.dils[0] <= 2 and W >= (HALO_F16_MIN_W if layer.CoutP > 64 else HALO_F16_MIN_W_NARROW)
                 and layer.CoutP <= HALO_F16_MAX_COUTP and residual is None
                 and H * W * pieces[0].stride < 2 ** 31)
        # fused dilated group of an MSBlock: three lattice-halo launches (out = o + sum_g relu(conv_g(o)))
        lattice = (split and LATTICE_ENABLED and layer.G == 3 and layer.kh == 3 and layer.kw == 3 and layer.pad == (1, 1)
                   and layer.CoutP in (32, 64) and residual is not None and pieces[0].scale is None
                   and min(W // d_ for d_ in layer.dils) >= LATTICE_MIN_W and H * W * pieces[0].stride < 2 ** 31)
        # 1x1 over raw slices: streaming split-f16 kernel (weights in LDS, operands straight from HBM)
        s1x1 = residual is None and self.stream1x1_ok(layer, pieces, B, H, W)
        # 1x1 over raw slices that the streaming kernel cannot take (K or Cout too large): LDS-staged split-f16 GEMM
        ms1x1 = (F16X3_ENABLED and MS1X1_ENABLED and not s1x1 and layer.split1 and layer.kh == 1 and layer.kw == 1 and layer.stride == 1
                 and layer.G == 1 and layer.pad == (0, 0) and residual is None and layer.post is None
                 and all(pc.scale is None for pc in pieces) and layer.Cout > 32 and B * H * W >= MS1X1_MIN_PIX)
        # wide trunk layers: deep 256-wide split-f16 kernel (weights by LDS-DMA, one barrier per K step)
        big = (split and BIG_ENABLED and layer.G == 1 and pieces[0].scale is None and pieces[0].Cp % 32 == 0 and residual is None
               and layer.post is None and layer.Cout % 128 == 0 and layer.Cout >= BIG_MIN_COUT and layer.Cin >= BIG_MIN_CIN
               and B * Ho * Wo >= 256 * 128)
        # ... or, for the standard dilations, ONE launch with the 4-way sum in registers (msblock_dil_f16.hip)
        msdil = (split and MSDIL_ENABLED and layer.G == 3 and layer.kh == 3 and layer.kw == 3 and layer.pad == (1, 1)
                 and layer.dils == (4, 8, 12) and layer.CoutP == 32 and pieces[0].Cp == 32 and residual is not None
                 and pieces[0].scale is None and layer.act == ACT_RELU and layer.post is None
                 and H * W * pieces[0].stride < 2 ** 29 and H * W * dst.stride < 2 ** 29)
        if msdil:
            lattice = False
        assert pieces[0].presplit is None or msdil if not isinstance(pieces[0], PlanarPiece) else True, "%s: only the one-launch dilated group reads split-pair storage" % name
        if msdil and pieces[0].presplit is not None:
            layer.k_perm = SPLIT_PAIR_PERM
        if self.dyn_scales:     # kernels that read egne_conv_desc.dyn_scale: role-split / resident-weights / halo / flat
            s1x1 = ms1x1 = big = lattice = msdil = c4h = False
        # narrow-input 3x3 layers on wide maps: producer / consumer waves (conv3x3_rs_f16.hip) instead of the all-in-one halo kernel
        rs = (split and RS_ENABLED and HALO_F16_ENABLED and not lattice and not msdil and layer.kh == 3 and layer.kw == 3
              and layer.G == 1 and layer.pad == (1, 1) and layer.stride == 1 and layer.pad_mode == 0 and layer.dils[0] == 1
              and len(pieces) == 1 and W >= (RS_MIN_W_F16 if self.f16_products == 1 else RS_MIN_W) and H * W * max(pieces[0].stride, dst.stride) < 2 ** 29
              and (residual is None or H * W * residual.stride < 2 ** 29))
        # wider inputs: the resident-weights kernel with its weights streamed chunk by chunk (no statistics from its epilogue)
        # (measured against the halo kernel: ahead for 128 -> 32 at 120x160 (290 vs 318 us), level or behind at 60x80 and for wide
        #  outputs -- every output block stages the input again --, so only single-block layers on wide maps take it by default)
        rw_wide = (rs and RW_ENABLED and 64 < pieces[0].Cp <= RW_MAX_CP and layer.sfrag_coutp() <= RW_MAX_COUTP and not stats
                   and W >= (RW_MIN_W_F16 if self.f16_products == 1 else RW_MIN_W) and min(layer.Cout_store, dst.Cp) % 8 == 0 and dst.stride % 4 == 0 and dst.off % 4 == 0
                   and (residual is None or (residual.stride % 4 == 0 and residual.off % 4 == 0)))
        rs = rw_wide or (rs and 8 <= pieces[0].Cp <= 64 and layer.sfrag_coutp() in (32, 64, 128)
                         and not (pieces[0].Cp <= 32 and layer.sfrag_coutp() == 128))
        # a slice whose last 32-channel chunk holds <= 16 channels (38 -> 64 of ESF-Net's second down block): the halo kernel skips the
        # zero half of that chunk's k-steps (3 of 4 for 40 channels), the role-split kernels' 16x16x32 MFMAs cannot
        if rs and not rw_wide and TAIL16_HALO and 32 < pieces[0].Cp <= 64 and 0 < pieces[0].Cp % 32 <= 16:
            rs = False
        shalo = shalo or rs
        if lattice or msdil:
            shalo = True
        if split and not shalo and halo and pieces[0].scale is not None and layer.CoutP <= 32 and W >= HALO_F16_MIN_W:
            split = False            # narrow fused-affine layers: the fp32 halo kernel beats the flat split kernel
        if smallcin:
            split = shalo = False
        # small problems (one or two frames at the deep levels): 64-wide tiles + split-K on the flat kernel (conv_f16x3.hip small_plan)
        small_ws = -1
        if SMALL_ENABLED and split and not self.train and layer.G == 1 and not (lattice or msdil):
            dq = _lib.ConvDesc()
            dq.B, dq.Ho, dq.Wo, dq.kh, dq.kw, dq.ngroups, dq.CoutP = B, Ho, Wo, layer.kh, layer.kw, 1, layer.split_coutp()
            dq.seg[0].Cp = pieces[0].Cp
            small_ws = int(self.L.egne_conv2d_f16x3_small_workspace_floats(C.byref(dq)))
        small = small_ws >= 0
        if small and stats and STATS_FUSED and shalo and dst.Cp == min(layer.Cout_store, dst.Cp):
            tall_ = ((H + 31) // 32) * ((W + 7) // 8) < ((W + 31) // 32) * ((H + 7) // 8)
            if rs or (layer.dils[0] == 1 and not tall_):
                small = False            # the halo / role-split kernel writes the consumer's InstanceNorm sums from its epilogue: one launch
        if small:
            shalo = rs = big = False
        if smallcin or split:
            halo = False
        big_tail = 0     # frames handed to the 128x128 kernel so that the 256x256 launch fills whole rounds of 256 CUs
        if big:
            smallcin = shalo = halo = lattice = s1x1 = False
            layer.need_big = True
            layer.need_flat = True
            big1 = BIG1_ENABLED and self.f16_products == 1 and (pad32(layer.Ktot) // 32 * layer.kh * layer.kw) % 2 == 0
            if big1:
                layer.need_big1 = True
            bn = 256 if layer.Cout % 256 == 0 else 128
            ny = (layer.Cout + bn - 1) // bn
            wgs = lambda nb: (nb * Ho * Wo + 255) // 256 * ny      # noqa: E731
            full = wgs(B) // BIG_CUS
            if BIG_SPLIT_TAIL and full >= 1 and 0.04 < wgs(B) / BIG_CUS - full < 0.65:
                b1 = B
                while b1 > 1 and wgs(b1) > full * BIG_CUS:
                    b1 -= 1
                if wgs(b1) >= 0.9 * full * BIG_CUS:
                    big_tail = B - b1
                    layer.need_split = True
        elif s1x1:
            smallcin = split = shalo = halo = lattice = False
            layer.need_s1 = True
            layer.need_flat = True
        elif ms1x1:
            smallcin = split = shalo = halo = lattice = False
            layer.need_m1 = True
            layer.need_flat = True
        elif smallcin and c4h:
            layer.need_c4h = True
            layer.need_flat = True
        assert not isinstance(pieces[0], PlanarPiece) or (smallcin and c4h), "%s: only the first-layer streaming kernel reads planar inputs" % name
        if False:
            pass
        elif smallcin:
            layer.need_c4 = True
            layer.need_flat = True   # the generic pack is still what the backward (wgrad) paths index with kinv
        elif shalo:
            layer.need_sfrag = True
        elif split and not big:
            layer.need_split = True
        elif halo:
            layer.need_frag = True
        else:
            layer.need_flat = True
        if layer not in self.layers:
            self.layers.append(layer)
        layer.ensure_packed(self.device)
        d = _lib.ConvDesc()
        d.ovf_flag, d.f16_products = self.ovf_ptr(), self.f16_products
        d.B, d.H, d.W, d.Ho, d.Wo = B, H, W, Ho, Wo
        d.kh, d.kw, d.stride = layer.kh, layer.kw, layer.stride
        d.pad_h, d.pad_w, d.pad_mode = layer.pad[0], layer.pad[1], layer.pad_mode
        d.ngroups = layer.G
        for g in range(_lib.MAXGROUP):
            d.dil[g] = layer.dils[g] if g < layer.G else 1
        d.nseg = len(pieces)
        for i, p in enumerate(pieces):
            s = d.seg[i]
            s.ptr, s.pix_stride, s.ch_off, s.Cp = p.ptr, p.stride, p.off, (p.C if isinstance(p, PlanarPiece) else p.Cp)
            s.scale = p.scale.data_ptr() if p.scale is not None else None
            s.shift = p.shift.data_ptr() if p.shift is not None else None
            s.act_in = p.act_in
        d.Ktot, d.CoutP = (pad32(layer.Ktot), layer.split_coutp()) if split else (layer.Ktot, layer.CoutP)
        if big:
            d.CoutP = layer.big_coutp
        if shalo:
            d.CoutP = layer.sfrag_coutp()
        d.w = (layer.wimg.data_ptr() if big else (layer.fhi.data_ptr() if shalo else layer.whi.data_ptr())) if split else (layer.wf.data_ptr() if halo else layer.wp.data_ptr())
        d.bias = layer.bp.data_ptr() if layer.biases is not None else None
        d.act = layer.act
        if layer.post is not None:
            d.post_scale, d.post_shift = layer.post[0].data_ptr(), layer.post[1].data_ptr()
        if residual is not None:
            d.residual, d.res_pix_stride, d.res_ch_off = residual.ptr, residual.stride, residual.off
        d.out, d.out_pix_stride, d.out_ch_off = dst.ptr, dst.stride, dst.off
        d.Cout_store = min(layer.Cout_store, dst.Cp)
        assert tuple(dst.buf.shape[1:3]) == (Ho, Wo), (name, tuple(dst.buf.shape), Ho, Wo)
        self.keep.append(d)
        flops = 2.0 * B * Ho * Wo * layer.Cout * layer.Cin * layer.kh * layer.kw * layer.G
        # split-f16 kernels multiply their input by a power of two before the f16 split; for RAW inputs (no normalisation
        # fused on load) that factor comes from the measured max |x| of the slices (calibration pass of Plan.run)
        raw = all(pc.scale is None for pc in pieces)
        cal3 = (3, list(pieces), B * H * W) if raw else None
        cal2 = (2, list(pieces), B * H * W) if raw else None
        if self.dyn_scales:
            layer.stale_scale_ok = True
        if self.dyn_scales and raw and split and not (smallcin and c4h):
            self._dyn_slot(d, pieces, B, B * H * W, name)
            cal3 = cal2 = None
        any16 = getattr(pieces[0], "f16s", None) is not None or dst.f16s is not None
        if any16 and big and big1:
            big_tail = 0                      # (the flat kernel behind a ragged last round reads and writes fp32)
        if any16 and not ((shalo and rs and not (big or ms1x1 or s1x1 or msdil or lattice)) or (smallcin and c4h and not big) or (big and big1)):
            raise NeedsFp32Storage(name)      # only the resident-weights 3x3, the first-layer kernel and the plain-f16 deep trunk kernel know f16 storage
        if big and big_tail:
            # two launches over disjoint frame ranges: [0, B - tail) on the 256-wide kernel, the rest on the 128x128 kernel
            layer.ensure_packed(self.device)
            d2 = _lib.ConvDesc()
            C.memmove(C.byref(d2), C.byref(d), C.sizeof(_lib.ConvDesc))
            b1 = B - big_tail
            d.B, d2.B = b1, big_tail
            p0 = pieces[0]
            d2.seg[0].ptr = p0.ptr + 4 * b1 * H * W * p0.stride
            d2.out = dst.ptr + 4 * b1 * Ho * Wo * dst.stride
            d2.Ktot, d2.CoutP = pad32(layer.Ktot), layer.split_coutp()
            self.keep.append(d2)
            self._add(self.L.egne_conv2d_f16_big1_fwd if big1 else self.L.egne_conv2d_f16x3_big_fwd,
                      (C.byref(d), (layer.wimg1 if big1 else layer.wimg).data_ptr(), F16X3_ASCALE, layer.w_scale_big), name,
                      flops=flops * b1 / B, kind="conv_f16x3:big", cal=cal2, ws=[(3, layer, "w_scale_big")])
            self._add(self.L.egne_conv2d_f16x3_fwd, (C.byref(d2), layer.whi.data_ptr(), layer.wlo.data_ptr(), F16X3_ASCALE, layer.w_scale),
                      name + ".tail", flops=flops * big_tail / B, kind="conv_f16x3:flat", cal=cal3, ws=[(4, layer, "w_scale")])
        elif big:
            if any16:
                if not (CALIBRATE and not self.dyn_scales and raw):
                    raise NeedsFp32Storage(name)
                if dst.f16s is not None:
                    cal2 = self._f16_out(d, layer, dst, B * Ho * Wo, cal2, pieces[0].f16s)
                if pieces[0].f16s is not None:
                    cal2 = self._f16_in(d, pieces[0].f16s, cal2, 2)
            self._add(self.L.egne_conv2d_f16_big1_fwd if big1 else self.L.egne_conv2d_f16x3_big_fwd,
                      (C.byref(d), (layer.wimg1 if big1 else layer.wimg).data_ptr(), F16X3_ASCALE, layer.w_scale_big), name,
                      flops=flops, kind="conv_f16x3:big", cal=cal2, ws=[(3, layer, "w_scale_big")])
        elif ms1x1:
            d.Ktot, d.CoutP = layer.m1_ktot, layer.m1_coutp
            self._add(self.L.egne_conv1x1_ms_f16x3_fwd, (C.byref(d), layer.m1hi.data_ptr(), layer.m1lo.data_ptr(), F16X3_ASCALE,
                                                         layer.w_scale_m1), name, flops=flops, kind="conv_f16x3:gemm1x1", cal=cal3, ws=[(4, layer, "w_scale_m1")])
        elif s1x1:
            if getattr(self, "_up_add", None) is not None:
                P, ph, pw = self._up_add
                assert (2 * ph, 2 * pw) == (H, W) and layer.act == ACT_NONE and P.Cp >= int(d.Cout_store) and tuple(P.buf.shape[:3]) == (B, ph, pw), name
                d.Ho, d.Wo = ph, pw
                d.residual, d.res_pix_stride, d.res_ch_off = P.ptr, P.stride, P.off
            self._add(self.L.egne_conv1x1_f16x3_fwd, (C.byref(d), layer.s1hi.data_ptr(), layer.s1lo.data_ptr(), F16X3_ASCALE,
                                                      layer.w_scale1), name, flops=flops, kind="conv_f16x3:stream1x1", cal=cal3, ws=[(4, layer, "w_scale1")])
        elif msdil and scores is not None:
            # the block's output is consumed by the stage's score heads only: they are evaluated in the epilogue and the
            # 32-channel map is never stored (scores = (weights [2][32], constants [2], s, s1, accumulate))
            cw_, cc_, s0_, s1_, accum = scores
            d.out = None
            cal3 = self._presplit_in(d, pieces, residual, cal3)
            self._add(self.L.egne_msblock_dil_scores_f16_fwd, (C.byref(d), layer.fhi.data_ptr(), layer.flo.data_ptr(), F16X3_ASCALE,
                                                               layer.w_scale, cw_.data_ptr(), cc_.data_ptr(), s0_.data_ptr(),
                                                               s1_.data_ptr(), int(accum)), name, flops=flops,
                      kind="conv_f16x3:msdil", cal=cal3, ws=[(4, layer, "w_scale")])
        elif msdil:
            cal3 = self._presplit_in(d, pieces, residual, cal3)
            self._add(self.L.egne_msblock_dil_f16_fwd, (C.byref(d), layer.fhi.data_ptr(), layer.flo.data_ptr(), F16X3_ASCALE,
                                                        layer.w_scale), name, flops=flops, kind="conv_f16x3:msdil", cal=cal3, ws=[(4, layer, "w_scale")])
        elif lattice:
            perf = 9 * layer.sfrag_coutp() * pad32(layer.Ktot)
            for g in range(3):
                dg = _lib.ConvDesc()
                C.memmove(C.byref(dg), C.byref(d), C.sizeof(_lib.ConvDesc))
                dg.ngroups = 1
                dg.dil[0] = layer.dils[g]
                dg.bias = layer.bp.data_ptr() + 4 * g * layer.CoutP if layer.biases is not None else None
                if g > 0:   # accumulate onto the running sum
                    dg.residual, dg.res_pix_stride, dg.res_ch_off = dst.ptr, dst.stride, dst.off
                self.keep.append(dg)
                self._add(self.L.egne_conv3x3_halo_f16_fwd, (C.byref(dg), layer.fhi.data_ptr() + 2 * g * perf,
                                                             layer.flo.data_ptr() + 2 * g * perf, F16X3_ASCALE, layer.w_scale),
                          name + ".g%d" % g, flops=flops / 3, kind="conv_f16x3:lattice", cal=cal3, ws=[(4, layer, "w_scale")])
        elif shalo and rs:
            pq = getattr(self, "_pool_req", None)
            if (pq is not None and POOL_FUSED and pieces[0].Cp > 32 and layer.sfrag_coutp() >= 64 and layer.post is None
                    and layer.act in (ACT_NONE, ACT_RELU, ACT_LEAKY) and pq.Cp >= int(d.Cout_store)):
                d.pool_out, d.pool_pix_stride, d.pool_ch_off = pq.ptr, pq.stride, pq.off
                self.last_pooled = True
            th = 8 if pieces[0].Cp <= 32 else 4
            nchunk = ((W + 31) // 32) * ((H + th - 1) // th) * (2 if (th == 4 and layer.sfrag_coutp() >= 64) else 4)
            fuse_stats = stats and STATS_FUSED and dst.Cp == int(d.Cout_store)
            # resident-weights form (conv3x3_rw_f16.hip) unless the layer writes InstanceNorm sums from its epilogue
            # (measured at 240x320x64 against the register-ring form: 64 -> 32 566 vs 777 us, 64 -> 64 1070 vs 1170, 32 -> 32 318 vs 356,
            #  32 -> 64 622 vs 740; 64 -> 128 at 120x160 525 vs 560)
            rw = rw_wide or (RW_ENABLED and not fuse_stats
                  and int(d.Cout_store) % 8 == 0 and dst.stride % 4 == 0 and dst.off % 4 == 0
                  and (residual is None or (residual.stride % 4 == 0 and residual.off % 4 == 0)))
            if fuse_stats:
                ws = self._stats_ws(d, B, nchunk)
            ps = dst.presplit
            if (ps is not None and rw and raw and CALIBRATE and not self.dyn_scales and layer.post is None and residual is None
                    and not self.last_pooled and int(d.Cout_store) % 32 == 0 and dst.off % 32 == 0):
                # split-pair output: hi / lo halves of out * s, s from a bound on |out| taken when the launch is calibrated
                d.out_split, d.out_split_scale = 1, ps.value
                self.last_presplit = True

                def cal_ps(args, vmax, d=d, layer=layer, ps=ps):
                    with torch.no_grad():
                        bound = vmax * float(layer.weights[0].detach().abs().sum(dim=(1, 2, 3)).max())
                        if layer.biases is not None and layer.biases[0] is not None:
                            bound += float(layer.biases[0].detach().abs().max())
                    ps.value = d.out_split_scale = _a_scale_for(bound)
                    return args[:3] + (_a_scale_for(vmax),) + args[4:]
                cal3 = (cal_ps, list(pieces), B * H * W)
            fin, fout = pieces[0].f16s, dst.f16s
            if fin is not None or fout is not None:
                if not (rw and self.f16_products == 1 and CALIBRATE and not self.dyn_scales and layer.post is None and residual is None and raw):
                    raise NeedsFp32Storage(name)
                if fout is not None:
                    if d.pool_out:
                        assert pq.f16s is fout, "%s: the pooled second output shares the main output's storage scale" % name
                    cal3 = self._f16_out(d, layer, dst, B * Ho * Wo, cal3, fin)
                elif d.pool_out:
                    raise NeedsFp32Storage(name)
                if fin is not None:
                    cal3 = self._f16_in(d, fin, cal3, 3)
            self._add(self.L.egne_conv3x3_rw_f16_fwd if rw else self.L.egne_conv3x3_rs_f16_fwd,
                      (C.byref(d), layer.fhi.data_ptr(), layer.flo.data_ptr(), F16X3_ASCALE, layer.w_scale), name, flops=flops,
                      kind="conv_f16x3:rw" if rw else "conv_f16x3:rs", cal=cal3, ws=[(4, layer, "w_scale")])
            if fuse_stats:
                self.last_stats = self._stats_finish(ws, d, B, H * W, nchunk, name)
                stats = False
        elif shalo:
            tx, ty = (W + 31) // 32, (H + 7) // 8
            tall = ((H + 31) // 32) * ((W + 7) // 8) < tx * ty          # the kernel would walk the map transposed
            fuse_stats = stats and STATS_FUSED and layer.dils[0] == 1 and dst.Cp == int(d.Cout_store)
            if tall and HALO_TALL:                 # the launcher's choice: the tiles (= statistics chunks) of the transposed walk
                tx, ty = (H + 31) // 32, (W + 7) // 8
            pq = getattr(self, "_pool_req", None)
            if (pq is not None and POOL_FUSED and HALO_POOL and not lattice and not stats and layer.dils[0] == 1 and layer.post is None
                    and residual is None and layer.act in (ACT_NONE, ACT_RELU, ACT_LEAKY) and pq.Cp >= int(d.Cout_store) and layer.sfrag_coutp() % 64 == 0
                    and ((H + 1) // 2) * ((W + 1) // 2) * pq.stride < 2 ** 29):
                d.pool_out, d.pool_pix_stride, d.pool_ch_off = pq.ptr, pq.stride, pq.off      # pool2 of vgg16_c.py:72 from conv2_2's epilogue
                self.last_pooled = True
            if fuse_stats:
                ws = self._stats_ws(d, B, tx * ty * 4)
            self._add(self.L.egne_conv3x3_halo_f16_fwd, (C.byref(d), layer.fhi.data_ptr(), layer.flo.data_ptr(), F16X3_ASCALE,
                                                         layer.w_scale), name, flops=flops, kind="conv_f16x3:halo", cal=cal3, ws=[(4, layer, "w_scale")])
            if fuse_stats:
                self.last_stats = self._stats_finish(ws, d, B, H * W, tx * ty * 4, name)
                stats = False
        elif split and small:
            wsb = self.vec(max(small_ws, 4))
            self._add(self.L.egne_conv2d_f16x3_small_fwd, (C.byref(d), layer.whi.data_ptr(), layer.wlo.data_ptr(), F16X3_ASCALE,
                                                           layer.w_scale, wsb.data_ptr(), small_ws), name, flops=flops,
                      kind="conv_f16x3:small", cal=cal3, ws=[(4, layer, "w_scale")])
        elif split:
            self._add(self.L.egne_conv2d_f16x3_fwd, (C.byref(d), layer.whi.data_ptr(), layer.wlo.data_ptr(), F16X3_ASCALE,
                                                     layer.w_scale), name, flops=flops, kind="conv_f16x3:flat", cal=cal3, ws=[(4, layer, "w_scale")])
        elif smallcin and c4h:
            d.CoutP = layer.c4_coutp
            if dst.f16s is not None:
                cal3 = self._f16_out(d, layer, dst, B * Ho * Wo, cal3, None)
            self._add(self.L.egne_conv3x3_smallcin_f16_fwd, (C.byref(d), layer.c4hi.data_ptr(), layer.c4lo.data_ptr(), F16X3_ASCALE,
                                                             layer.w_scale_c4), name, flops=flops, kind="conv_f16x3:first", cal=cal3, ws=[(4, layer, "w_scale_c4")])
        elif smallcin:
            self._add(self.L.egne_conv3x3_smallcin_fwd, (C.byref(d), layer.w40.data_ptr()), name, flops=flops, kind="conv3x3_smallcin")
        elif halo:
            self._add(self.L.egne_conv3x3_halo_fwd, (C.byref(d),), name, flops=flops, kind="conv3x3_halo")
        else:
            if self.dyn_scales and residual is None and int(d.Cout_store) >= dst.Cp:
                d.absmax_out = self._publish_absmax(dst, B)       # a split-f16 3x3 usually reads this next
            self._add(self.L.egne_conv2d_fwd, (C.byref(d),), name, flops=flops, kind="conv_igemm")
        assert scores is None or msdil, "score fusion needs the one-launch MSBlock kernel (check Plan.msdil_ok first)"
        if stats:
            self.last_stats = self.norm_stats(dst, B, Ho * Wo, name=name + ".stats")[:2]
        if self.train:
            # the backward kernels index the exact-fp32 weight layout: undo what a split-f16 forward put into the descriptor
            db = _lib.ConvDesc()
            C.memmove(C.byref(db), C.byref(d), C.sizeof(_lib.ConvDesc))
            db.B, db.Ktot, db.CoutP = B, layer.Ktot, layer.CoutP
            db.seg[0].ptr, db.out = pieces[0].ptr, dst.ptr
            self.keep.append(db)
            self.tape.append(lambda bw: self._bw_conv(bw, layer, list(pieces), dst, db, B, H, W, Ho, Wo, name))
        return Ho, Wo

    def _f16_out(self, d, layer, dst, npix, cal, fin):
        """dst is held as f16 (Piece.f16s): out_split = 2.  The calibrating run first stores with a scale from a BOUND on |out| (max |in|
        * max_co sum |w| + max |b|: nothing can overflow), then Plan._run_calibrating measures the stored tensor, sets the scale that
        puts its maximum in [1024, 2048) and runs the launch again (post_cal)."""
        fs = dst.f16s
        assert self.f16_products == 1 and CALIBRATE and not self.dyn_scales and cal is not None, "f16 storage needs a calibrated plain-f16 plan"
        d.out_split, d.out_split_scale = 2, fs.value
        ai, pcs, npx = cal

        def cal_out(args, vmax, d=d, layer=layer, fs=fs, ai=ai, fin=fin):
            if fin is not None and not vmax:
                vmax = getattr(fin, "vmax", None) or 2047.0 / fin.value
            with torch.no_grad():
                bound = vmax * float(layer.weights[0].detach().abs().sum(dim=(1, 2, 3)).max())
                if layer.biases is not None and layer.biases[0] is not None:
                    bound += float(layer.biases[0].detach().abs().max())
            fs.value = d.out_split_scale = _a_scale_for(bound)
            return ai(args, vmax) if callable(ai) else args[:ai] + (_a_scale_for(vmax),) + args[ai + 1:]
        self.post_cal[len(self.calls)] = (d, dst, fs, npix)
        return (cal_out, pcs, npx)

    def _f16_in(self, d, fin, cal, ai):
        """pieces[0] is held as f16 (Piece.f16s = fin): presplit = 2 and the launch's a_scale argument (index ``ai``) IS the storage scale --
        nothing is measured; an inner calibration callback (the output's bound) starts from the maximum measured when the tensor's own scale
        was calibrated (SplitScale.vmax)."""
        d.seg[0].presplit = 2
        inner = cal[0] if (cal is not None and callable(cal[0])) else None

        def cal_in(args, vmax, fin=fin, inner=inner, ai=ai):
            if inner is not None:
                args = inner(args, getattr(fin, "vmax", None) or 2047.0 / fin.value)
            return args[:ai] + (fin.value,) + args[ai + 1:]
        return (cal_in, [], 0)

    def _presplit_in(self, d, pieces, residual, cal):
        """One-launch dilated group whose input was written in split-pair storage: flag the slice and take the launch's pre-scale
        from the producer's SplitScale when the plan is calibrated (the producer's launch comes first)."""
        ps = pieces[0].presplit
        if ps is None:
            return cal
        assert residual is not None and residual.buf is pieces[0].buf and residual.off == pieces[0].off, "split-pair input must be the residual too"
        d.seg[0].presplit = 1
        return (lambda args, vmax, ps=ps: args[:3] + (ps.value,) + args[4:], [], 0)

    def _conv_bf16(self, layer, pieces, dst, B, H, W, residual, name, stats, scores):
        """Convolutions of a plan with bf16 activation storage: the 3x3 "same" convolutions over one slice run on bf16 MFMAs
        (conv3x3_bf16.hip: weights rounded to bf16, fp32 accumulate), first layers (Cin <= 4) on the taps-in-K kernel and
        everything else (1x1 over several slices, strided / reflect-padded / valid convolutions, linear layers) on the exact-fp32
        implicit GEMM reading and writing bf16 (egne_conv_desc.dtype = 1).  No pre-scales, no calibration: bf16 has fp32's range."""
        assert scores is None and layer.G == 1 and not isinstance(pieces[0], PlanarPiece), name
        Ho, Wo = layer.out_hw(H, W)
        one = len(pieces) == 1 and layer.stride == 1 and layer.pad_mode == 0 and layer.dils[0] == 1 and layer.kh == 3 and layer.kw == 3 \
            and layer.pad == (1, 1)
        smallcin = (one and SMALLCIN_ENABLED and layer.Cin <= 4 and pad8(layer.Cout) <= 64 and pieces[0].scale is None and residual is None
                    and not isinstance(layer, (DgradLayer, SplitDgradLayer, BfDgradLayer)) and layer.post is None)
        fast3 = (one and not smallcin and BF16_FAST3X3 and layer.post is None and pieces[0].Cp % 8 == 0 and pieces[0].off % 8 == 0 and pieces[0].stride % 8 == 0
                 and layer.CoutP <= 256 and min(layer.Cout_store, dst.Cp) % 4 == 0 and min(layer.Cout_store, dst.Cp) >= 8
                 and H * W * max(pieces[0].stride, dst.stride) < 2 ** 30
                 and (residual is None or H * W * residual.stride < 2 ** 30))
        # 1x1 over raw slices: streaming bf16-MFMA kernel (conv1x1_bf16.hip); its weights must fit LDS (k-steps x blocks x 1 KB)
        fast1 = (BF16_FAST1X1 and layer.kh == 1 and layer.kw == 1 and layer.stride == 1 and layer.pad == (0, 0) and layer.post is None
                 and all(p.scale is None and p.Cp % 8 == 0 and p.off % 8 == 0 and p.stride % 8 == 0 for p in pieces)
                 and B * H * W >= 4096 and min(layer.Cout_store, dst.Cp) % 8 == 0 and dst.off % 8 == 0 and dst.stride % 8 == 0
                 and (residual is None or (residual.off % 8 == 0 and residual.stride % 8 == 0)))
        if fast1:
            dq = _lib.ConvDesc()
            dq.nseg, dq.CoutP, dq.Ktot = len(pieces), layer.CoutP, layer.Ktot
            for i, p in enumerate(pieces):
                dq.seg[i].Cp = p.Cp
            fast1 = int(self.L.egne_conv1x1_bf16_pack_elems(C.byref(dq))) > 0
        if smallcin:
            layer.need_c4 = True
            layer.need_flat = True
        elif fast3:
            layer.need_bfrag = True
            if self.train:
                layer.need_flat = True       # kinv / the generic pack index the weight-gradient paths
        elif fast1:
            layer.need_flat = True
            layer.need_b1 = True
        else:
            layer.need_flat = True
        up_add = getattr(self, "_up_add", None)
        assert up_add is None or (fast1 and residual is None and layer.act == ACT_NONE), "%s: an up-sampled addend needs the streaming bf16 1x1 kernel" % name
        if layer not in self.layers:
            self.layers.append(layer)
        layer.ensure_packed(self.device)
        d = _lib.ConvDesc()
        d.dtype = 1
        d.B, d.H, d.W, d.Ho, d.Wo = B, H, W, Ho, Wo
        d.kh, d.kw, d.stride = layer.kh, layer.kw, layer.stride
        d.pad_h, d.pad_w, d.pad_mode = layer.pad[0], layer.pad[1], layer.pad_mode
        d.ngroups = 1
        for g in range(_lib.MAXGROUP):
            d.dil[g] = layer.dils[0] if g == 0 else 1
        d.nseg = len(pieces)
        for i, p in enumerate(pieces):
            sg = d.seg[i]
            sg.ptr, sg.pix_stride, sg.ch_off, sg.Cp = p.ptr, p.stride, p.off, p.Cp
            sg.scale = p.scale.data_ptr() if p.scale is not None else None
            sg.shift = p.shift.data_ptr() if p.shift is not None else None
            sg.act_in = p.act_in
        d.Ktot, d.CoutP = (pad32(layer.Ktot) if fast3 else layer.Ktot), layer.CoutP
        d.w = None if fast3 else layer.wp.data_ptr()
        d.bias = layer.bp.data_ptr() if layer.biases is not None else None
        d.act = layer.act
        if layer.post is not None:
            d.post_scale, d.post_shift = layer.post[0].data_ptr(), layer.post[1].data_ptr()
        if residual is not None:
            d.residual, d.res_pix_stride, d.res_ch_off = residual.ptr, residual.stride, residual.off
        d.out, d.out_pix_stride, d.out_ch_off = dst.ptr, dst.stride, dst.off
        d.Cout_store = min(layer.Cout_store, dst.Cp)
        assert tuple(dst.buf.shape[1:3]) == (Ho, Wo), (name, tuple(dst.buf.shape), Ho, Wo)
        assert dst.buf.dtype == torch.bfloat16 and all(p.buf.dtype == torch.bfloat16 for p in pieces), name
        self.keep.append(d)
        flops = 2.0 * B * Ho * Wo * layer.Cout * layer.Cin * layer.kh * layer.kw
        db = None
        if self.train:          # the backward descriptor: the plain convolution (an up-sampled addend has a backward of its own below)
            db = _lib.ConvDesc()
            C.memmove(C.byref(db), C.byref(d), C.sizeof(_lib.ConvDesc))
            db.Ktot, db.CoutP = layer.Ktot, layer.CoutP
            self.keep.append(db)
        # statistics of the consumer's normalisation from this launch's epilogue (bf16 3x3: per-(tile, consumer wave) partial sums of what is
        # stored, egne_conv_desc.stats_ws): `stats` (InstanceNorm: finished per sample below) or the caller's `_want_partials` (BatchNorm:
        # the caller finishes over its sample range, esf_engine._train_bn)
        ws_stats, want_partials = None, bool(getattr(self, "_want_partials", False))
        self._want_partials, self.last_partials = False, None
        if (stats or want_partials) and fast3 and STATS_FUSED and STATS_FUSED_BF16 and self.train:
            nchunk_s = ((W + 31) // 32) * ((H + 7) // 8) * 4
            ws_stats = self._stats_ws(d, B, nchunk_s)
            if want_partials:
                self.last_partials = (ws_stats, nchunk_s, int(d.Cout_store))
        if up_add is not None:
            # dst = conv1x1(pieces) + b + up2x(P): P [B][H/2][W/2] in bf16, egne_conv1x1_bf16_fwd's half-resolution "residual"
            P, ph, pw = up_add
            assert (2 * ph, 2 * pw) == (H, W) and P.Cp >= int(d.Cout_store) and tuple(P.buf.shape[:3]) == (B, ph, pw) and P.off % 8 == 0 and P.stride % 8 == 0, name
            d.Ho, d.Wo = ph, pw
            d.residual, d.res_pix_stride, d.res_ch_off = P.ptr, P.stride, P.off
        if smallcin:
            self._add(self.L.egne_conv3x3_smallcin_fwd, (C.byref(d), layer.w40.data_ptr()), name, flops=flops, kind="conv3x3_smallcin")
        elif fast3:
            self._add(self.L.egne_conv3x3_bf16_fwd, (C.byref(d), layer.bfrag.data_ptr()), name, flops=flops, kind="conv_bf16:3x3")
            self._last_b3_desc = d               # (a data gradient may get its slice's mask later: _bw_conv)
        elif fast1:
            self._add(self.L.egne_conv1x1_bf16_fwd, (C.byref(d), layer.b1frag.data_ptr()), name, flops=flops, kind="conv_bf16:1x1")
        elif BF16_NARROW and int(self.L.egne_conv_narrow_bf16_supported(C.byref(d))):
            # k x k onto <= 8 channels (the data gradient of the StyleEncoder's 7x7): LDS-halo kernel instead of 98 implicit-GEMM steps
            self._add(self.L.egne_conv_narrow_bf16_fwd, (C.byref(d),), name, flops=flops, kind="conv_bf16:narrow")
        else:
            self._add(self.L.egne_conv2d_fwd, (C.byref(d),), name, flops=flops, kind="conv_igemm")
        if stats:
            self.last_stats = (self._stats_finish(ws_stats, d, B, Ho * Wo, nchunk_s, name) if ws_stats is not None
                               else self.norm_stats(dst, B, Ho * Wo, name=name + ".stats")[:2])
        if self.train:
            def emit(bw, up_add=up_add):
                self._bw_conv(bw, layer, list(pieces), dst, db, B, H, W, Ho, Wo, name)
                if up_add is not None:
                    # gP (+)= up2x^T(gz): gz = the gradient of dst as _bw_conv left it (the layer has no activation); the first of P's
                    # readers to come by stores (no zero pass for P's twin)
                    P, ph, pw = up_add
                    Pq = Piece(P.buf, P.off, min(P.C, int(db.Cout_store)), int(db.Cout_store), P.n0)
                    first = self.first_touch(Pq.buf, Pq.off, Pq.Cp, Pq.n0, B)
                    gz, gP = self.gp(dst, B), self.gp(Pq, B)
                    if first:
                        self.mark_stored(Pq, B)
                    bw.raw(self.L.egne_upsample2x_bwd_store if first else self.L.egne_upsample2x_bwd,
                           (gz.ptr, gz.stride, gz.off, gP.ptr, gP.stride, gP.off, B, ph, pw, Pq.Cp), name + ".up_add.bwd")
            self.tape.append(emit)
        return Ho, Wo

    def bf16_stream1x1_ok(self, layer, pieces, dst, B, H, W):
        """True if _conv_bf16 runs this 1x1 on the streaming bf16-MFMA kernel (conv1x1_bf16.hip) -- the one that can add an
        up-sampled half-resolution tensor in its epilogue (``up_add``)."""
        if not (self.bf16 and BF16_FAST1X1 and layer.kh == 1 and layer.kw == 1 and layer.stride == 1 and layer.pad == (0, 0) and layer.post is None
                and all(p.scale is None and p.Cp % 8 == 0 and p.off % 8 == 0 and p.stride % 8 == 0 for p in pieces)
                and B * H * W >= 4096 and min(layer.Cout_store, dst.Cp) % 8 == 0 and dst.off % 8 == 0 and dst.stride % 8 == 0
                and len(pieces) <= _lib.MAXSEG):
            return False
        dq = _lib.ConvDesc()
        dq.nseg, dq.CoutP, dq.Ktot = len(pieces), layer.CoutP, layer.Ktot
        for i, p in enumerate(pieces):
            dq.seg[i].Cp = p.Cp
        return int(self.L.egne_conv1x1_bf16_pack_elems(C.byref(dq))) > 0

    def pair_fusable(self, l1, pieces, l2, dst, H, W):
        """True if conv_pair will run l2(l1(cat(pieces))) as ONE launch (conv_fused_1x1_3x3_f16.hip)."""
        return (FUSE_1X1 and F16X3_ENABLED and not self.train and not self.bf16 and l1.split1 and l2.split and l1.kh == 1 and l1.kw == 1
                and l1.stride == 1 and l1.G == 1 and l1.pad == (0, 0) and l1.act == ACT_NONE and l1.post is None
                and all(pc.scale is None for pc in pieces) and l1.CoutP in (32, 64) and len(pieces) <= _lib.MAXSEG
                and sum((pc.Cp + 15) // 16 for pc in pieces) <= 12
                and l2.kh == 3 and l2.kw == 3 and l2.stride == 1 and l2.G == 1 and l2.pad == (1, 1) and l2.dils[0] == 1
                and l2.pad_mode == 0 and len(l2.in_layout) == 1 and l2.in_layout[0][0] == l1.Cout and l2.CoutP in (32, 64)
                and W >= FUSE_1X1_MIN_W and all(H * W * pc.stride < 2 ** 29 for pc in pieces) and H * W * dst.stride < 2 ** 29)

    def td_pool_fusable(self, layer, pieces, dst):
        """Transition_down of an inference plan as ONE launch (conv1x1_pool_f16x3_kernel): 1x1 over normalised slices with the 2x2
        average folded in front of it."""
        G = sum((p.Cp + 15) // 16 for p in pieces)
        return (TDPOOL_FUSED and F16X3_ENABLED and not self.train and not self.bf16 and layer.split1 and layer.kh == 1 and layer.kw == 1 and layer.stride == 1
                and layer.G == 1 and layer.post is None and all(p.scale is not None for p in pieces) and len(pieces) <= _lib.MAXSEG
                and layer.CoutP <= 96 and G * (layer.CoutP // 32) * 2048 <= 80 * 1024 and dst.Cp % 4 == 0)

    def conv1x1_pooled(self, layer, pieces, dst, B, H, W, name="td"):
        """dst[B][H/2][W/2] = avg_pool2d(conv1x1(act_in(pieces * scale + shift)), 2) (models/RITnet_v2.py:32-44), one launch."""
        assert self.td_pool_fusable(layer, pieces, dst), name
        Ho, Wo = H // 2, W // 2
        layer.need_s1 = layer.need_flat = True
        if layer not in self.layers:
            self.layers.append(layer)
        layer.ensure_packed(self.device)
        d = _lib.ConvDesc()
        d.ovf_flag, d.f16_products = self.ovf_ptr(), self.f16_products
        d.B, d.H, d.W, d.Ho, d.Wo = B, H, W, Ho, Wo
        d.kh = d.kw = d.stride = d.ngroups = 1
        d.nseg = len(pieces)
        for i, p in enumerate(pieces):
            sg = d.seg[i]
            sg.ptr, sg.pix_stride, sg.ch_off, sg.Cp = p.ptr, p.stride, p.off, p.Cp
            sg.scale, sg.shift, sg.act_in = p.scale.data_ptr(), p.shift.data_ptr(), p.act_in
        d.Ktot, d.CoutP = layer.Ktot, layer.CoutP
        d.bias = layer.bp.data_ptr() if layer.biases is not None else None
        d.act = layer.act
        d.out, d.out_pix_stride, d.out_ch_off = dst.ptr, dst.stride, dst.off
        d.Cout_store = min(layer.Cout_store, dst.Cp)
        assert tuple(dst.buf.shape[1:3]) == (Ho, Wo), (name, tuple(dst.buf.shape), Ho, Wo)
        self.keep.append(d)
        flops = 2.0 * B * Ho * Wo * layer.Cout * layer.Cin
        # normalised operands: the fixed pre-scale of the other fused-affine layers (|x| < 4094 after the InstanceNorm affine)
        self._add(self.L.egne_conv1x1_pool2_f16x3_fwd, (C.byref(d), layer.s1hi.data_ptr(), layer.s1lo.data_ptr(), F16X3_ASCALE, layer.w_scale1),
                  name, flops=flops, kind="conv_f16x3:tdpool1x1", ws=[(4, layer, "w_scale1")])
        LAYER_BYTES[name] = 4.0 * B * (H * W * sum(p.Cp for p in pieces) + Ho * Wo * int(d.Cout_store))
        return Ho, Wo

    def conv_pair(self, l1, pieces, l2, dst, B, H, W, tmp=None, residual=None, name="pair", stats=False, up_add=None):
        r = self._conv_pair_impl(l1, pieces, l2, dst, B, H, W, tmp, residual, name, stats, up_add)
        LAYER_BYTES[name] = 4.0 * B * H * W * (sum(p.Cp for p in pieces) + min(l2.Cout_store, dst.Cp) + (residual.Cp if residual is not None else 0)
                                               + (8 if up_add is not None else 0))
        return r

    def _conv_pair_impl(self, l1, pieces, l2, dst, B, H, W, tmp=None, residual=None, name="pair", stats=False, up_add=None):
        """``l2(l1(cat(pieces)))``: a 1x1 convolution over raw slices followed by the 3x3 that is its only consumer
        (RITnet_v2.py:59-62,84-87).  Inference plans run the pair as ONE launch whose intermediate stays in LDS
        (conv_fused_1x1_3x3_f16.hip); otherwise two launches through ``tmp`` (a Piece, allocated here if None)."""
        fused = self.pair_fusable(l1, pieces, l2, dst, H, W)
        # up_add = (P, ph, pw): a half-resolution tensor whose bilinear x2 upsampling is added to the 1x1 result (the up-sampled
        # operand of an up block folded through the 1x1; only the fused kernel does this -- callers check pair_fusable first)
        # (bf16-storage plans: the streaming bf16 1x1 adds it in its epilogue, conv1x1_bf16.hip, two launches)
        assert up_add is None or self.bf16 or (fused and l1.CoutP == 32 and l2.CoutP == 32 and sum((pc.Cp + 15) // 16 for pc in pieces) <= 8), name
        # convBlock (utils.py:1047-1048): a 3x3 on <= 4 input channels in front of the 3x3 -- same kernel, taps folded into K
        fused_c4 = (FUSE_1X1 and FUSE_C4 and F16X3_ENABLED and not self.train and not self.bf16 and l1.split and l2.split and l1.kh == 3 and l1.kw == 3
                    and l1.stride == 1 and l1.G == 1 and l1.pad == (1, 1) and l1.pad_mode == 0 and l1.dils[0] == 1 and l1.Cin <= 4
                    and l1.post is None and len(pieces) == 1 and pieces[0].scale is None and pieces[0].Cp >= 4 and l1.CoutP == 32
                    and l2.kh == 3 and l2.kw == 3 and l2.stride == 1 and l2.G == 1 and l2.pad == (1, 1) and l2.dils[0] == 1
                    and l2.pad_mode == 0 and len(l2.in_layout) == 1 and l2.in_layout[0][0] == l1.Cout and l2.CoutP == 32
                    and W >= FUSE_1X1_MIN_W and H * W * pieces[0].stride < 2 ** 29 and H * W * dst.stride < 2 ** 29
                    and (not isinstance(pieces[0], PlanarPiece) or l1.Cin == 1))
        assert fused_c4 or not isinstance(pieces[0], PlanarPiece), "%s: only the fused convBlock kernel reads planar inputs" % name
        if fused_c4:
            return self._conv_pair_c4(l1, pieces[0], l2, dst, B, H, W, residual, name, stats)
        if not fused:
            if tmp is None:
                tmp = Piece(self.buf(B, H, W, pad8(l1.Cout)), 0, l1.Cout)
            if (PAIR_BIAS and self.train and l1.act == ACT_NONE and l1.biases is not None and l2.biases is not None and l1.G == 1
                    and l2.G == 1 and l2.kh == 3 and l2.kw == 3 and l2.pad == (1, 1) and l2.dils[0] == 1 and l2.stride == 1
                    and l2.pad_mode == 0 and l1.Cout <= 256 and pad8(l2.Cout) <= 256 and l2.Cin == l1.Cout):
                self._pair_links[id(l1)] = l2         # _bw_conv: the 'a' bias gradient comes from b's output gradient
                self._pair_links[id(l2)] = l1
            self.conv(l1, pieces, tmp, B, H, W, name=name + ".a", up_add=up_add)
            return self.conv(l2, [tmp], dst, B, H, W, residual=residual, name=name + ".b", stats=stats)
        for p, (c, cp) in zip(pieces, l1.in_layout):
            assert p.Cp == cp and p.C == c, (name, p.C, p.Cp, c, cp)
        l1.need_s1 = True
        l1.need_flat = True
        l2.need_sfrag = True
        for l in (l1, l2):
            if l not in self.layers:
                self.layers.append(l)
            l.ensure_packed(self.device)
        d1, d2 = _lib.ConvDesc(), _lib.ConvDesc()
        d2.ovf_flag, d2.f16_products = self.ovf_ptr(), self.f16_products
        for d, l in ((d1, l1), (d2, l2)):
            d.B, d.H, d.W, d.Ho, d.Wo = B, H, W, H, W
            d.kh, d.kw, d.stride = l.kh, l.kw, 1
            d.pad_h, d.pad_w, d.pad_mode, d.ngroups = l.pad[0], l.pad[1], 0, 1
            for g in range(_lib.MAXGROUP):
                d.dil[g] = 1
            d.bias = l.bp.data_ptr() if l.biases is not None else None
            d.act = l.act
        d1.nseg = len(pieces)
        for i, p in enumerate(pieces):
            sg = d1.seg[i]
            sg.ptr, sg.pix_stride, sg.ch_off, sg.Cp, sg.act_in = p.ptr, p.stride, p.off, p.Cp, ACT_NONE
        d1.Ktot, d1.CoutP = l1.Ktot, l1.CoutP
        if up_add is not None:
            P, ph, pw = up_add
            assert (2 * ph, 2 * pw) == (H, W) and P.Cp >= 32 and tuple(P.buf.shape[:3]) == (B, ph, pw)
            d1.Ho, d1.Wo = ph, pw
            d1.residual, d1.res_pix_stride, d1.res_ch_off = P.ptr, P.stride, P.off
        d2.nseg = 0
        d2.Ktot, d2.CoutP = l1.CoutP, l2.sfrag_coutp()
        if l2.post is not None:
            d2.post_scale, d2.post_shift = l2.post[0].data_ptr(), l2.post[1].data_ptr()
        if residual is not None:
            d2.residual, d2.res_pix_stride, d2.res_ch_off = residual.ptr, residual.stride, residual.off
        d2.out, d2.out_pix_stride, d2.out_ch_off = dst.ptr, dst.stride, dst.off
        d2.Cout_store = min(l2.Cout_store, dst.Cp)
        assert tuple(dst.buf.shape[1:3]) == (H, W), (name, tuple(dst.buf.shape), H, W)
        self.keep += [d1, d2]

        def rescale(args, vmax, l1=l1, up_add=up_add):
            # a1 from the measured max of the slices; a2 from a bound on the 1x1 result: max|in| * max_co sum_c |w| + max|b|
            # (+ max|P| for an up-sampled addend: an interpolation never exceeds its samples)
            with torch.no_grad():
                w = l1.weights[0].detach()
                bound = vmax * float(w.abs().sum(dim=(1, 2, 3)).max())
                if l1.biases is not None and l1.biases[0] is not None:
                    bound += float(l1.biases[0].detach().abs().max())
                if up_add is not None:
                    bound += float(up_add[0].buf[..., up_add[0].off:up_add[0].off + up_add[0].Cp].abs().max())
            return args[:4] + (_a_scale_for(vmax),) + args[5:8] + (_a_scale_for(bound),) + args[9:]
        flops = 2.0 * B * H * W * (l1.Cout * l1.Cin + l2.Cout * l2.Cin * 9)
        fuse_stats = stats and STATS_FUSED and dst.Cp == int(d2.Cout_store)
        if fuse_stats:
            th = 8 if l1.CoutP == 32 else 4
            nchunk = ((W + 31) // 32) * ((H + th - 1) // th) * (2 if (l1.CoutP == 64 and d2.CoutP == 64) else 4)
            ws = self._stats_ws(d2, B, nchunk)
        self._add(self.L.egne_conv1x1_3x3_fused_f16_fwd,
                  (C.byref(d1), C.byref(d2), l1.s1hi.data_ptr(), l1.s1lo.data_ptr(), F16X3_ASCALE, l1.w_scale1,
                   l2.fhi.data_ptr(), l2.flo.data_ptr(), F16X3_ASCALE, l2.w_scale), name, flops=flops, kind="conv_f16x3:fused1x1",
                  cal=(rescale, list(pieces), B * H * W), ws=[(5, l1, "w_scale1"), (9, l2, "w_scale")])
        if fuse_stats:
            self.last_stats = self._stats_finish(ws, d2, B, H * W, nchunk, name)
        elif stats:
            self.last_stats = self.norm_stats(dst, B, H * W, name=name + ".stats")[:2]
        return H, W

    def _conv_pair_c4(self, l1, src, l2, dst, B, H, W, residual, name, stats):
        l1.need_c4h = True
        l1.need_flat = True
        l2.need_sfrag = True
        for l in (l1, l2):
            if l not in self.layers:
                self.layers.append(l)
            l.ensure_packed(self.device)
        assert l1.c4_coutp == 32
        d1, d2 = _lib.ConvDesc(), _lib.ConvDesc()
        d2.ovf_flag, d2.f16_products = self.ovf_ptr(), self.f16_products
        for d, l in ((d1, l1), (d2, l2)):
            d.B, d.H, d.W, d.Ho, d.Wo = B, H, W, H, W
            d.kh, d.kw, d.stride = 3, 3, 1
            d.pad_h, d.pad_w, d.pad_mode, d.ngroups = 1, 1, 0, 1
            for g in range(_lib.MAXGROUP):
                d.dil[g] = 1
            d.bias = l.bp.data_ptr() if l.biases is not None else None
            d.act = l.act
        d1.nseg = 1
        sg = d1.seg[0]
        sg.ptr, sg.pix_stride, sg.ch_off, sg.Cp, sg.act_in = src.ptr, src.stride, src.off, src.Cp, ACT_NONE
        d1.Ktot, d1.CoutP = src.Cp, 32
        d2.nseg = 0
        d2.Ktot, d2.CoutP = 32, 32
        if l2.post is not None:
            d2.post_scale, d2.post_shift = l2.post[0].data_ptr(), l2.post[1].data_ptr()
        if residual is not None:
            d2.residual, d2.res_pix_stride, d2.res_ch_off = residual.ptr, residual.stride, residual.off
        d2.out, d2.out_pix_stride, d2.out_ch_off = dst.ptr, dst.stride, dst.off
        d2.Cout_store = min(l2.Cout_store, dst.Cp)
        self.keep += [d1, d2]
        if C1V and isinstance(src, PlanarPiece) and l1.Cin == 1 and l1.Cout <= 32:
            # one-channel input: the first convolution on the vector ALU in exact fp32 (conv_fused_1x1_3x3_f16.hip, C1V); its weights as
            # a compact table [32][12] = nine taps, bias, two zeros per channel (scalar operands of the kernel)
            w12 = self.vec(32, 12)

            def refresh_w12(w12=w12, l1=l1):
                with torch.no_grad():
                    w12.zero_()
                    w12[:l1.Cout, :9].copy_(l1.weights[0].detach().reshape(l1.Cout, 9))
                    if l1.biases is not None and l1.biases[0] is not None:
                        w12[:l1.Cout, 9].copy_(l1.biases[0].detach())
            self.pre.append(VersionGuard([l1.weights[0]] + ([l1.biases[0]] if l1.biases is not None and l1.biases[0] is not None else []), refresh_w12))
            d1.w = w12.data_ptr()

        def rescale(args, vmax, l1=l1):
            with torch.no_grad():
                bound = vmax * float(l1.weights[0].detach().abs().sum(dim=(1, 2, 3)).max())
                if l1.biases is not None and l1.biases[0] is not None:
                    bound += float(l1.biases[0].detach().abs().max())
            return args[:4] + (_a_scale_for(vmax),) + args[5:8] + (_a_scale_for(bound),) + args[9:]
        flops = 2.0 * B * H * W * 9 * (l1.Cout * l1.Cin + l2.Cout * l2.Cin)
        fuse_stats = stats and STATS_FUSED and dst.Cp == int(d2.Cout_store)
        nchunk = ((W + 31) // 32) * ((H + 7) // 8) * 4
        if fuse_stats:
            ws = self._stats_ws(d2, B, nchunk)
        self._add(self.L.egne_conv3x3c4_3x3_fused_f16_fwd,
                  (C.byref(d1), C.byref(d2), l1.c4hi.data_ptr(), l1.c4lo.data_ptr(), F16X3_ASCALE, l1.w_scale_c4,
                   l2.fhi.data_ptr(), l2.flo.data_ptr(), F16X3_ASCALE, l2.w_scale), name, flops=flops, kind="conv_f16x3:fused3x3c4",
                  cal=(rescale, [src], B * H * W), ws=[(5, l1, "w_scale_c4"), (9, l2, "w_scale")])
        if fuse_stats:
            self.last_stats = self._stats_finish(ws, d2, B, H * W, nchunk, name)
        elif stats:
            self.last_stats = self.norm_stats(dst, B, H * W, name=name + ".stats")[:2]
        return H, W

    def _bw_conv(self, bw, layer, pieces, dst, d, B, H, W, Ho, Wo, name):
        """Backward of y = act(conv(pieces) + b): mask + bias grad, weight grad, data grads."""
        L = self.L
        Cs = int(d.Cout_store)
        npix = B * Ho * Wo
        # Was the last writer of this layer's output gradient a multi-destination 1x1 data gradient over the same pixels?  Then THAT
        # launch applies the activation mask and leaves the bias sums (egne_dst.mask_y / sums): no egne_act_bwd_bias pass here.
        masked = None
        pend = self._pending.pop((id(dst.buf), dst.off, dst.n0), None) if self._pending else None
        cand = self._mask_cands.get((id(dst.buf), dst.off, Cs, dst.n0, B, Ho, Wo)) if (MASK_ON_WRITE and self.bf16 and pend is None) else None
        if cand is not None:
            ents = self._touched.get(id(dst.buf), [])
            last = max((i for i, e in enumerate(ents) if e[0] < dst.off + dst.Cp and e[1] > dst.off), default=-1)
            needs_sums = (layer.biases is not None and layer.biases[0] is not None) or id(layer) in self._pair_links
            taken = any(q.sums for q in cand[1])              # a launch sums for ONE destination (<= 128 channels)
            if last == cand[0] and layer.act in (ACT_NONE, ACT_RELU, ACT_LEAKY) and (not needs_sums or (not taken and pad32(Cs) <= 128)):
                masked = cand
        masked3 = None
        cand3 = self._mask_cands3.get((id(dst.buf), dst.off, Cs, dst.n0, B, Ho, Wo)) if (MASK3_ON_WRITE and self.bf16 and pend is None and masked is None) else None
        if cand3 is not None and layer.act in (ACT_NONE, ACT_RELU, ACT_LEAKY):
            ents = self._touched.get(id(dst.buf), [])
            last = max((i for i, e in enumerate(ents) if e[0] < dst.off + dst.Cp and e[1] > dst.off), default=-1)
            if last == cand3[0] and int(cand3[1].Cout_store) == Cs:
                masked3 = cand3[1]
        # (the fused InstanceNorm backward below is the one masking pass that can leave part of g unread: the samples no earlier writer touched)
        acc_n = self.touched_prefix(dst.buf, dst.off, dst.Cp, dst.n0, B) if pend is not None else B
        gy = self.gp(dst, B)
        if acc_n < B:
            ents = self._touched[id(dst.buf)]
            ents[-1][4] = dst.n0 + acc_n                   # accumulates onto [n0, n0 + acc_n), stores the rest
            ents.append([dst.off, dst.off + dst.Cp, True, dst.n0 + acc_n, dst.n0 + B])
            if acc_n == 0:
                del ents[-2]
        ws = bw.vec((int(L.egne_act_bwd_bias_workspace_bytes(npix, Cs)) + 7) // 8, dtype=torch.float64)
        bias = layer.biases[0] if layer.biases is not None else None
        split_dgrad = (bw.dyn_scales and F16X3_ENABLED and layer.kh == 3 and layer.kw == 3 and layer.pad == (1, 1) and layer.dils[0] == 1
                       and layer.stride == 1 and layer.pad_mode == 0 and layer.G == 1 and layer.Cout_store >= 32)
        # weight gradient on split-f16 products too where the halo form applies and the input has a pre-scale (wgrad_halo.hip)
        split_wgrad = (bw.dyn_scales and F16X3_ENABLED and WGRAD_SPLIT and layer.kh == 3 and layer.kw == 3 and layer.pad == (1, 1)
                       and layer.dils[0] == 1 and layer.stride == 1 and layer.pad_mode == 0 and layer.G == 1 and len(pieces) == 1
                       and W >= 16 and not isinstance(pieces[0], PlanarPiece))
        if split_wgrad and pieces[0].scale is None and not d.dyn_scale:
            # raw input whose forward launch took no device pre-scale (the first-layer kernel is exact fp32): measure it here -- a
            # pass over the narrow input (8 channels for the network's first layer) against a 2x faster weight gradient
            if pieces[0].Cp <= 16:
                xm = bw._new_slot()
                bw._add(self.L.egne_absmax, (pieces[0].ptr, pieces[0].stride, pieces[0].off, pieces[0].Cp, B * H * W, xm),
                        name + ".wgrad.absmax", kind="absmax")
                d2 = _lib.ConvDesc()
                C.memmove(C.byref(d2), C.byref(d), C.sizeof(_lib.ConvDesc))
                d2.dyn_scale = xm
                bw.keep.append(d2)
                d = d2
            else:
                split_wgrad = False
        gz_max = bw._new_slot() if (split_dgrad or split_wgrad) else None     # max |gz| for the split-f16 gradients, from this pass
        peer = self._pair_links.get(id(layer))
        lead = peer is not None and layer.kh == 3            # the pair's 3x3: its chunk sums feed both bias gradients
        dbias = bias.grad.data_ptr() if bias is not None and not lead else None
        # a bias gradient is accumulated either by the main stream's reductions or by egne_pair_bias_bwd on the second stream, never by
        # both: the two read-modify-writes are ordered only by the end-of-run join (two layers wrapping ONE Parameter would race)
        if bias is not None:
            how = "side" if (peer is not None and PAIR_BIAS_SIDE) else "main"
            seen = bw.__dict__.setdefault("_bias_writers", {})
            assert seen.setdefault(id(bias), how) == how, "%s: its bias Parameter is also written from the other stream of the backward plan" % name
        done = self._premasked.pop((id(dst.buf), dst.off), None)
        if done is not None:
            # a BatchNorm behind this layer has written gz = act'(y) * BatchNorm-backward(.) and added the bias sums (egne_bn_act_bwd)
            assert done == set(range(dst.n0, dst.n0 + B)) and pend is None and masked is None and peer is None and not (split_dgrad or split_wgrad), name
        elif pend is not None:
            # the InstanceNorm backward of this layer's output (its normalised readers left their upstream gradients in _pending) inside
            # the masking pass: gz = act'(y) (g + IN-backward(a1 + act_q' up(gq) / 4)), bias sums as egne_act_bwd_bias leaves them
            assert masked is None and not (peer is not None and not lead) and (pend["B"], pend["H"], pend["W"]) == (B, Ho, Wo), name
            sums = bw.vec(B * Cs * 2)
            wsn = bw.vec((int(L.egne_norm_bwd_workspace_bytes(B, Ho * Wo, Cs, 1)) + 7) // 8, dtype=torch.float64)
            a1, gq = pend["a1"], pend["gq"]
            bw.raw(L.egne_act_norm_bwd, (gy.ptr, gy.stride, gy.off, dst.ptr, dst.stride, dst.off, layer.act, pend["scale"].data_ptr(), pend["shift"].data_ptr(),
                                         a1.ptr if a1 is not None else None, a1.stride if a1 is not None else 0, a1.off if a1 is not None else 0,
                                         gq.ptr if gq is not None else None, gq.stride if gq is not None else 0, gq.off if gq is not None else 0,
                                         pend["act_q"], Cs, B, Ho, Wo, sums.data_ptr(), wsn.data_ptr(), dbias, layer.Cout, ws.data_ptr(), acc_n), name + ".act_norm_bwd")
        elif peer is not None and not lead:
            pass                                             # the pair's 1x1: no activation to mask, bias gradient already taken (below)
        elif masked3 is not None and not (peer is not None and not lead):
            # the bf16 3x3 data gradient that wrote this slice last masks it and leaves per-wave channel sums (egne_conv_desc.mask_y / mask_sums)
            dm3 = masked3
            if layer.act != ACT_NONE:
                dm3.mask_y, dm3.mask_pix_stride, dm3.mask_ch_off, dm3.mask_act = dst.ptr, dst.stride, dst.off, layer.act
            if dbias is not None or lead:
                if layer.act == ACT_NONE:        # (sums need the masked epilogue: an identity mask on the gradient itself would do, but no layer asks)
                    dm3.mask_y, dm3.mask_pix_stride, dm3.mask_ch_off, dm3.mask_act = dst.ptr, dst.stride, dst.off, ACT_NONE
                nrows = int(L.egne_conv3x3_bf16_sum_rows())
                sums = bw.vec(nrows * Cs)
                dm3.mask_sums = sums.data_ptr()
                bw.raw(L.egne_group_sums_reduce, (sums.data_ptr(), nrows, Cs, layer.Cout if dbias is not None else Cs, dbias,
                                                  ws.data_ptr() if lead else None, 1), name + ".bias_sums")
        elif masked is not None:
            # mask and channel sums come out of the writer's epilogue; here only the sums' second stage (fixed order: deterministic).
            # A pair's 3x3 hands the totals to egne_pair_bias_bwd in the chunk-sum layout it reads (chunk 0 = the total, the rest stays zero)
            _, arr, j, dm = masked
            if layer.act != ACT_NONE:
                arr[j].mask_y, arr[j].mask_pix_stride, arr[j].mask_ch_off, arr[j].act = dst.ptr, dst.stride, dst.off, layer.act
            if dbias is not None or lead:
                nrows = int(L.egne_conv1x1_bf16_multi_waves(C.byref(dm), len(arr), arr))
                sums = bw.vec(nrows * Cs)
                arr[j].sums = sums.data_ptr()
                bw.raw(L.egne_group_sums_reduce, (sums.data_ptr(), nrows, Cs, layer.Cout if dbias is not None else Cs, dbias,
                                                  ws.data_ptr() if lead else None, 1), name + ".bias_sums")
        elif self.bf16 and layer.act == ACT_NONE and bias is None and peer is None:
            pass                                             # nothing to mask, nothing to sum (the up blocks' half-resolution 1x1)
        elif self.bf16:
            bw.raw(L.egne_act_bwd_bias, (gy.ptr, gy.stride, gy.off, dst.ptr, dst.stride, dst.off, layer.act, Cs, npix,
                                         dbias, layer.Cout, 1, ws.data_ptr()), name + ".act_bwd")
        else:
            bw.raw(L.egne_act_bwd_bias_absmax, (gy.ptr, gy.stride, gy.off, dst.ptr, dst.stride, dst.off, layer.act, Cs, npix,
                                                dbias, layer.Cout, 1, ws.data_ptr(), gz_max), name + ".act_bwd")
        if lead:
            # both bias gradients of the 1x1 -> 3x3 pair from this layer's chunk sums (still in L2) and border sums
            wsp = bw.vec((int(L.egne_pair_bias_bwd_workspace_bytes(B, Cs)) + 7) // 8, dtype=torch.float64)
            w3 = layer.weights[0]
            assert w3.is_contiguous() and tuple(w3.shape) == (layer.Cout, peer.Cout, 3, 3) and (Ho, Wo) == (H, W)
            bw._add(L.egne_pair_bias_bwd, (gy.ptr, gy.stride, gy.off, Cs, B, Ho, Wo, ws.data_ptr(), w3.data_ptr(), layer.Cout, peer.Cout,
                                           bias.grad.data_ptr(), peer.biases[0].grad.data_ptr(), wsp.data_ptr()), name + ".pair_bias",
                    kind="egne_pair_bias_bwd", side=PAIR_BIAS_SIDE)
        w = layer.weights[0]
        assert layer.G == 1 and w.grad is not None and w.grad.is_contiguous()
        gw = (C.c_void_p * 1)(w.grad.data_ptr())
        wsw = bw.vec((int(L.egne_conv2d_wgrad_workspace_bytes(C.byref(d))) + 3) // 4)
        bw.keep.append(gw)
        flops = 2.0 * npix * layer.Cout * layer.Cin * layer.kh * layer.kw
        LAYER_BYTES[name + ".wgrad"] = float(self.esz) * (B * H * W * sum(q.Cp for q in pieces) + npix * Cs)      # x and gz read once
        if split_wgrad:
            bw._add(L.egne_conv2d_wgrad_f16, (C.byref(d), gy.ptr, gy.stride, gy.off, gz_max, layer.Cout, layer.Cin, layer.kinv.data_ptr(),
                                              gw, wsw.data_ptr()), name + ".wgrad", flops=flops, kind="conv_f16x3:wgrad", side=WGRAD_SIDE_STREAM)
        else:
            bw._add(L.egne_conv2d_wgrad, (C.byref(d), gy.ptr, gy.stride, gy.off, layer.Cout, layer.Cin, layer.kinv.data_ptr(),
                                          gw, wsw.data_ptr()), name + ".wgrad", flops=flops,
                    kind="conv_bf16:wgrad" if self.bf16 else "conv_wgrad", side=WGRAD_SIDE_STREAM)
        gin = Piece(gy.buf, gy.off, layer.Cout, Cs, gy.n0)
        # 1x1 over adjacent raw slices of one buffer (dense-block conv21 / conv31 over [x | x1 | x22]): one data-gradient launch
        # for the whole run of slices instead of one per slice, each re-reading gz
        merged, skip = {}, set()
        if MERGE_DGRAD and layer.kh == 1 and layer.kw == 1 and layer.stride == 1 and layer.pad == (0, 0) and layer.pad_mode == 0 and layer.G == 1:
            i = 0
            while i < len(pieces):
                j = i
                ok = lambda q: not q.nograd and q.scale is None and q.C == q.Cp and not isinstance(q, PlanarPiece)   # noqa: E731
                while (ok(pieces[i]) and j + 1 < len(pieces) and ok(pieces[j + 1]) and pieces[j + 1].buf is pieces[i].buf
                       and pieces[j + 1].n0 == pieces[i].n0 and pieces[j + 1].off == pieces[j].off + pieces[j].Cp
                       and sum(q.Cp for q in pieces[i:j + 2]) <= 64):        # (96 outputs in one launch ran no faster than three launches)
                    j += 1
                if j > i:
                    merged[i] = j - i + 1
                    skip.update(range(i + 1, j + 1))
                i = j + 1
        if (MULTI_DGRAD and self.bf16 and BF16_FAST1X1 and layer.kh == 1 and layer.kw == 1 and layer.stride == 1 and layer.pad == (0, 0)
                and layer.pad_mode == 0 and layer.G == 1 and not merged and Cs % 8 == 0 and gy.off % 8 == 0 and gy.stride % 8 == 0
                and npix >= 4096):
            skip |= self._multi_dgrad(bw, layer, pieces, gin, gy, Cs, B, Ho, Wo, name)
        for i, pc in enumerate(pieces):
            if pc.nograd or i in skip:
                continue
            if i in merged:
                n = merged[i]
                ctot = sum(q.Cp for q in pieces[i:i + n])
                dl = DgradLayer(layer, i, span=n)
                first = self.first_touch(pc.buf, pc.off, ctot, pc.n0, B)
                mp = Piece(pc.buf, pc.off, ctot, ctot, pc.n0)
                tgt = self.gp(mp, B)
                if first:
                    self.mark_stored(mp, B)
                bw.conv(dl, [gin], tgt, B, Ho, Wo, residual=None if first else tgt, name=name + ".dgrad%d-%d" % (i, i + n - 1))
                continue
            if layer.stride != 1 or layer.pad_mode == 1:
                # reflect-padded / strided blocks: gradient w.r.t. the padded input, then fold the padding back
                assert pc.scale is None and layer.pad[0] == layer.pad[1] and (layer.pad_mode == 1 or (layer.stride, layer.kh, layer.kw, layer.pad[0]) == (2, 2, 2, 0))
                tl = TransposedLayer(layer, i, self.device)
                self.pre.append(tl.guard)            # forward plan: refreshed before the backward plan re-packs
                hp, wp = tl.out_hw(Ho, Wo)
                P = layer.pad[0]
                if tl.phase:
                    if (H + 2 * P) % 2 or (W + 2 * P) % 2 or (hp, wp) != ((H + 2 * P) // 2, (W + 2 * P) // 2):
                        raise NotImplementedError("stride-2 dgrad needs even input sizes, got %dx%d" % (H, W))
                else:
                    assert (hp, wp) == (H + 2 * P, W + 2 * P), (hp, wp, H, W, P)
                tmp = bw.buf(B, hp, wp, tl.Cout_store)
                bw.conv(tl, [gin], Piece(tmp, 0, tl.Cout, tl.Cout_store), B, Ho, Wo, name=name + ".dgrad_pad")
                tgt = self.gp(pc)
                bw.raw(L.egne_reflect_pad_bwd, (tmp.data_ptr(), tmp.shape[-1], 0, 1 if tl.phase else 0, pc.Cp, tgt.ptr, tgt.stride, tgt.off,
                                                B, H, W, P), name + ".pad_bwd")
                continue
            bf_dgrad = (self.bf16 and BF16_FAST3X3 and layer.kh == 3 and layer.kw == 3 and layer.pad == (1, 1) and layer.dils[0] == 1
                        and layer.stride == 1 and layer.pad_mode == 0 and layer.G == 1 and Cs % 8 == 0 and gy.off % 8 == 0)
            if split_dgrad:
                dl = SplitDgradLayer(layer, i, self.device)     # split-f16 arithmetic for the data gradient too
                dl.stale_scale_ok = True
                self.pre.append(dl.guard)
                bw._dyn_hint = gz_max
            elif bf_dgrad and BF16_DGRAD_PACK and pad32(pc.C) <= 256 and gy.stride % 8 == 0 and H * W * max(gy.stride, pc.stride) < 2 ** 30:
                dl = BfDgradLayer(layer, i)     # an ordinary 3x3 over gz with flipped / transposed weights, packed straight from the forward weight
            elif bf_dgrad:
                dl = SplitDgradLayer(layer, i, self.device)     # (the same through a derived tensor: two flips and a strided copy per step)
                self.pre.append(dl.guard)
            else:
                dl = DgradLayer(layer, i)
            first = self.first_touch(pc.buf, pc.off, pc.Cp, pc.n0, B)
            tgt = self.gp(pc, B)
            if first and dl.Cout_store == pc.Cp:
                self.mark_stored(pc, B)       # (both routes below store every channel of the slice when they come first)
            if pc.scale is None:
                # the first writer of a gradient slice stores, later ones accumulate (the twin is zero before the backward pass)
                ent3 = len(self._touched[id(pc.buf)]) - 1 if self._touching else -1
                bw._last_b3_desc = None
                bw.conv(dl, [gin], tgt, B, Ho, Wo, residual=None if first else tgt, name=name + ".dgrad%d" % i)
                bw._dyn_hint = None
                if MASK3_ON_WRITE and self.bf16 and bw._last_b3_desc is not None and dl.Cout_store == pc.Cp:
                    # a bf16 3x3 data gradient: possibly the LAST writer of the slice -- its producer may hang mask and bias sums on it
                    self._mask_cands3[(id(pc.buf), pc.off, pc.Cp, pc.n0, B, H, W)] = (ent3, bw._last_b3_desc)
            elif self.norm_fusable(pc) and pc.act_in == ACT_NONE and pc.scale.shape[0] == B:
                # the data gradient w.r.t. IN(x) waits for x's producer (egne_act_norm_bwd): no normalisation-backward pass here
                tmp = bw.buf(B, H, W, pc.Cp)
                bw.conv(dl, [gin], Piece(tmp, 0, pc.C, pc.Cp), B, Ho, Wo, name=name + ".dgrad%d" % i)
                bw._dyn_hint = None
                self.defer_norm_bwd(pc, pc.scale, pc.shift, B, H, W, a1=Piece(tmp, 0, pc.C, pc.Cp))
            else:
                tmp = bw.buf(B, H, W, pc.Cp)
                bw.conv(dl, [gin], Piece(tmp, 0, pc.C, pc.Cp), B, Ho, Wo, name=name + ".dgrad%d" % i)
                bw._dyn_hint = None
                sums = bw.vec(B * pc.Cp * 2)
                wsn = bw.vec((int(L.egne_norm_bwd_workspace_bytes(B, H * W, pc.Cp, 1)) + 7) // 8, dtype=torch.float64)
                bw.raw(L.egne_norm_bwd_store if first else L.egne_norm_bwd,
                                        (pc.ptr, pc.stride, pc.off, pc.scale.data_ptr(), pc.shift.data_ptr(), None,
                                         tmp.data_ptr(), tmp.shape[-1], 0, pc.act_in, pc.Cp, B, H * W, 1,
                                         tgt.ptr, tgt.stride, tgt.off, sums.data_ptr(), None, None, 0, wsn.data_ptr()),
                       name + ".in_bwd%d" % i)

    def _multi_dgrad(self, bw, layer, pieces, gin, gy, Cs, B, H, W, name):
        """Data gradients of a 1x1 over several raw bf16 slices as ONE launch (egne_conv1x1_bf16_multi_fwd: gz is read once instead of
        once per member of the would-be torch.cat).  Returns the indices of the pieces it took (the caller emits the rest one by
        one).  Each destination is remembered as the possible LAST writer of its gradient slice (``_mask_cands``): the layer
        whose output the slice is may then hang its activation mask and bias sums on it (_bw_conv)."""
        L = self.L
        ok = [i for i, q in enumerate(pieces) if not q.nograd and q.scale is None and not isinstance(q, PlanarPiece)
              and q.Cp % 8 == 0 and q.off % 8 == 0 and q.stride % 8 == 0 and q.buf.dtype == torch.bfloat16]
        best = []
        dm = _lib.ConvDesc()
        dm.dtype = 1
        dm.B, dm.H, dm.W, dm.Ho, dm.Wo = B, H, W, H, W
        dm.kh = dm.kw = dm.stride = dm.ngroups = 1
        for g in range(_lib.MAXGROUP):
            dm.dil[g] = 1
        dm.nseg = 1
        sg = dm.seg[0]
        sg.ptr, sg.pix_stride, sg.ch_off, sg.Cp = gy.ptr, gy.stride, gy.off, Cs
        dm.Ktot = Cs
        probe = (_lib.Dst * _lib.MAXDST)()
        for i in ok[:_lib.MAXDST]:            # the longest prefix of the eligible pieces whose weights fit LDS together
            probe[len(best)].C, probe[len(best)].CoutP = pieces[i].Cp, pad32(pieces[i].C)
            if not int(L.egne_conv1x1_bf16_multi_supported(C.byref(dm), len(best) + 1, probe)):
                break
            best.append(i)
        # (a layer with ONE input slice gains nothing from the shared read, but its launch may carry the mask and bias sums of the slice's
        #  producer: the up blocks' half-resolution 1x1, whose data gradient is the only writer of the previous block's output gradient)
        if len(best) < (1 if (MULTI_SINGLE and len(pieces) == 1) else 2):
            return set()
        arr = (_lib.Dst * len(best))()
        flops = 0.0
        for j, i in enumerate(best):
            pc = pieces[i]
            dl = DgradLayer(layer, i)
            dl.need_flat = dl.need_b1 = True
            if dl not in bw.layers:
                bw.layers.append(dl)
            dl.ensure_packed(self.device)
            # samples of the slice with an earlier writer: none (store), all (accumulate), or a prefix (accumulate onto it, store the rest)
            acc_n = self.touched_prefix(pc.buf, pc.off, pc.Cp, pc.n0, B) if dl.Cout_store == pc.Cp else (0 if self.first_touch(pc.buf, pc.off, pc.Cp, pc.n0, B) else B)
            first = acc_n == 0
            tgt = self.gp(pc, B)
            if acc_n < B and dl.Cout_store == pc.Cp:
                if acc_n:
                    self._touched[id(pc.buf)][-1][4] = pc.n0 + acc_n
                    self._touched[id(pc.buf)].append([pc.off, pc.off + pc.Cp, False, pc.n0 + acc_n, pc.n0 + B])
                self.mark_stored(pc, B - acc_n, n0=pc.n0 + acc_n)
            ent = len(self._touched[id(pc.buf)]) - 1 if self._touching else -1
            q = arr[j]
            q.out, q.out_pix_stride, q.out_ch_off, q.C, q.CoutP = tgt.ptr, tgt.stride, tgt.off, pc.Cp, dl.CoutP
            q.wfrag = dl.b1frag.data_ptr()
            if not first:
                q.residual, q.res_pix_stride, q.res_ch_off = tgt.ptr, tgt.stride, tgt.off
                q.res_pixels = acc_n * H * W if acc_n < B else 0
            self._mask_cands[(id(pc.buf), pc.off, pc.Cp, pc.n0, B, H, W)] = (ent, arr, j, dm)
            flops += 2.0 * B * H * W * pc.C * layer.Cout
        self.keep += [dm, arr]
        LAYER_BYTES[name + ".dgrad_multi"] = float(self.esz) * B * H * W * (Cs + sum(pieces[i].Cp * (1 if arr[j].residual is None else 2) for j, i in enumerate(best)))
        bw._add(L.egne_conv1x1_bf16_multi_fwd, (C.byref(dm), len(best), arr), name + ".dgrad_multi", flops=flops, kind="conv_bf16:1x1")
        return set(best)

    def norm_stats(self, piece, B, HW, per_sample=True, eps=1e-5, want_moments=False, name="norm_stats"):
        Bn = B if per_sample else 1
        scale, shift = self.vec(Bn, piece.Cp), self.vec(Bn, piece.Cp)
        mean = var = None
        if want_moments:
            mean, var = self.vec(Bn, piece.Cp), self.vec(Bn, piece.Cp)
        nbytes = self.L.egne_norm_stats_workspace_bytes(B, HW, piece.Cp, 1 if per_sample else 0)
        ws = self.vec((nbytes + 7) // 8, dtype=torch.float64)
        self._add(self.L.egne_norm_stats,
                  (piece.ptr, piece.stride, piece.off, piece.Cp, B, HW, 1 if per_sample else 0, eps,
                   scale.data_ptr(), shift.data_ptr(), mean.data_ptr() if want_moments else None,
                   var.data_ptr() if want_moments else None, ws.data_ptr()), name, kind="norm_stats")
        return scale, shift, mean, var

    def affine_inplace(self, piece, npix, scale, shift, name="affine"):
        self._add(self.L.egne_affine_inplace, (piece.ptr, piece.stride, piece.off, piece.Cp, npix,
                                               scale.data_ptr(), shift.data_ptr()), name, kind="affine_inplace")

    def avgpool2(self, src, dst, B, H, W, name="avgpool"):
        assert src.Cp == dst.Cp
        self._add(self.L.egne_avgpool2, (src.ptr, src.stride, src.off, dst.ptr, dst.stride, dst.off, B, H, W, src.Cp), name, kind="avgpool2")
        if self.train:
            def emit(bw, src=src, dst=dst):
                gs, gd = self.gp(src), self.gp(dst)
                bw.raw(self.L.egne_avgpool2_bwd, (gd.ptr, gd.stride, gd.off, gs.ptr, gs.stride, gs.off, B, H, W, src.Cp),
                       name + ".bwd")
            self.tape.append(emit)

    def maxpool2(self, src, dst, B, H, W, stride, name="maxpool"):
        o = lambda n: min((n - 2 + stride - 1) // stride + 1, (n - 1) // stride + 1)  # noqa: E731
        Ho, Wo = o(H), o(W)
        assert src.Cp == dst.Cp
        f16 = getattr(src, "f16s", None) is not None
        assert (not f16 and dst.f16s is None) or dst.f16s is src.f16s, "%s: a pooled f16 slice keeps its input's storage scale" % name
        self._add(self.L.egne_maxpool2_f16 if f16 else self.L.egne_maxpool2, (src.ptr, src.stride, src.off, dst.ptr, dst.stride, dst.off, B, H, W, Ho, Wo,
                                                                            stride, src.Cp), name, kind="maxpool2")
        return Ho, Wo

    def upsample2x(self, src, dst, B, H, W, name="upsample"):
        assert src.Cp == dst.Cp
        self._add(self.L.egne_upsample2x, (src.ptr, src.stride, src.off, dst.ptr, dst.stride, dst.off, B, H, W, src.Cp), name, kind="upsample2x")
        if self.train:
            def emit(bw, src=src, dst=dst):
                gs, gd = self.gp(src), self.gp(dst)
                bw.raw(self.L.egne_upsample2x_bwd, (gd.ptr, gd.stride, gd.off, gs.ptr, gs.stride, gs.off, B, H, W, src.Cp),
                       name + ".bwd")
            self.tape.append(emit)

    def upsample2x_nearest(self, src, dst, B, H, W, name="upsample_nn"):
        """F.interpolate(scale_factor=2, mode='nearest') (models/RITnet_v1.py:89)."""
        assert src.Cp == dst.Cp
        self._add(self.L.egne_upsample2x_nearest, (src.ptr, src.stride, src.off, dst.ptr, dst.stride, dst.off, B, H, W, src.Cp), name, kind="upsample2x")
        if self.train:
            def emit(bw, src=src, dst=dst):
                gs, gd = self.gp(src), self.gp(dst)
                bw.raw(self.L.egne_upsample2x_nearest_bwd, (gd.ptr, gd.stride, gd.off, gs.ptr, gs.stride, gs.off, B, H, W, src.Cp), name + ".bwd")
            self.tape.append(emit)

    def raw(self, fn, args, name):
        self._add(fn, args, name, kind=getattr(fn, "__name__", None) or "host")

    # ---- execution ---------------------------------------------------------------------------
    def run(self, events=None):
        """Replay the launches on torch's current stream.  ``events`` (a list) receives one
        (kernel family, flops, start_event, end_event) per launch -- HIP events recorded on the
        launch stream, used by bench.py to time individual kernels inside the timed region."""
        for f in self.pre:          # derived tensors first (folded BatchNorm, sliced / concatenated weights): packing reads them
            f()
        repacked = False
        for layer in self.layers:
            repacked = bool(layer.ensure_packed(self.device)) or repacked
        if repacked:
            self._refresh_wscales()
        st = _lib.stream_ptr()
        if self.ovf is not None and self.overflowed(wait=False):
            raise RuntimeError("an earlier run of this launch plan overflowed the f16 range of its calibrated pre-scales (a batch whose "
                               "activations exceed 32x the calibration batch's): the outputs of that run are invalid.  The plan has "
                               "been marked for re-calibration -- run that batch again; callers that check Plan.overflowed() / the "
                               "model's overflowed() where they synchronise never get here")
        self._runs_since_cal = getattr(self, "_runs_since_cal", 0) + 1
        if self.cal and (repacked or not self.calibrated or (RECAL_EVERY and self._runs_since_cal >= RECAL_EVERY)):
            self._runs_since_cal = 0
            r = self._run_calibrating(st)
            self._ovf_publish()
            return r
        if self.side_calls and not (events is not None and self.serial_timing):
            r = self._run_two_streams(st, events)
            self._ovf_publish()
            return r
        if events is None:
            for i, (fn, args, name) in enumerate(self.calls):
                if i == self.window_at and WINDOW_HOOKS:
                    self._open_window()
                if i == self.tail_at and self.tail_hook is not None:
                    self.tail_hook()
                rc = fn(*args, st)
                if rc != 0:
                    _lib.check(rc, name)
            self._ovf_publish()
            return
        for i, ((fn, args, name), (kind, flops)) in enumerate(zip(self.calls, self.meta)):
            if i == self.window_at and WINDOW_HOOKS:
                self._open_window()
            if i == self.tail_at and self.tail_hook is not None:
                self.tail_hook()
            if EVENT_KINDS is not None and kind.split(":")[0] not in EVENT_KINDS:      # untimed launch (each event pair costs ~2 us of GPU time)
                rc = fn(*args, st)
                if rc != 0:
                    _lib.check(rc, name)
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args, st)
            e1.record()
            if rc != 0:
                _lib.check(rc, name)
            events.append((kind, flops, e0, e1, name))
        self._ovf_publish()

    def _open_window(self):
        """Launch index ``window_at`` is where work that should run NEXT TO this plan's following launches is released (the ellipse
        searches of an earlier batch next to the edge network's deep trunk layers, whose workgroups are handed out one per free CU:
        a CU held by a search costs them 1/256 of their rate, while a launch of 256 persistent workgroups with a static share of
        the tiles each waits for its last workgroup to find a CU).  Every pending hook gets an event recorded here on the current
        stream; it makes its own stream wait for it and queues its work."""
        if torch.cuda.is_current_stream_capturing():      # (a hipGraph capture of this plan: the hooks belong to the eager loop around it)
            return
        ev = torch.cuda.Event()
        ev.record()
        # (only the hooks registered for THIS plan's device: a plan of another GPU of the process must not launch them)
        mine = torch.device(self.device).index
        mine = torch.cuda.current_device() if mine is None else mine
        hooks = [h for h in WINDOW_HOOKS if getattr(h, "device_index", mine) == mine]
        WINDOW_HOOKS[:] = [h for h in WINDOW_HOOKS if getattr(h, "device_index", mine) != mine]
        for h in hooks:
            h(ev)

    def _refresh_wscales(self):
        """A repack may have chosen another power-of-two weight scale (ensure_packed re-measures max |w|): the launches carry the
        scale by value, so rewrite the arguments that no longer match the pack."""
        for ci, ai, layer, attr in self.wscale_refs:
            v = getattr(layer, attr)
            fn, args, name = self.calls[ci]
            if args[ai] != v:
                self.calls[ci] = (fn, args[:ai] + (v,) + args[ai + 1:], name)

    def _run_two_streams(self, st, events):
        """Backward plans: the weight-gradient launches (MFMA bound, reading gz and the saved input) go to a second stream so
        that they overlap the HBM-bound data path that follows them on the main one (activation / normalisation backward, data
        gradients with their read-modify-write).  A side launch waits for an event recorded on the main stream where the plan
        placed it (gz is final there; nothing later in the run writes what it reads); the main stream joins the side stream
        at the end of the run, before the optimiser or the gradient all-reduce can see the weight gradients."""
        main = torch.cuda.current_stream()
        if self.side_stream is None:
            self.side_stream = torch.cuda.Stream(device=self.device)
            for i in self.side_calls:
                self.side_calls[i] = torch.cuda.Event()
        side = self.side_stream
        sp = C.c_void_p(side.cuda_stream)
        for i, ((fn, args, name), (kind, flops)) in enumerate(zip(self.calls, self.meta)):
            on_side = i in self.side_calls
            if i == self.window_at and WINDOW_HOOKS:
                self._open_window()
            if i in self.join_before:
                main.wait_stream(side)
            if i == self.tail_at and self.tail_hook is not None:
                # every launch that writes a non-encoder parameter gradient is queued (main stream up to here, second stream up to
                # here): the early bucket of the gradient all-reduce is issued behind both, on the second stream -- RCCL's own stream
                # waits for it, the main stream goes on with the encoder's backward
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    self.tail_hook()
            if on_side:
                ev = self.side_calls[i]
                ev.record(main)
                side.wait_event(ev)
            timed = events is not None and (EVENT_KINDS is None or kind.split(":")[0] in EVENT_KINDS)
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side if on_side else main)
            if on_side and getattr(fn, "python", False):
                with torch.cuda.stream(side):       # a python call queues torch work on the current stream: behind the second stream's launches
                    rc = fn(*args, sp)
            else:
                rc = fn(*args, sp if on_side else st)
            if timed:
                e1.record(side if on_side else main)
                events.append((kind, flops, e0, e1, name))
            if rc != 0:
                _lib.check(rc, name)
        main.wait_stream(side)


def _a_scale_for(vmax):
    """Power-of-two pre-scale that puts the measured max |x| in [1024, 2048): 32x of headroom below the f16 range for
    later batches, and elements down to 2^-14 of the max still split into two NORMAL f16 halves."""
    import math
    if vmax == 0.0:
        return F16X3_ASCALE
    e = math.floor(math.log2(2048.0 / vmax))
    return 2.0 ** max(-100, min(100, e))


def _run_calibrating(self, st):
    """First run of an inference plan (and the run after a weight update): every split-f16 launch that reads RAW
    activations first has the max |x| of its input measured on the device (egne_absmax) and its a_scale argument set
    from it.  One small sync per such launch, paid once; later runs replay the stored scales with no sync."""
    import math
    mx = torch.zeros(1, dtype=torch.int32, device=self.device)
    for i, (fn, args, name) in enumerate(self.calls):
        ent = self.cal.get(i)
        if ent is not None:
            ai, pieces, npix = ent
            mx.zero_()
            for pc in pieces:
                if isinstance(pc, PlanarPiece):        # [N][C][H][W] floats: measured as rows of four
                    assert npix * pc.C % 4 == 0
                    _lib.check(self.L.egne_absmax(pc.ptr, 4, 0, 4, npix * pc.C // 4, mx.data_ptr(), st), "absmax")
                    continue
                _lib.check(self.L.egne_absmax(pc.ptr, pc.stride, pc.off, pc.Cp, npix, mx.data_ptr(), st), "absmax")
            v = float(mx.view(torch.float32).item())
            if not math.isfinite(v):
                raise RuntimeError("non-finite activations enter %s (max |x| = %r)" % (name, v))
            args = ai(args, v) if callable(ai) else args[:ai] + (_a_scale_for(v),) + args[ai + 1:]
            self.calls[i] = (fn, args, name)
        rc = fn(*args, st)
        if rc != 0:
            _lib.check(rc, name)
        pc = self.post_cal.get(i)
        if pc is not None:        # an f16 output: stored under a bound's scale -- measure it, take the scale of its maximum, store again
            d, dst, fs, npix = pc
            mx.zero_()
            _lib.check(self.L.egne_absmax_f16(dst.ptr, dst.stride, dst.off, dst.Cp, npix, mx.data_ptr(), st), "absmax_f16")
            v = float(mx.view(torch.float32).item())
            if not math.isfinite(v):
                raise RuntimeError("non-finite activations leave %s (max |x s| = %r)" % (name, v))
            fs.vmax = v / fs.value          # max |x| of the tensor (as stored: f16-rounded): what a consumer's own bounds start from
            new = _a_scale_for(fs.vmax)
            if new != fs.value:
                fs.value = d.out_split_scale = new
                _lib.check(fn(*args, st), name)
    self.calibrated = True


Plan._run_calibrating = _run_calibrating


class _StorageLib:
    """The C-ABI as a plan sees it: entry points that have a bf16-storage twin (``_lib.BF16_TWINS``) resolve to the twin in a
    plan whose activation buffers are bf16, everything else to the library's own symbol."""

    def __init__(self, L, bf16):
        self._L, self._bf16 = L, bf16

    def __getattr__(self, name):
        fn = getattr(self._L, name + "_bf16" if (self._bf16 and name in _lib.BF16_TWINS) else name)
        setattr(self, name, fn)
        return fn


class VersionGuard:
    """Runs ``fn`` only when one of ``tensors`` was modified in place since the last run
    (load_state_dict, optimizer step): keeps tiny host-side refreshes off the steady-state path."""

    def __init__(self, tensors, fn):
        self.tensors, self.fn, self.seen = list(tensors), fn, None

    def __call__(self):
        v = tuple((t._version, t.data_ptr()) for t in self.tensors)
        if v != self.seen:
            with torch.no_grad():
                self.fn()
            self.seen = v


def maxpool_out(n, stride):
    return min((n - 2 + stride - 1) // stride + 1, (n - 1) // stride + 1)


def require_cuda(t, what):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise RuntimeError("%s must be a CUDA (ROCm) tensor: this package has no CPU path" % what)
