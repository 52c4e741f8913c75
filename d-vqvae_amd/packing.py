"""One-time weight packing: reference ``state_dict`` tensors -> the layouts the HIP kernels consume.

* eval-mode BatchNorm is folded into the preceding conv/linear in fp64 and rounded once
  (pointnet_encoder.py:29-37,150-162; eps = 1e-5);
* the STN's "+ identity" (pointnet_encoder.py:39-43) is folded into fc3's bias;
* PixelCNN conv kernels are split into per-tap [out,in] matrices; mask 'A' (models.py:61-63) is applied by
  never referencing the masked taps; the 2*dim output channels are *gate-packed* so that a tanh channel
  and its sigmoid partner land in the same MFMA accumulator position of adjacent 32-column tiles:
  packed row p -> natural row  (p//128)*64 + ((p%128)//64)*32 + p%32  +  dim * ((p%64)//32).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Optional

import numpy as np
import torch

from . import _lib

Tensor = torch.Tensor
BN_EPS = 1e-5


def _dev(t: Tensor, device) -> Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


def fold_bn(w: Tensor, b: Tensor, sd: Mapping[str, Tensor], bn: str):
    """(W', b') with BN(Wx + b) == W'x + b' (eval mode), computed in fp64."""
    w64 = w.detach().double().cpu().reshape(w.shape[0], -1)
    b64 = b.detach().double().cpu()
    g, beta = sd[bn + ".weight"].detach().double().cpu(), sd[bn + ".bias"].detach().double().cpu()
    mu, var = sd[bn + ".running_mean"].detach().double().cpu(), sd[bn + ".running_var"].detach().double().cpu()
    s = g / torch.sqrt(var + BN_EPS)
    return (w64 * s[:, None]).float(), ((b64 - mu) * s + beta).float()


def split_bf16x3(w: Tensor) -> Tensor:
    """[3, *w.shape] int16 bf16 bit patterns with planes[0] + planes[1] + planes[2] == w exactly (dvq_split_bf16x3)."""
    lib = _lib.load()
    w = w.contiguous()
    out = torch.empty((3,) + tuple(w.shape), dtype=torch.int16, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(lib.dvq_split_bf16x3(w.data_ptr(), w.numel(), out.data_ptr(),
                                        torch.cuda.current_stream(w.device).cuda_stream), "dvq_split_bf16x3")
    return out


_KIND_OVERRIDE: list = []


class gemm_kind_as:
    """``with gemm_kind_as(_lib.PLANES_BF16X3): ...`` -- weight images are (re)built in the given kind inside the block
    (GenNet.gen's range fallback); the images of the surrounding kind are rebuilt at the next use after it."""

    def __init__(self, kind: int):
        self.kind = kind

    def __enter__(self):
        _KIND_OVERRIDE.append(self.kind)
        return self

    def __exit__(self, *exc):
        _KIND_OVERRIDE.pop()
        return False


def gemm_kind() -> int:
    """Which weight image the GEMM weights are packed into: the fp16 three-product split (default, DVQ_GEMM unset or
    "f16x2"), or the exact three-plane bf16 split with DVQ_GEMM=bf16x3 (and with DVQ_GEMM=fp32, where the library ignores
    the images and runs the fp32 matrix-core kernel)."""
    import os
    if _KIND_OVERRIDE:
        return _KIND_OVERRIDE[-1]
    v = os.environ.get("DVQ_GEMM", "").strip().lower()
    if v in ("", "f16x2"):
        return _lib.PLANES_F16X2
    if v in ("bf16x3", "fp32"):
        return _lib.PLANES_BF16X3
    raise RuntimeError(f"DVQ_GEMM={v!r}: expected f16x2 (default), bf16x3 or fp32")


class Planes:
    """Pre-split image of one weight tensor [..., N, K]: ``planes`` int16 [P, *w.shape] (P = 3: bf16x3, 2: f16x2), ``scale`` =
    the f16x2 row scales [N] (shared by every tensor split in the same group), ``kind`` = DVQ_PLANES_*."""
    __slots__ = ("kind", "planes", "scale")

    def __init__(self, kind: int, planes: Tensor, scale: Optional[Tensor] = None):
        self.kind, self.planes, self.scale = kind, planes, scale


def split_f16x2(ws) -> list:
    """fp16 images of tensors whose products are summed into the SAME output rows (the taps of a conv; horiz_stack and
    vert_to_horiz of a gated layer): each ``w`` is [N, K] or [outer, N, K]; one power-of-two scale per row for the whole group
    (dvq_f16x2_row_absmax + dvq_split_f16x2).  Returns one ``Planes`` per tensor, all holding the same ``scale`` tensor."""
    lib = _lib.load()
    ws = [w.contiguous() for w in ws]
    N = ws[0].shape[-2]
    dev = ws[0].device
    if any(w.dim() not in (2, 3) or w.shape[-2] != N or w.dtype != torch.float32 or w.device != dev for w in ws):
        raise RuntimeError("split_f16x2: float32 tensors [N,K] / [outer,N,K] with the same N on one device")
    absmax = torch.zeros(N, dtype=torch.float32, device=dev)
    scale = torch.empty(N, dtype=torch.float32, device=dev)
    out = []
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        for w in ws:
            outer = w.shape[0] if w.dim() == 3 else 1
            _lib.check(lib.dvq_f16x2_row_absmax(w.data_ptr(), outer, N, w.shape[-1], absmax.data_ptr(), st), "dvq_f16x2_row_absmax")
        for w in ws:
            outer = w.shape[0] if w.dim() == 3 else 1
            pl = torch.empty((2,) + tuple(w.shape), dtype=torch.int16, device=dev)
            _lib.check(lib.dvq_split_f16x2(w.data_ptr(), outer, N, w.shape[-1], absmax.data_ptr(), pl.data_ptr(), scale.data_ptr(), st),
                       "dvq_split_f16x2")
            out.append(Planes(_lib.PLANES_F16X2, pl, scale))
    return out


def split_planes(w: Tensor, kind: Optional[int] = None) -> Planes:
    """The weight image of one [N, K] matrix in the current (or the given) kind."""
    kind = gemm_kind() if kind is None else kind
    if kind == _lib.PLANES_F16X2:
        return split_f16x2([w])[0]
    return Planes(_lib.PLANES_BF16X3, split_bf16x3(w))


def pointnet_filter_image(w2: Tensor, w3: Tensor) -> Tensor:
    """uint8 image of a trunk's conv2 [128,64] and conv3 [1024,128] weights (BatchNorm folded) for the filtered PointNet trunk
    (dvq_pointnet_pack_filter)."""
    lib = _lib.load()
    w2, w3 = w2.contiguous(), w3.contiguous()
    if tuple(w3.shape) != (1024, 128) or tuple(w2.shape) != (128, 64):
        raise RuntimeError(f"pointnet filter image: conv2 / conv3 weights must be [128,64] / [1024,128], got {tuple(w2.shape)} / {tuple(w3.shape)}")
    out = torch.empty(lib.dvq_pointnet_filter_bytes(), dtype=torch.uint8, device=w3.device)
    with torch.cuda.device(w3.device):
        _lib.check(lib.dvq_pointnet_pack_filter(w2.data_ptr(), w3.data_ptr(), out.data_ptr(),
                                                torch.cuda.current_stream(w3.device).cuda_stream), "dvq_pointnet_pack_filter")
    return out


class _Packed:
    """Holds device tensors + the ctypes struct that points at them; re-homed lazily with .to(device)."""

    def __init__(self):
        self.tensors: Dict[str, Tensor] = {}
        self.device = None
        self.cstruct = None

    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self.device == device and getattr(self, "kind", None) == self._kind():
            return self
        derived = ("__planes", "__filter", "__scale", "__tables")
        stash = self.__dict__.setdefault("_stash", {})
        if self.device == device and getattr(self, "kind", None) is not None:
            # the weight-image kind flips on the same device (a range fallback and the return from it): park this kind's images and
            # bound struct instead of dropping them -- the way back (class tables of the prior included) costs nothing
            stash[self.kind] = ({k: v for k, v in self.tensors.items() if k.endswith(derived)}, self.cstruct,
                                {a: self.__dict__[a] for a in ("_layers",) if a in self.__dict__})
            hit = stash.get(self._kind())
            if hit is not None:
                self.tensors = {k: v for k, v in self.tensors.items() if not k.endswith(derived)}
                self.tensors.update(hit[0])
                self.cstruct = hit[1]
                self.__dict__.update(hit[2])
                self.kind = self._kind()
                return self
        else:
            stash.clear()
        self.tensors = {k: (v.to(device) if v.dtype == torch.int16 else _dev(v, device)) for k, v in self.tensors.items()
                        if not k.endswith(derived)}
        self.device = device
        self.kind = self._kind()
        for scale_key, keys in self._plane_groups():    # weight images of the GEMM weights, built on the device
            srcs = [self._plane_source(k, self.tensors[k]) for k in keys]
            if self.kind == _lib.PLANES_F16X2:
                pls = split_f16x2(srcs)
                for k, pl in zip(keys, pls):
                    self.tensors[k + "__planes"] = pl.planes
                self.tensors[scale_key + "__scale"] = pls[0].scale
            else:
                for k, src in zip(keys, srcs):
                    self.tensors[k + "__planes"] = split_bf16x3(src)
        self._bind()
        return self

    def _kind(self) -> int:
        return gemm_kind()

    def _plane_groups(self):
        """[(name of the group's row scales, [tensor keys whose products are summed into the same outputs])]"""
        return []

    def _plane_source(self, key, t):
        return t

    def planes_ptr(self, key):
        t = self.tensors.get(key + "__planes")
        return t.data_ptr() if t is not None else None

    def scale_ptr(self, key):
        t = self.tensors.get(key + "__scale")
        return t.data_ptr() if t is not None else None

    def _bind(self):
        raise NotImplementedError


class PackedPointNet(_Packed):
    def __init__(self, sd: Mapping[str, Tensor], prefix: str = ""):
        super().__init__()
        g = lambda k: sd[prefix + k]
        self.C = int(g("conv1.weight").shape[1])
        if self.C not in (3, 4):
            raise RuntimeError(f"PointNetEncoder with channel={self.C} is not supported by the HIP path (3 or 4)")
        t = self.tensors
        local = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        for pre, tag in (("stn.", "s_"), ("", "")):
            for i in (1, 2, 3):
                w, b = fold_bn(g(f"{pre}conv{i}.weight"), g(f"{pre}conv{i}.bias"), local, f"{pre}bn{i}")
                if i == 1:                      # pad the C (3 or 4) input channels to 4
                    w4 = torch.zeros(64, 4)
                    w4[:, : self.C] = w
                    w = w4
                t[f"{tag}w{i}"], t[f"{tag}b{i}"] = w, b
        t["s_f1"], t["s_c1"] = fold_bn(g("stn.fc1.weight"), g("stn.fc1.bias"), local, "stn.bn4")
        t["s_f2"], t["s_c2"] = fold_bn(g("stn.fc2.weight"), g("stn.fc2.bias"), local, "stn.bn5")
        t["s_f3"] = g("stn.fc3.weight").detach().float().cpu()
        t["s_c3"] = g("stn.fc3.bias").detach().float().cpu() + torch.eye(3).reshape(9)

    def _bind(self):
        s = _lib.PointnetWeights()
        s.C = self.C
        for key in ("w3", "s_w3"):                      # conv3 filter images (fp16 rows + scales + norms), built on the device
            self.tensors[key + "__filter"] = pointnet_filter_image(self.tensors[key[:-1] + "2"], self.tensors[key])   # (s_)w2, (s_)w3
        for name, _ in _lib.PointnetWeights._fields_[2:]:
            if name.endswith("f"):
                setattr(s, name, self.tensors[name[:-1] + "__filter"].data_ptr())
            elif name.endswith("p"):
                setattr(s, name, self.planes_ptr(name[:-1]))
            else:
                setattr(s, name, self.tensors[name].data_ptr())
        self.cstruct = s

    def _kind(self):                                     # the trunk kernels multiply conv2 / conv3 themselves (six-product bf16 split,
        return _lib.PLANES_BF16X3                        # fp16 filter): their images -- and the small STN FCs' -- stay bf16x3

    def _plane_groups(self):
        return [(k, [k]) for k in ("s_w2", "s_w3", "s_f1", "s_f2", "s_f3", "w2", "w3")]

    # the fused trunk kernel consumes conv3's weights with the k order inside every 16-channel block permuted to the order in
    # which conv2's accumulator delivers its channels (pos 8h+j <-> channel 8(j>>2)+4h+(j&3)); see csrc/pointnet.hip
    K_PERM16 = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)

    def _plane_source(self, key, t):
        if key in ("w3", "s_w3"):
            idx = torch.tensor([16 * b + p for b in range(t.shape[1] // 16) for p in self.K_PERM16], device=t.device)
            return t.index_select(1, idx).contiguous()
        return t


def gate_perm(dim: int) -> Tensor:
    p = torch.arange(2 * dim)
    return (p // 128) * 64 + ((p % 128) // 64) * 32 + (p % 32) + dim * ((p % 64) // 32)


class PackedPixelCNN(_Packed):
    def __init__(self, sd: Mapping[str, Tensor], prefix: str = ""):
        super().__init__()
        g = lambda k: sd[prefix + k].detach().float().cpu()
        emb = g("embedding.weight")
        self.n_in, self.dim = int(emb.shape[0]), int(emb.shape[1])
        if self.dim % 64 != 0:
            raise RuntimeError(f"GatedPixelCNN dim={self.dim}: the HIP path needs a multiple of 64")
        n = 0
        while f"{prefix}layers.{n}.vert_stack.weight" in sd:
            n += 1
        self.n_layers = n
        self.n_classes = int(g("layers.0.class_cond_embedding.weight").shape[0])
        self.n_hidden = int(g("output_conv.0.weight").shape[0])
        self.n_out = int(g("output_conv.2.weight").shape[0])
        if self.n_out != self.n_in:
            raise RuntimeError("GatedPixelCNN: output classes != input tokens")
        P = gate_perm(self.dim)
        t = self.tensors
        t["tok_emb"] = emb
        for i in range(n):
            lp = f"layers.{i}."
            wv = g(lp + "vert_stack.weight")            # [2d, d, KR, k]
            KR, k = wv.shape[2], wv.shape[3]
            assert k == (5 if i == 0 else 3) and KR == k // 2 + 1, "unexpected PixelCNN kernel geometry"
            t[f"wv{i}"] = wv.permute(2, 3, 0, 1)[:, :, P, :].reshape(KR * k, 2 * self.dim, self.dim)
            t[f"bv{i}"] = g(lp + "vert_stack.bias")[P]
            wh = g(lp + "horiz_stack.weight")           # [2d, d, 1, KC]
            t[f"wh{i}"] = wh[:, :, 0, :].permute(2, 0, 1)[:, P, :]
            t[f"wv2h{i}"] = g(lp + "vert_to_horiz.weight")[:, :, 0, 0][P][:, P]
            t[f"bh{i}"] = (g(lp + "horiz_stack.bias") + g(lp + "vert_to_horiz.bias"))[P]
            t[f"cls{i}"] = g(lp + "class_cond_embedding.weight")[:, P]
            t[f"wr{i}"] = g(lp + "horiz_resid.weight")[:, :, 0, 0]
            t[f"br{i}"] = g(lp + "horiz_resid.bias")
        t["w0"], t["b0"] = g("output_conv.0.weight")[:, :, 0, 0], g("output_conv.0.bias")
        t["w2"], t["b2"] = g("output_conv.2.weight")[:, :, 0, 0], g("output_conv.2.bias")

    def _bind(self):
        t = self.tensors
        self._layers = (_lib.PixelcnnLayer * self.n_layers)()
        for i in range(self.n_layers):
            for name, _ in _lib.PixelcnnLayer._fields_:
                if name.endswith("_p"):
                    setattr(self._layers[i], name, self.planes_ptr(f"{name[:-2]}{i}"))
                elif name in ("sv", "sh", "sr"):
                    setattr(self._layers[i], name, self.scale_ptr(f"{name}{i}"))
                else:
                    setattr(self._layers[i], name, t[f"{name}{i}"].data_ptr())
        s = _lib.PixelcnnWeights()
        s.n_layers, s.dim, s.n_in, s.n_classes, s.n_hidden = self.n_layers, self.dim, self.n_in, self.n_classes, self.n_hidden
        s.tok_emb = t["tok_emb"].data_ptr()
        s.layers_host = C.cast(self._layers, C.POINTER(_lib.PixelcnnLayer))
        s.w0, s.b0, s.w2, s.b2 = (t[k].data_ptr() for k in ("w0", "b0", "w2", "b2"))
        s.w0_p, s.w2_p = self.planes_ptr("w0"), self.planes_ptr("w2")
        s.s0, s.s2 = self.scale_ptr("s0"), self.scale_ptr("s2")
        s.planes_kind = self.kind
        s.class_tables = None
        self.cstruct = s
        # What depends on the class label only (grid row 0's vertical stack, position (0, 0), the accumulator states that follow from
        # them) is a function of the weights: built once here (include/dvq.h: dvq_pixelcnn_build_tables), read by every call.
        lib = _lib.load()
        nbytes = lib.dvq_pixelcnn_tables_bytes(C.byref(s)) if self.device is not None and self.device.type == "cuda" else 0
        if nbytes:
            with torch.cuda.device(self.device):
                buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                _lib.check(lib.dvq_pixelcnn_build_tables(C.byref(s), buf.data_ptr(), nbytes, torch.cuda.current_stream(self.device).cuda_stream),
                           "dvq_pixelcnn_build_tables")
            self.tensors["class__tables"] = buf
            s.class_tables = buf.data_ptr()

    def _plane_groups(self):
        g = []
        for i in range(self.n_layers):                  # horiz_stack's taps and vert_to_horiz feed ONE gate (models.py:76-81)
            g += [(f"sv{i}", [f"wv{i}"]), (f"sh{i}", [f"wh{i}", f"wv2h{i}"]), (f"sr{i}", [f"wr{i}"])]
        return g + [("s0", ["w0"]), ("s2", ["w2"])]


class PackedMano(_Packed):
    """arrays: v_template [778,3], shapedirs [778,3,10], posedirs [778,3,135], J_regressor [16,778],
    weights [778,16], hands_components [45,45], hands_mean [45], parents [16]."""

    def __init__(self, arrays: Mapping[str, np.ndarray], flat_hand_mean: bool = True, n_comps: int = 45):
        super().__init__()
        if n_comps != 45:
            raise RuntimeError("the grasp path uses num_pca_comps=45 (gen_diverse_grasp_obman.py:358)")
        f64 = lambda k: np.asarray(arrays[k], dtype=np.float64)
        vt, sh, po = f64("v_template"), f64("shapedirs")[:, :, :10], f64("posedirs")
        jr = f64("J_regressor")
        t = self.tensors
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        t["v_template"] = tt(vt)
        blend = np.zeros((778 * 3, 160))                  # row e = [shapedirs[e, :10] | posedirs[e, :135] | 0]: V = X @ blend^T
        blend[:, :10] = sh.reshape(778 * 3, 10)
        blend[:, 10:145] = po.reshape(778 * 3, 135)
        t["blend_w"] = tt(blend)
        t["j_template"] = tt(jr @ vt)
        t["j_shapedirs"] = tt(np.einsum("jv,vkl->ljk", jr, sh).reshape(10, 48))
        t["weights"] = tt(f64("weights"))
        t["comps"] = tt(f64("hands_components")[:45])
        mean = np.zeros(45) if flat_hand_mean else f64("hands_mean")
        t["pose_mean"] = tt(np.concatenate([np.zeros(3), mean]))
        self.parents = [int(p) for p in np.asarray(arrays["parents"]).reshape(-1)]
        self.parents[0] = -1
        self.faces = np.asarray(arrays.get("faces", np.zeros((0, 3), dtype=np.int64)))

    def _plane_groups(self):
        return [("blend_w", ["blend_w"])]

    def _bind(self):
        s = _lib.ManoModel()
        for name in ("v_template", "blend_w", "j_template", "j_shapedirs", "weights", "comps", "pose_mean"):
            setattr(s, name, self.tensors[name].data_ptr())
        s.blend_w_planes = self.planes_ptr("blend_w")
        s.blend_w_scale = self.scale_ptr("blend_w")
        s.planes_kind = self.kind
        for j, p in enumerate(self.parents):
            s.parents[j] = p
        self.cstruct = s
