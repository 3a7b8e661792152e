"""Torch7 ``.t7`` checkpoint tools (pure Python, no compute).

The reference loads its weights with ``torch.load(path).model`` (fast_artistic_video_core.lua:39-46);
the checkpoints are written by train_video.lua:508-534.  No checkpoint can be downloaded offline, so
this module provides

* ``write_checkpoint`` / ``make_synthetic_checkpoint`` -- emit the Torch7 binary grammar
   for a model built the way models_video.lua:55-140 builds it, including the
  lazily inserted ``nn.SpatialReflectionPadding`` (train_video.lua:319-325);
* ``load`` / ``extract_layers`` -- an independent Python reader used by the tests to cross-check
  the product's C++ reader (csrc/t7_reader.cpp) and to feed the CPU oracle.

The product path (libfav) reads ``.t7`` files with its own C++ reader; it never calls this module.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import numpy as np

TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN = 0, 1, 2, 3, 4, 5

_STORAGE_DTYPES = {
    "torch.FloatStorage": np.dtype("<f4"), "torch.DoubleStorage": np.dtype("<f8"),
    "torch.LongStorage": np.dtype("<i8"), "torch.IntStorage": np.dtype("<i4"),
    "torch.ByteStorage": np.dtype("u1"), "torch.CharStorage": np.dtype("i1"),
    "torch.ShortStorage": np.dtype("<i2"),
}
_TENSOR_TO_STORAGE = {k.replace("Storage", "Tensor"): k for k in _STORAGE_DTYPES}
_DTYPE_TO_TENSOR = {np.dtype("float32"): "torch.FloatTensor", np.dtype("float64"): "torch.DoubleTensor",
                    np.dtype("int64"): "torch.LongTensor", np.dtype("uint8"): "torch.ByteTensor"}


@dataclass
class TorchObject:
    """A serialised torch class instance that is not a tensor/storage (e.g. an nn module)."""
    cls: str
    fields: Dict[Any, Any] = field(default_factory=dict)

    def __getitem__(self, k):
        return self.fields[k]

    def get(self, k, d=None):
        return self.fields.get(k, d)


# ----------------------------------------------------------------------------------------------
# writer
# ----------------------------------------------------------------------------------------------
class _Writer:
    def __init__(self):
        self.b = bytearray()
        self.idx = 0
        self.seen: Dict[int, int] = {}

    def i32(self, v): self.b += struct.pack("<i", int(v))
    def i64(self, v): self.b += struct.pack("<q", int(v))
    def raw_string(self, s: str):
        e = s.encode("latin-1"); self.i32(len(e)); self.b += e

    def obj(self, o):
        if o is None:
            self.i32(TYPE_NIL)
        elif isinstance(o, bool):
            self.i32(TYPE_BOOLEAN); self.i32(1 if o else 0)
        elif isinstance(o, (int, float, np.integer, np.floating)):
            self.i32(TYPE_NUMBER); self.b += struct.pack("<d", float(o))
        elif isinstance(o, str):
            self.i32(TYPE_STRING); self.raw_string(o)
        elif isinstance(o, np.ndarray):
            self._tensor(o)
        elif isinstance(o, TorchObject):
            self.i32(TYPE_TORCH)
            if self._index(o):
                return
            self.raw_string("V 1"); self.raw_string(o.cls)
            self.obj(dict(o.fields))
        elif isinstance(o, (list, tuple)):
            self.obj({i + 1: v for i, v in enumerate(o)})
        elif isinstance(o, dict):
            self.i32(TYPE_TABLE)
            self.idx += 1; self.i32(self.idx)            # plain tables are never shared here
            self.i32(len(o))
            for k, v in o.items():
                self.obj(k); self.obj(v)
        else:
            raise TypeError(f"cannot serialise {type(o)}")

    def _index(self, o) -> bool:
        """write the object index; True if it was a back-reference (no body follows)."""
        if id(o) in self.seen:
            self.i32(self.seen[id(o)]); return True
        self.idx += 1; self.seen[id(o)] = self.idx; self.i32(self.idx); return False

    def _tensor(self, a: np.ndarray):
        tname = _DTYPE_TO_TENSOR[a.dtype]
        a = np.ascontiguousarray(a)
        self.i32(TYPE_TORCH); self.idx += 1; self.i32(self.idx)
        self.raw_string("V 1"); self.raw_string(tname)
        self.i32(a.ndim)
        for s in a.shape: self.i64(s)
        for s in a.strides: self.i64(s // a.itemsize)
        self.i64(1)                                      # storageOffset, 1-based
        if a.size == 0:
            self.i32(TYPE_NIL); return
        self.i32(TYPE_TORCH); self.idx += 1; self.i32(self.idx)
        self.raw_string("V 1"); self.raw_string(_TENSOR_TO_STORAGE[tname])
        self.i64(a.size)
        self.b += a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()


def write_checkpoint(path: str, checkpoint: dict) -> None:
    w = _Writer(); w.obj(checkpoint)
    with open(path, "wb") as f:
        f.write(bytes(w.b))


# ----------------------------------------------------------------------------------------------
# reader
# ----------------------------------------------------------------------------------------------
class _Reader:
    def __init__(self, data: bytes):
        self.d = data; self.p = 0; self.memo: Dict[int, Any] = {}

    def i32(self):
        v = struct.unpack_from("<i", self.d, self.p)[0]; self.p += 4; return v
    def i64(self):
        v = struct.unpack_from("<q", self.d, self.p)[0]; self.p += 8; return v
    def raw_string(self):
        n = self.i32(); s = self.d[self.p:self.p + n].decode("latin-1"); self.p += n; return s

    def obj(self):
        t = self.i32()
        if t == TYPE_NIL: return None
        if t == TYPE_NUMBER:
            v = struct.unpack_from("<d", self.d, self.p)[0]; self.p += 8
            return int(v) if float(v).is_integer() and abs(v) < 2 ** 53 else v
        if t == TYPE_STRING: return self.raw_string()
        if t == TYPE_BOOLEAN: return self.i32() != 0
        if t == TYPE_TABLE:
            idx = self.i32()
            if idx in self.memo: return self.memo[idx]
            out: Dict[Any, Any] = {}; self.memo[idx] = out
            for _ in range(self.i32()):
                k = self.obj(); out[k] = self.obj()
            return out
        if t == TYPE_TORCH:
            idx = self.i32()
            if idx in self.memo: return self.memo[idx]
            version = self.raw_string()
            cls = self.raw_string() if version.startswith("V ") else version
            if cls in _TENSOR_TO_STORAGE:
                nd = self.i32()
                size = [self.i64() for _ in range(nd)]
                stride = [self.i64() for _ in range(nd)]
                off = self.i64() - 1
                storage = self.obj()
                if storage is None or nd == 0:
                    arr = np.zeros((0,), _STORAGE_DTYPES[_TENSOR_TO_STORAGE[cls]])
                else:
                    arr = np.lib.stride_tricks.as_strided(
                        storage[off:], shape=size, strides=[s * storage.itemsize for s in stride]).copy()
                self.memo[idx] = arr; return arr
            if cls in _STORAGE_DTYPES:
                n = self.i64(); dt = _STORAGE_DTYPES[cls]
                arr = np.frombuffer(self.d, dt, n, self.p).copy(); self.p += n * dt.itemsize
                self.memo[idx] = arr; return arr
            o = TorchObject(cls); self.memo[idx] = o
            body = self.obj()
            o.fields = body if isinstance(body, dict) else {"_payload": body}
            return o
        raise ValueError(f"unsupported .t7 type tag {t} at byte {self.p - 4}")


def load(path: str):
    with open(path, "rb") as f:
        return _Reader(f.read()).obj()


def _seq(table) -> list:
    """Lua array table {1:..., 2:...} -> list."""
    return [table[k] for k in sorted(k for k in table if isinstance(k, int))]


def extract_layers(module) -> List[dict]:
    """Flatten an ``nn.Sequential`` (as read by :func:`load`) into the layer list the oracle runs."""
    out: List[dict] = []
    for m in _seq(module["modules"]):
        c = m.cls
        if c == "nn.Sequential":
            mods = _seq(m["modules"])
            if len(mods) == 2 and mods[0].cls == "nn.ConcatTable" and mods[1].cls == "nn.CAddTable":
                branches = _seq(mods[0]["modules"])
                skip = branches[1]
                shave = int(skip["size"]) if skip.cls == "nn.ShaveImage" else 0
                out.append({"type": "res", "block": extract_layers(branches[0]), "shave": shave})
            else:
                out.extend(extract_layers(m))
        elif c in ("nn.SpatialReflectionPadding", "nn.SpatialReplicationPadding"):
            out.append({"type": "pad", "l": int(m["pad_l"]), "r": int(m["pad_r"]),
                        "t": int(m["pad_t"]), "b": int(m["pad_b"]),
                        "mode": "replicate" if c == "nn.SpatialReplicationPadding" else "reflect"})
        elif c == "nn.SpatialConvolution":
            cin, cout, kw, kh = int(m["nInputPlane"]), int(m["nOutputPlane"]), int(m["kW"]), int(m["kH"])
            w = np.asarray(m["weight"], np.float32).reshape(cout, cin, kh, kw)
            b = m.get("bias"); b = None if b is None or np.size(b) == 0 else np.asarray(b, np.float32)
            out.append({"type": "conv", "w": w, "b": b, "stride": int(m["dW"]), "pad": int(m.get("padW", 0))})
            assert int(m["dW"]) == int(m["dH"]) and int(m.get("padW", 0)) == int(m.get("padH", 0))
        elif c == "nn.SpatialFullConvolution":
            cin, cout, kw, kh = int(m["nInputPlane"]), int(m["nOutputPlane"]), int(m["kW"]), int(m["kH"])
            w = np.asarray(m["weight"], np.float32).reshape(cin, cout, kh, kw)
            b = m.get("bias"); b = None if b is None or np.size(b) == 0 else np.asarray(b, np.float32)
            out.append({"type": "fullconv", "w": w, "b": b, "stride": int(m["dW"]), "pad": int(m.get("padW", 0)), "adj": int(m.get("adjW", 0))})
        elif c == "nn.SpatialBatchNormalization":
            var = m.get("running_var")
            if var is None:    # very old checkpoints store running_std = 1/sqrt(var + eps)
                std = np.asarray(m["running_std"], np.float64); var = 1.0 / (std * std) - float(m.get("eps", 1e-5))
            out.append({"type": "bn", "mean": np.asarray(m["running_mean"], np.float32), "var": np.asarray(var, np.float32),
                        "gamma": np.asarray(m["weight"], np.float32), "beta": np.asarray(m["bias"], np.float32), "eps": float(m.get("eps", 1e-5))})
        elif c == "nn.InstanceNormalization":
            out.append({"type": "in", "gamma": np.asarray(m["weight"], np.float32),
                        "beta": np.asarray(m["bias"], np.float32), "eps": float(m.get("eps", 1e-5))})
        elif c == "nn.ReLU":
            out.append({"type": "relu"})
        elif c == "nn.SpatialUpSamplingNearest":
            out.append({"type": "up", "s": int(m["scale_factor"])})
        elif c == "nn.Tanh":
            out.append({"type": "tanh"})
        elif c == "nn.MulConstant":
            out.append({"type": "mul", "k": float(m["constant_scalar"])})
        elif c in ("nn.TotalVariation", "nn.Identity"):
            out.append({"type": "identity"})
        else:
            raise ValueError(f"unsupported module {c}")
    return out


# ----------------------------------------------------------------------------------------------
# synthetic checkpoints (the published ones cannot be downloaded offline)
# ----------------------------------------------------------------------------------------------
CANONICAL_ARCH = "c9s1-32,d64,d128,R128,R128,R128,R128,R128,U2,c3s1-64,U2,c9s1-3"   # README.md:299-307


def _conv(rng, cin, cout, k, s, p):
    std = np.sqrt(2.0 / (cin * k * k))
    w = (rng.standard_normal((cout, cin, k, k)) * std).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
    return TorchObject("nn.SpatialConvolution", {
        "nInputPlane": cin, "nOutputPlane": cout, "kW": k, "kH": k, "dW": s, "dH": s, "padW": p, "padH": p,
        "weight": w, "bias": b, "gradWeight": np.zeros((0,), np.float32), "gradBias": np.zeros((0,), np.float32),
        "_type": "torch.FloatTensor", "train": False, "output": np.zeros((0,), np.float32),
        "gradInput": np.zeros((0,), np.float32)})


def _fullconv(rng, cin, cout, k, s, p, adj):
    std = np.sqrt(2.0 / (cin * k * k))
    w = (rng.standard_normal((cin, cout, k, k)) * std).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, cout).astype(np.float32)
    return TorchObject("nn.SpatialFullConvolution", {
        "nInputPlane": cin, "nOutputPlane": cout, "kW": k, "kH": k, "dW": s, "dH": s, "padW": p, "padH": p, "adjW": adj, "adjH": adj,
        "weight": w, "bias": b, "_type": "torch.FloatTensor", "train": False})


def _bnorm(rng, c):
    return TorchObject("nn.SpatialBatchNormalization", {
        "eps": 1e-5, "momentum": 0.1, "affine": True, "train": False,
        "running_mean": rng.standard_normal(c).astype(np.float32), "running_var": rng.uniform(0.5, 2.0, c).astype(np.float32),
        "weight": rng.uniform(0.0, 1.0, c).astype(np.float32), "bias": rng.uniform(-0.2, 0.2, c).astype(np.float32)})


def _inorm(rng, c):
    # InstanceNormalization.lua:18-30: weight ~ U(0,1), bias = 0; nested bn is baggage the reader skips
    bn = TorchObject("nn.SpatialBatchNormalization", {"eps": 1e-5, "momentum": 0.1, "affine": True, "train": True,
                                                      "weight": np.zeros((c,), np.float32),
                                                      "bias": np.zeros((c,), np.float32)})
    return TorchObject("nn.InstanceNormalization", {
        "eps": 1e-5, "nOutput": c, "prev_N": 1, "weight": rng.uniform(0.0, 1.0, c).astype(np.float32),
        "bias": np.zeros((c,), np.float32), "gradWeight": np.zeros((c,), np.float32),
        "gradBias": np.zeros((c,), np.float32), "bn": bn, "_type": "torch.FloatTensor", "train": False})


def _simple(cls, **kw):
    kw.setdefault("_type", "torch.FloatTensor"); kw.setdefault("train", False)
    return TorchObject(cls, kw)


def _sequential(mods):
    return TorchObject("nn.Sequential", {"modules": list(mods), "_type": "torch.FloatTensor", "train": False,
                                         "output": np.zeros((0,), np.float32),
                                         "gradInput": np.zeros((0,), np.float32)})


IMAGE_ARCH = "c9s1-32,d64,d128,R128,R128,R128,R128,R128,u64,u32,c9s1-3"        # train_video.lua:21 (fast-neural-style image models)


PADDING_TYPES = ("reflect-start", "none", "reflect", "replicate", "zero")          # train_video.lua:25


def _padlayer(padding_type, p):
    cls = "nn.SpatialReflectionPadding" if padding_type == "reflect" else "nn.SpatialReplicationPadding"
    return TorchObject(cls, {"pad_l": p, "pad_r": p, "pad_t": p, "pad_b": p, "_type": "torch.FloatTensor", "train": False})


def build_model(arch: str = CANONICAL_ARCH, seed: int = 0, in_channels: int = 7,
                tanh_constant: float = 150.0, insert_pad: bool = True, use_instance_norm: bool = True,
                recurrent_gain: float = 1.0, padding_type: str = "reflect-start") -> TorchObject:
    """Mirror of models_video.lua:55-140 for every padding_type of train_video.lua:25 (default 'reflect-start', use_instance_norm=1).

    padding_type (models_video.lua:10-53,65-80):
      'reflect-start' / 'none'  residual blocks unpadded with nn.ShaveImage(2) on the skip; 'reflect-start' additionally gets ONE
                                nn.SpatialReflectionPadding at the front (train_video.lua:319-325).  The c-layers keep their zero padding
                                in both (the `elseif padding_type == 'none'` of :76 tests an undefined global and never fires);
      'reflect' / 'replicate'   a padding layer of (f-1)/2 in front of every c-convolution (which then has padW = 0) and of 1 in front of
                                both convolutions of every residual block; nn.Identity on the skip;
      'zero'                    residual-block convolutions with padW = 1, nn.Identity on the skip.
    The d / u layers always carry their own zero padding (:90-93,99-102).

    recurrent_gain scales the first convolution's weights on input channels 4-7 (1-based: the warped, masked previous output and the
    certainty plane, fast_artistic_video_core.lua:166-171).  Random-init weights make frame -> frame an expanding map (any perturbation
    of the previous output grows ~3x per frame), which no trained model with the temporal-consistency loss does; a gain well below 1
    gives a CONTRACTIVE synthetic checkpoint on which free-running whole-clip parity is a meaningful gate."""
    assert padding_type in PADDING_TYPES, padding_type
    rng = np.random.default_rng(seed)
    mods = []
    prev = in_channels
    items = arch.split(",")
    n_res, down, res_down = 0, 1, 1
    for i, v in enumerate(items):
        needs_bn = needs_relu = True
        c0 = v[0]
        if c0 == "c":
            f, s, nxt = int(v[1]), int(v[3]), int(v[5:])
            p = (f - 1) // 2
            if padding_type in ("reflect", "replicate"):                    # :70-75
                mods.append(_padlayer(padding_type, p)); p = 0
            mods.append(_conv(rng, prev, nxt, f, s, p))                     # :65-80 (otherwise the zero pad stays)
            if i == 0 and recurrent_gain != 1.0 and prev == 7:
                mods[-1].fields["weight"][:, 3:7] *= np.float32(recurrent_gain)
        elif c0 == "d":
            nxt = int(v[1:]); mods.append(_conv(rng, prev, nxt, 3, 2, 1)); down *= 2   # :90-93
        elif c0 == "U":
            nxt = prev; mods.append(_simple("nn.SpatialUpSamplingNearest", scale_factor=int(v[1:])))  # :94-98
        elif c0 == "u":
            nxt = int(v[1:]); mods.append(_fullconv(rng, prev, nxt, 3, 2, 1, 1)); down //= 2          # :99-102
        elif c0 == "R":
            nxt = int(v[1:]); n_res += 1; res_down = down
            norm = _inorm if use_instance_norm else _bnorm
            padded = padding_type in ("reflect", "replicate")
            pc = 1 if padding_type == "zero" else 0
            block = ([_padlayer(padding_type, 1)] if padded else []) + [_conv(rng, nxt, nxt, 3, 1, pc), norm(rng, nxt), _simple("nn.ReLU", inplace=True)] + \
                    ([_padlayer(padding_type, 1)] if padded else []) + [_conv(rng, nxt, nxt, 3, 1, pc), norm(rng, nxt)]       # :10-39
            skip = _simple("nn.ShaveImage", size=2) if padding_type in ("none", "reflect-start") else _simple("nn.Identity")      # :45-49
            concat = TorchObject("nn.ConcatTable", {"modules": [_sequential(block), skip]})
            mods.append(_sequential([concat, _simple("nn.CAddTable", inplace=False)]))            # :41-53
            needs_bn = needs_relu = False
        else:
            raise ValueError(f"arch item {v!r} is outside the hot-path scope")
        if i == len(items) - 1:
            needs_bn = needs_relu = False                                                        # :117-120
        if needs_bn: mods.append(_inorm(rng, nxt) if use_instance_norm else _bnorm(rng, nxt))
        if needs_relu: mods.append(_simple("nn.ReLU", inplace=True))
        prev = nxt
    mods.append(_simple("nn.Tanh"))
    mods.append(_simple("nn.MulConstant", constant_scalar=float(tanh_constant), inplace=False))
    mods.append(_simple("nn.TotalVariation", strength=1e-6))
    if insert_pad and n_res and padding_type == "reflect-start":
        p = 2 * n_res * res_down  # train_video.lua:319-325: 2 px/side/block at 1/res_down res (40 for 5 blocks at 1/4)
        mods.insert(0, TorchObject("nn.SpatialReflectionPadding",
                                   {"pad_l": p, "pad_r": p, "pad_t": p, "pad_b": p,
                                    "_type": "torch.FloatTensor", "train": False}))
    return _sequential(mods)


def make_synthetic_checkpoint(path: str, arch: str = CANONICAL_ARCH, seed: int = 0, **kw) -> None:
    model = build_model(arch, seed, **kw)
    ckpt = {"opt": {"arch": arch, "padding_type": kw.get("padding_type", "reflect-start"), "use_instance_norm": 1,
                    "tanh_constant": kw.get("tanh_constant", 150.0)},
            "train_loss_history": {}, "val_loss_history": {}, "iter": 0, "model": model}
    write_checkpoint(path, ckpt)
