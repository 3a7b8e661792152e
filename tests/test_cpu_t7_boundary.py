"""The .t7 boundary (fast_artistic_video_core.lua:38-57 loads what train_video.lua:507-534 saves) driven by a SECOND,
independent emitter of the Torch7 File.lua binary grammar -- not fav_amd.t7's writer, whose author and recalled grammar the C++
reader shares.  This emitter produces the things a checkpoint written by a real Torch7 would contain and the sibling writer never
does: legacy version-0 objects, Double / Cuda typed tensors, cudnn.* and nn.SpatialConvolutionMM class names (2-D weights),
all tensors as strided views with non-zero offsets into shared storages (storages back-referenced), modules referenced twice,
closures (function objects) and nested junk in `opt`, gradWeight / output / _type / train fields, InstanceNormalization's nested
`bn`.  The product's C++ reader (csrc/t7_reader.cpp, through the C ABI: fav_t7_describe_host / fav_net_pack_host) must recover
the exact architecture and bit-exact float weights.  No Torch7 exists offline, so the grammar itself stays [recalled]
(INTEGRATION.md); what this pins is that the reader implements that grammar in general, not one writer's subset of it."""
import ctypes as C
import os
import struct

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import fav_amd

L_PAD, L_CONV, L_IN, L_RELU, L_RES, L_UP, L_TANH, L_MUL, L_IDENTITY, L_BN = range(10)


# ------------------------------------------------------------------------------------------------ emitter
class Emitter:
    """Torch7 binary serialiser (little-endian): int32 type tag; number = f64; string = int32 len + bytes; boolean = int32;
    table / torch object / function carry an int32 index, a repeated index is a back-reference without a body."""

    def __init__(self, rng, legacy=False, double=False, cuda=False):
        self.out = bytearray(); self.next_index = 1; self.rng = rng
        self.legacy, self.double, self.cuda = legacy, double, cuda
        self.store = None       # (index, ndarray) of the shared storage once written
        self.pool = []          # tensors to lay out in the shared storage

    # primitives
    def i32(self, v): self.out += struct.pack("<i", v)
    def i64(self, v): self.out += struct.pack("<q", v)
    def string_body(self, s): b = s.encode(); self.i32(len(b)); self.out += b
    def nil(self): self.i32(0)
    def number(self, v): self.i32(1); self.out += struct.pack("<d", float(v))
    def string(self, s): self.i32(2); self.string_body(s)
    def boolean(self, v): self.i32(5); self.i32(1 if v else 0)

    def new_index(self):
        k = self.next_index; self.next_index += 1; return k

    def class_header(self, tag_index, cls):
        self.i32(4); self.i32(tag_index)
        if not self.legacy: self.string_body("V 1")
        self.string_body(cls)                       # legacy (version 0): the class name stands where the version string would

    def table(self, items):
        """items: list of (key emitter thunk, value emitter thunk)"""
        self.i32(3); self.i32(self.new_index()); self.i32(len(items))
        for k, v in items:
            k(); v()

    def function(self):
        # [recalled, File.lua] TYPE_FUNCTION (6, the legacy record) has NO memo index: size + dumped bytes + upvalues; only the
        # RECUR_FUNCTION records (7 legacy, 8) carry one
        tag = int(self.rng.choice([6, 7, 8]))
        self.i32(tag)
        if tag != 6: self.i32(self.new_index())
        self.string_body("\x1bLJ\x02 dumped bytecode")
        self.table([(lambda: self.number(1), lambda: self.string("upvalue"))])

    # tensors: every tensor is a view into ONE storage that is written with the first tensor and back-referenced afterwards
    def plan(self, arrays):
        """lay the arrays out in one storage with random gaps; remember offsets"""
        self.layout = {}; pos = int(self.rng.integers(0, 5))
        for key, a in arrays.items():
            self.layout[key] = pos; pos += a.size + int(self.rng.integers(0, 7))
        dt = np.float64 if self.double else np.float32
        buf = (self.rng.standard_normal(pos + 3) * 7).astype(dt)          # junk between the views
        for key, a in arrays.items():
            buf[self.layout[key]:self.layout[key] + a.size] = a.astype(dt).ravel()
        self.storage_data = buf

    def tensor(self, key, shape, transposed_view=False):
        """a view of the planned array `key`; transposed_view: dims stored permuted with matching strides (non-contiguous)"""
        tname = "torch.CudaTensor" if self.cuda else ("torch.DoubleTensor" if self.double else "torch.FloatTensor")
        sname = tname.replace("Tensor", "Storage")
        self.class_header(self.new_index(), tname)
        shape = list(shape)
        strides = [int(np.prod(shape[i + 1:])) for i in range(len(shape))]
        self.i32(len(shape))
        for s in shape: self.i64(s)
        for s in strides: self.i64(s)
        self.i64(self.layout[key] + 1)              # storageOffset, 1-based
        if self.store is None:
            idx = self.new_index(); self.store = idx
            self.class_header(idx, sname)
            self.i64(self.storage_data.size); self.out += self.storage_data.tobytes()
        else:
            self.i32(4); self.i32(self.store)       # back-reference: no body

    def empty_tensor(self):
        self.class_header(self.new_index(), "torch.FloatTensor")
        self.i32(0); self.i64(1); self.nil()        # no dimensions, no storage

    def module(self, cls, fields):
        self.class_header(self.new_index(), cls)
        junk = [(lambda: self.string("_type"), lambda: self.string("torch.FloatTensor")),
                (lambda: self.string("train"), lambda: self.boolean(False)),
                (lambda: self.string("output"), self.empty_tensor), (lambda: self.string("gradInput"), self.empty_tensor)]
        if self.rng.random() < 0.3:
            junk.append((lambda: self.string("hook"), self.function))
        items = [(lambda k=k: self.string(k), v) for k, v in fields]
        allf = items + junk
        order = self.rng.permutation(len(allf))
        self.table([allf[i] for i in order])


def emit_layers(e, layers, prefix=""):
    """emit the `modules` array of an nn.Sequential; returns thunks"""
    thunks = []
    relu_index = [None]
    for n, L in enumerate(layers):
        key = f"{prefix}{n}"
        t = L["t"]
        if t == "conv":
            def th(L=L, key=key):
                cls = L["cls"]
                wshape = (L["cout"], L["cin"] * L["k"] * L["k"]) if cls == "nn.SpatialConvolutionMM" else (L["cout"], L["cin"], L["k"], L["k"])
                f = [("nInputPlane", lambda: e.number(L["cin"])), ("nOutputPlane", lambda: e.number(L["cout"])), ("kW", lambda: e.number(L["k"])),
                     ("kH", lambda: e.number(L["k"])), ("dW", lambda: e.number(L["s"])), ("dH", lambda: e.number(L["s"])),
                     ("padW", lambda: e.number(L["p"])), ("padH", lambda: e.number(L["p"])),
                     ("weight", lambda: e.tensor(key + "w", wshape)), ("gradWeight", lambda: e.tensor(key + "gw", wshape))]
                if L["bias"]:
                    f += [("bias", lambda: e.tensor(key + "b", (L["cout"],))), ("gradBias", lambda: e.tensor(key + "gb", (L["cout"],)))]
                if cls.startswith("cudnn."):
                    f += [("groups", lambda: e.number(1))]
                e.module(cls, f)
        elif t == "full":
            def th(L=L, key=key):
                f = [("nInputPlane", lambda: e.number(L["cin"])), ("nOutputPlane", lambda: e.number(L["cout"])), ("kW", lambda: e.number(L["k"])),
                     ("kH", lambda: e.number(L["k"])), ("dW", lambda: e.number(2)), ("dH", lambda: e.number(2)), ("padW", lambda: e.number(1)),
                     ("padH", lambda: e.number(1)), ("adjW", lambda: e.number(1)), ("adjH", lambda: e.number(1)),
                     ("weight", lambda: e.tensor(key + "w", (L["cin"], L["cout"], L["k"], L["k"]))), ("bias", lambda: e.tensor(key + "b", (L["cout"],)))]
                e.module(L["cls"], f)
        elif t == "in":
            def th(L=L, key=key):
                bn = lambda: e.module("nn.SpatialBatchNormalization", [("eps", lambda: e.number(1e-5)), ("momentum", lambda: e.number(0.1)),
                                                                       ("weight", lambda: e.tensor(key + "bnw", (L["c"],))), ("affine", lambda: e.boolean(True))])
                e.module("nn.InstanceNormalization", [("eps", lambda: e.number(L["eps"])), ("nOutput", lambda: e.number(L["c"])), ("prev_N", lambda: e.number(1)),
                                                      ("weight", lambda: e.tensor(key + "g", (L["c"],))), ("bias", lambda: e.tensor(key + "be", (L["c"],))),
                                                      ("gradWeight", lambda: e.tensor(key + "gg", (L["c"],))), ("bn", bn)])
        elif t == "bn":
            def th(L=L, key=key):
                e.module(L["cls"], [("eps", lambda: e.number(L["eps"])), ("running_mean", lambda: e.tensor(key + "m", (L["c"],))),
                                    ("running_var", lambda: e.tensor(key + "v", (L["c"],))), ("weight", lambda: e.tensor(key + "g", (L["c"],))),
                                    ("bias", lambda: e.tensor(key + "be", (L["c"],)))])
        elif t == "relu":
            def th(L=L):
                # one ReLU instance shared by every position that uses it: the second occurrence is a bare back-reference
                if relu_index[0] is None:
                    relu_index[0] = e.next_index
                    e.module(L["cls"], [("inplace", lambda: e.boolean(True)), ("threshold", lambda: e.number(0)), ("val", lambda: e.number(0))])
                else:
                    e.i32(4); e.i32(relu_index[0])
        elif t == "pad":
            def th(L=L):
                e.module(L.get("cls", "nn.SpatialReflectionPadding"), [(k, (lambda v=L["p"]: e.number(v))) for k in ("pad_l", "pad_r", "pad_t", "pad_b")])
        elif t == "up":
            def th(L=L):
                e.module("nn.SpatialUpSamplingNearest", [("scale_factor", lambda: e.number(2)), ("inputSize", lambda: e.nil())])
        elif t == "tanh":
            def th(L=L): e.module(L["cls"], [])
        elif t == "mul":
            def th(L=L): e.module("nn.MulConstant", [("constant_scalar", lambda: e.number(L["v"])), ("inplace", lambda: e.boolean(False))])
        elif t == "tv":
            def th(L=L): e.module("nn.TotalVariation", [("strength", lambda: e.number(1e-6))])
        elif t == "res":
            def th(L=L, key=key):
                inner = emit_layers(e, L["block"], key + "_")
                branch = lambda: e.module("nn.Sequential", [("modules", lambda: e.table([(lambda i=i: e.number(i + 1), t2) for i, t2 in enumerate(inner)]))])
                shave = (lambda: e.module("nn.ShaveImage", [("size", lambda: e.number(L["shave"]))])) if L["shave"] else (lambda: e.module("nn.Identity", []))
                concat = lambda: e.module("nn.ConcatTable", [("modules", lambda: e.table([(lambda: e.number(1), branch), (lambda: e.number(2), shave)]))])
                cadd = lambda: e.module("nn.CAddTable", [("inplace", lambda: e.boolean(False))])
                e.module("nn.Sequential", [("modules", lambda: e.table([(lambda: e.number(1), concat), (lambda: e.number(2), cadd)]))])
        thunks.append(th)
    return thunks


def collect_arrays(layers, rng, prefix=""):
    arrays = {}
    for n, L in enumerate(layers):
        key = f"{prefix}{n}"; t = L["t"]
        if t in ("conv", "full"):
            arrays[key + "w"] = rng.standard_normal(L["cin"] * L["cout"] * L["k"] * L["k"]).astype(np.float32)
            arrays[key + "gw"] = rng.standard_normal(L["cin"] * L["cout"] * L["k"] * L["k"]).astype(np.float32)
            if L.get("bias", True):
                arrays[key + "b"] = rng.standard_normal(L["cout"]).astype(np.float32); arrays[key + "gb"] = rng.standard_normal(L["cout"]).astype(np.float32)
        elif t == "in":
            for sfx in ("g", "be", "gg", "bnw"): arrays[key + sfx] = rng.standard_normal(L["c"]).astype(np.float32)
        elif t == "bn":
            for sfx in ("m", "g", "be"): arrays[key + sfx] = rng.standard_normal(L["c"]).astype(np.float32)
            arrays[key + "v"] = (rng.random(L["c"]) + 0.5).astype(np.float32)
        elif t == "res":
            arrays.update(collect_arrays(L["block"], rng, key + "_"))
    return arrays


def write_checkpoint(path, layers, rng, legacy=False, double=False, cuda=False, bare=False):
    e = Emitter(rng, legacy, double, cuda)
    arrays = collect_arrays(layers, rng)
    e.plan(arrays)
    thunks = emit_layers(e, layers)
    model = lambda: e.module("nn.Sequential", [("modules", lambda: e.table([(lambda i=i: e.number(i + 1), t) for i, t in enumerate(thunks)]))])
    if bare:
        model()
    else:   # train_video.lua:508-534: {opt=..., *_history=..., iter=t, model=...}
        opt = lambda: e.table([(lambda: e.string("arch"), lambda: e.string("c9s1-32,d64")), (lambda: e.string("use_cudnn"), lambda: e.number(1)),
                               (lambda: e.string("nested"), lambda: e.table([(lambda: e.number(1), lambda: e.boolean(True)), (lambda: e.number(2), e.function)]))])
        hist = lambda: e.table([(lambda i=i: e.number(i + 1), lambda i=i: e.number(0.5 * i)) for i in range(5)])
        e.table([(lambda: e.string("opt"), opt), (lambda: e.string("train_loss_history"), hist), (lambda: e.string("iter"), lambda: e.number(40000)),
                 (lambda: e.string("model"), model), (lambda: e.string("val_loss_history"), hist)])
    with open(path, "wb") as f:
        f.write(bytes(e.out))
    return arrays


# ------------------------------------------------------------------------------------------------ what the C++ reader made of it
def parse_blob(blob):
    pos = [0]
    def i32(): v = struct.unpack_from("<i", blob, pos[0])[0]; pos[0] += 4; return v
    def f32(): v = struct.unpack_from("<f", blob, pos[0])[0]; pos[0] += 4; return v
    def vec():
        n = i32(); a = np.frombuffer(blob, np.float32, n, pos[0]).copy(); pos[0] += 4 * n; return a
    assert i32() == 0x42564146 and i32() == 3
    def layers():
        out = []
        for _ in range(i32()):
            L = {"type": i32()}
            L["pads"] = [i32() for _ in range(4)]; L["pad_mode"] = i32()
            L["cin"], L["cout"], L["k"], L["stride"], L["pad"] = (i32() for _ in range(5))
            L["scale"], L["shave"] = i32(), i32(); L["mul"], L["eps"] = f32(), f32(); L["transposed"], L["adj"] = i32(), i32()
            for k in ("w", "b", "gamma", "beta", "mean", "var"): L[k] = vec()
            L["block"] = layers()
            out.append(L)
        return out
    ls = layers()
    assert pos[0] == len(blob)
    return ls


def check(parsed, layers, arrays, dtype, prefix=""):
    cast = lambda a: a.astype(dtype).astype(np.float32)       # what a Double checkpoint holds, read back as float
    assert len(parsed) == len(layers), (len(parsed), len(layers))
    for n, (P, L) in enumerate(zip(parsed, layers)):
        key = f"{prefix}{n}"; t = L["t"]
        if t in ("conv", "full"):
            assert P["type"] == L_CONV and (P["cin"], P["cout"], P["k"]) == (L["cin"], L["cout"], L["k"]) and P["transposed"] == (t == "full")
            assert (P["stride"], P["pad"]) == ((L["s"], L["p"]) if t == "conv" else (2, 1)) and (t == "conv" or P["adj"] == 1)
            assert np.array_equal(P["w"], cast(arrays[key + "w"]))
            assert np.array_equal(P["b"], cast(arrays[key + "b"])) if L.get("bias", True) else P["b"].size == 0
        elif t == "in":
            assert P["type"] == L_IN and np.array_equal(P["gamma"], cast(arrays[key + "g"])) and np.array_equal(P["beta"], cast(arrays[key + "be"]))
            assert P["eps"] == float(np.float32(L["eps"]))
        elif t == "bn":
            assert P["type"] == L_BN and np.array_equal(P["mean"], cast(arrays[key + "m"])) and np.array_equal(P["var"], cast(arrays[key + "v"]))
            assert np.array_equal(P["gamma"], cast(arrays[key + "g"])) and np.array_equal(P["beta"], cast(arrays[key + "be"]))
        elif t == "relu": assert P["type"] == L_RELU
        elif t == "pad": assert P["type"] == L_PAD and P["pads"] == [L["p"]] * 4 and P["pad_mode"] == (1 if L.get("cls") == "nn.SpatialReplicationPadding" else 0)
        elif t == "up": assert P["type"] == L_UP and P["scale"] == 2
        elif t == "tanh": assert P["type"] == L_TANH
        elif t == "mul": assert P["type"] == L_MUL and abs(P["mul"] - L["v"]) < 1e-6
        elif t == "tv": assert P["type"] == L_IDENTITY
        elif t == "res":
            assert P["type"] == L_RES and P["shave"] == L["shave"]
            check(P["block"], L["block"], arrays, dtype, key + "_")


def canonical_spec(dims=(8, 16, 32), cudnn=False):
    cv = "cudnn.SpatialConvolution" if cudnn else "nn.SpatialConvolution"
    relu, tanh = ("cudnn.ReLU", "cudnn.Tanh") if cudnn else ("nn.ReLU", "nn.Tanh")
    conv = lambda cin, cout, k, s, p, cls=cv: {"t": "conv", "cls": cls, "cin": cin, "cout": cout, "k": k, "s": s, "p": p, "bias": True}
    IN = lambda c: {"t": "in", "c": c, "eps": 1e-5}
    R = {"t": "relu", "cls": relu}
    a, b, c = dims
    block = lambda d: {"t": "res", "shave": 2, "block": [conv(d, d, 3, 1, 0), IN(d), R, conv(d, d, 3, 1, 0, "nn.SpatialConvolutionMM"), IN(d)]}
    return [{"t": "pad", "p": 40}, conv(7, a, 9, 1, 4), IN(a), R, conv(a, b, 3, 2, 1), IN(b), R, conv(b, c, 3, 2, 1), IN(c), R, block(c), block(c),
            {"t": "up"}, IN(c), R, conv(c, b, 3, 1, 1), IN(b), R, {"t": "up"}, IN(b), R, conv(b, 3, 9, 1, 4), {"t": "tanh", "cls": tanh},
            {"t": "mul", "v": 150.0}, {"t": "tv"}]


@pytest.mark.parametrize("legacy,double,cuda,cudnn,bare", [(False, False, False, False, False), (True, False, False, False, False),
                                                            (False, True, False, False, False), (False, False, True, True, False),
                                                            (True, True, False, True, True)])
def test_reader_on_independent_emitter(favlib, tmp_path, legacy, double, cuda, cudnn, bare):
    """the architecture models_video.lua:55-140 builds, as a second emitter writes it: version-0 headers, Double / Cuda tensors,
    cudnn.* and SpatialConvolutionMM classes, shared storage with offsets, a shared ReLU instance, closures and junk fields"""
    rng = np.random.default_rng(11)
    spec = canonical_spec(cudnn=cudnn)
    p = str(tmp_path / "m.t7")
    arrays = write_checkpoint(p, spec, rng, legacy=legacy, double=double, cuda=cuda, bare=bare)
    text = favlib.describe_t7(p)
    assert text.splitlines()[0] == "pad 40 40 40 40" and text.count("res shave=2") == 2 and "conv 7 8 9 1 4 bias=1" in text
    check(parse_blob(favlib.pack_checkpoint(p)), spec, arrays, np.float64 if double else np.float32)


@pytest.mark.parametrize("ptype", ["reflect", "replicate", "zero"])
def test_reader_on_the_other_padding_types(favlib, tmp_path, ptype):
    """train_video.lua:25 -padding_type reflect | replicate | zero (models_video.lua:10-53,65-80): a padding MODULE in front of every
    c-convolution and of both convolutions of a residual block (nn.SpatialReflectionPadding / nn.SpatialReplicationPadding, the
    convolutions then carry padW = 0) or zero-padded block convolutions, nn.Identity on the skip -- read back with the right mode"""
    rng = np.random.default_rng(3)
    conv = lambda cin, cout, k, s, p: {"t": "conv", "cls": "nn.SpatialConvolution", "cin": cin, "cout": cout, "k": k, "s": s, "p": p, "bias": True}
    IN = lambda c: {"t": "in", "c": c, "eps": 1e-5}
    R = {"t": "relu", "cls": "nn.ReLU"}
    cls = {"reflect": "nn.SpatialReflectionPadding", "replicate": "nn.SpatialReplicationPadding"}.get(ptype)
    P = (lambda p: [{"t": "pad", "p": p, "cls": cls}]) if cls else (lambda p: [])
    pc = 1 if ptype == "zero" else 0
    block = {"t": "res", "shave": 0, "block": P(1) + [conv(16, 16, 3, 1, pc), IN(16), R] + P(1) + [conv(16, 16, 3, 1, pc), IN(16)]}
    spec = P(4) + [conv(7, 8, 9, 1, 0 if cls else 4), IN(8), R, conv(8, 16, 3, 2, 1), IN(16), R, block, {"t": "up"}, IN(16), R] + P(1) + \
           [conv(16, 8, 3, 1, 0 if cls else 1), IN(8), R] + P(4) + [conv(8, 3, 9, 1, 0 if cls else 4), {"t": "tanh", "cls": "nn.Tanh"}, {"t": "mul", "v": 150.0}, {"t": "tv"}]
    p = str(tmp_path / "m.t7")
    arrays = write_checkpoint(p, spec, rng)
    text = favlib.describe_t7(p)
    want_first = {"reflect": "pad 4 4 4 4", "replicate": "replicate-pad 4 4 4 4", "zero": "conv 7 8 9 1 4 bias=1"}[ptype]
    assert text.splitlines()[0] == want_first and "res shave=0" in text
    assert text.count("replicate-pad 1 1 1 1") == (3 if ptype == "replicate" else 0)
    check(parse_blob(favlib.pack_checkpoint(p)), spec, arrays, np.float32)


layer_st = st.deferred(lambda: st.one_of(
    st.builds(lambda cin, cout, k, s, p, cls, bias: {"t": "conv", "cls": cls, "cin": cin, "cout": cout, "k": k, "s": s, "p": p, "bias": bias},
              st.integers(1, 6), st.integers(1, 6), st.sampled_from([1, 3, 5]), st.integers(1, 2), st.integers(0, 2),
              st.sampled_from(["nn.SpatialConvolution", "cudnn.SpatialConvolution", "nn.SpatialConvolutionMM"]), st.booleans()),
    st.builds(lambda cin, cout, cls: {"t": "full", "cls": cls, "cin": cin, "cout": cout, "k": 3},
              st.integers(1, 5), st.integers(1, 5), st.sampled_from(["nn.SpatialFullConvolution", "cudnn.SpatialFullConvolution"])),
    st.builds(lambda c, eps: {"t": "in", "c": c, "eps": eps}, st.integers(1, 7), st.sampled_from([1e-5, 1e-3])),
    st.builds(lambda c, cls: {"t": "bn", "c": c, "eps": 1e-5, "cls": cls}, st.integers(1, 7), st.sampled_from(["nn.SpatialBatchNormalization", "cudnn.SpatialBatchNormalization"])),
    st.sampled_from([{"t": "relu", "cls": "nn.ReLU"}, {"t": "relu", "cls": "cudnn.ReLU"}, {"t": "up"}, {"t": "tanh", "cls": "nn.Tanh"}, {"t": "tv"},
                     {"t": "mul", "v": 150.0}, {"t": "pad", "p": 3}, {"t": "pad", "p": 1, "cls": "nn.SpatialReplicationPadding"}]),
    st.builds(lambda block, shave: {"t": "res", "shave": shave, "block": block}, st.lists(layer_st, min_size=1, max_size=3), st.sampled_from([0, 2]))))


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(layers=st.lists(layer_st, min_size=1, max_size=7), seed=st.integers(0, 2 ** 16), legacy=st.booleans(), double=st.booleans(), bare=st.booleans())
def test_reader_on_random_module_trees(favlib, tmp_path, layers, seed, legacy, double, bare):
    """hypothesis-generated module trees (any order; the reader parses, the network builder judges the architecture later)"""
    rng = np.random.default_rng(seed)
    p = str(tmp_path / "h.t7")
    arrays = write_checkpoint(p, layers, rng, legacy=legacy, double=double, bare=bare)
    check(parse_blob(favlib.pack_checkpoint(p)), layers, arrays, np.float64 if double else np.float32)


def test_reader_rejects_what_it_cannot_bound(favlib, tmp_path):
    """crafted files: a storage count near 2^62, tensor sizes whose product overflows, a container that contains itself, an
    offset outside the storage -- all must come back as a status (in a child process: a crash would be a test failure)"""
    import subprocess, sys, textwrap
    rng = np.random.default_rng(5)
    spec = [{"t": "conv", "cls": "nn.SpatialConvolution", "cin": 2, "cout": 2, "k": 1, "s": 1, "p": 0, "bias": True}]
    good = str(tmp_path / "g.t7"); write_checkpoint(good, spec, rng, bare=True)
    data = bytearray(open(good, "rb").read())
    cases = []
    # (1) storage element count -> 2^62 - 1: find the storage header (class name + int64 count)
    k = data.find(b"torch.FloatStorage") + len(b"torch.FloatStorage")
    bad = bytearray(data); bad[k:k + 8] = struct.pack("<q", (1 << 62) - 1); cases.append(bytes(bad))
    bad = bytearray(data); bad[k:k + 8] = struct.pack("<q", -5); cases.append(bytes(bad))
    # (2) first tensor: sizes -> huge (product overflows int64), and a negative size, and a far offset
    wkey = data.find(struct.pack("<i", 6) + b"weight")                       # the `weight` field's key string; its value (the tensor) follows
    t = data.find(b"torch.FloatTensor", wkey) + len(b"torch.FloatTensor")    # int32 nDim follows
    nd = struct.unpack_from("<i", data, t)[0]
    for vals in ([1 << 40] * nd, [-3] + [1] * (nd - 1)):
        bad = bytearray(data); bad[t + 4:t + 4 + 8 * nd] = b"".join(struct.pack("<q", v) for v in vals); cases.append(bytes(bad))
    bad = bytearray(data); bad[t + 4 + 16 * nd:t + 12 + 16 * nd] = struct.pack("<q", 1 << 50); cases.append(bytes(bad))      # storageOffset
    # (3) a Sequential whose `modules` table holds a back-reference to the Sequential itself
    e = Emitter(rng)
    seq_index = e.next_index
    e.module("nn.Sequential", [("modules", lambda: e.table([(lambda: e.number(1), lambda: (e.i32(4), e.i32(seq_index)))]))])
    cases.append(bytes(e.out))
    paths = []
    for n, c in enumerate(cases):
        pth = str(tmp_path / f"bad{n}.t7"); open(pth, "wb").write(c); paths.append(pth)
    script = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {os.path.dirname(os.path.dirname(fav_amd.__file__))!r})
        import fav_amd
        for p in {paths!r}:
            try:
                fav_amd.pack_checkpoint(p); print("parsed", p)
            except fav_amd.FavError as ex:
                assert str(ex)
                print("rejected")
    """)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    assert r.stdout.count("rejected") == len(cases), r.stdout
