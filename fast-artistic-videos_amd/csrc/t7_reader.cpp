// t7_reader.cpp -- host-side reader for Torch7 binary checkpoints and the flat weight blob.
//
// Replaces `torch.load(path).model` (fast_artistic_video_core.lua:39-46).  The checkpoint layout is
// what train_video.lua:508-534 writes: a table with field `model` holding an nn.Sequential whose
// modules are those models_video.lua:55-140 builds (plus the lazily inserted
// nn.SpatialReflectionPadding of train_video.lua:319-325).  Grammar: Torch7 File.lua binary format
// (not in the reference tree; restated in SURVEY.md Appendix B).  Unknown fields (gradWeight, bn,
// output, _type, train, ...) are parsed and ignored.
#include <climits>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>

#include "fav_internal.h"

namespace fav {
namespace {

struct Val;
using VP = std::shared_ptr<Val>;

struct Val {
    enum Kind { NIL, NUM, STR, BOOL, TABLE, OBJECT, TENSOR, STORAGE, FUNC } kind = NIL;
    double num = 0;
    std::string str;                        // STR value / class name
    bool b = false;
    std::vector<std::pair<VP, VP>> items;   // TABLE
    VP body;                                // OBJECT: its field table
    // TENSOR
    std::vector<long long> size, stride;
    long long offset = 0;
    VP storage;
    // STORAGE
    int elem = 0;                           // bytes per element
    char stype = 'f';                       // f,d,l,i,b,c,s
    std::vector<uint8_t> raw;
};

struct Reader {
    const uint8_t* d; size_t n; size_t p = 0;
    std::map<int, VP> memo;
    bool ok = true;
    std::string err;

    // k > n - p, not p + k > n: a crafted 2^62-element storage must not wrap the comparison
    bool need(size_t k) { if (k > n - p) { ok = false; if (err.empty()) err = "unexpected end of file"; return false; } return true; }
    int i32() { if (!need(4)) return 0; int v; memcpy(&v, d + p, 4); p += 4; return v; }
    long long i64() { if (!need(8)) return 0; long long v; memcpy(&v, d + p, 8); p += 8; return v; }
    double f64() { if (!need(8)) return 0; double v; memcpy(&v, d + p, 8); p += 8; return v; }
    std::string rawstr() {
        int len = i32();
        if (len < 0 || !need((size_t)len)) { ok = false; if (err.empty()) err = "bad string length"; return ""; }
        std::string s(reinterpret_cast<const char*>(d + p), (size_t)len); p += (size_t)len; return s;
    }

    static bool storage_info(const std::string& cls, int& elem, char& t) {
        if (cls == "torch.FloatStorage") { elem = 4; t = 'f'; return true; }
        if (cls == "torch.DoubleStorage") { elem = 8; t = 'd'; return true; }
        if (cls == "torch.LongStorage") { elem = 8; t = 'l'; return true; }
        if (cls == "torch.IntStorage") { elem = 4; t = 'i'; return true; }
        if (cls == "torch.ByteStorage") { elem = 1; t = 'b'; return true; }
        if (cls == "torch.CharStorage") { elem = 1; t = 'c'; return true; }
        if (cls == "torch.ShortStorage") { elem = 2; t = 's'; return true; }
        // cutorch storages are serialised with their host element type (a model saved without :float())
        if (cls == "torch.CudaStorage") { elem = 4; t = 'f'; return true; }
        if (cls == "torch.CudaDoubleStorage") { elem = 8; t = 'd'; return true; }
        if (cls == "torch.CudaLongStorage") { elem = 8; t = 'l'; return true; }
        if (cls == "torch.CudaIntStorage") { elem = 4; t = 'i'; return true; }
        if (cls == "torch.CudaByteStorage") { elem = 1; t = 'b'; return true; }
        if (cls == "torch.CudaCharStorage") { elem = 1; t = 'c'; return true; }
        if (cls == "torch.CudaShortStorage") { elem = 2; t = 's'; return true; }
        return false;
    }
    static bool is_tensor(const std::string& cls) {
        return cls.size() > 6 && cls.compare(0, 6, "torch.") == 0 && cls.compare(cls.size() - 6, 6, "Tensor") == 0;
    }

    VP obj(int depth = 0) {
        VP v = std::make_shared<Val>();
        if (!ok) return v;
        if (depth > 256) { ok = false; err = "nesting too deep"; return v; }
        const int t = i32();
        switch (t) {
        case 0: return v;
        case 1: v->kind = Val::NUM; v->num = f64(); return v;
        case 2: v->kind = Val::STR; v->str = rawstr(); return v;
        case 5: v->kind = Val::BOOL; v->b = i32() != 0; return v;
        case 3: {
            const int idx = i32();
            auto it = memo.find(idx);
            if (it != memo.end()) return it->second;
            v->kind = Val::TABLE; memo[idx] = v;
            const int cnt = i32();
            if (cnt < 0) { ok = false; err = "negative table size"; return v; }
            for (int i = 0; i < cnt && ok; ++i) { VP k = obj(depth + 1); VP x = obj(depth + 1); v->items.emplace_back(k, x); }
            return v;
        }
        case 4: {
            const int idx = i32();
            auto it = memo.find(idx);
            if (it != memo.end()) return it->second;
            memo[idx] = v;
            std::string version = rawstr();
            std::string cls = version.compare(0, 2, "V ") == 0 ? rawstr() : version;
            v->str = cls;
            int elem; char st;
            if (storage_info(cls, elem, st)) {
                v->kind = Val::STORAGE; v->elem = elem; v->stype = st;
                const long long cnt = i64();
                if (!ok) return v;
                if (cnt < 0 || (unsigned long long)cnt > (unsigned long long)((n - p) / (size_t)elem)) { ok = false; if (err.empty()) err = "bad storage size"; return v; }
                v->raw.assign(d + p, d + p + (size_t)cnt * elem); p += (size_t)cnt * elem;
                return v;
            }
            if (is_tensor(cls)) {
                v->kind = Val::TENSOR;
                const int nd = i32();
                if (nd < 0 || nd > 16) { ok = false; err = "bad tensor rank"; return v; }
                for (int i = 0; i < nd; ++i) v->size.push_back(i64());
                for (int i = 0; i < nd; ++i) v->stride.push_back(i64());
                v->offset = i64();
                if (v->offset != LLONG_MIN) v->offset -= 1;       // 1-based in the file
                v->storage = obj(depth + 1);
                return v;
            }
            v->kind = Val::OBJECT;
            v->body = obj(depth + 1);
            return v;
        }
        case 6: {
            // TYPE_FUNCTION (the legacy record) [recalled, File.lua]: NO memo index -- dumped bytecode (int32 length + bytes),
            // then the upvalue table.  Only the RECUR_FUNCTION records below carry an index.
            v->kind = Val::FUNC;
            (void)rawstr();
            (void)obj(depth + 1);
            return v;
        }
        case 7: case 8: {
            // LEGACY_TYPE_RECUR_FUNCTION / TYPE_RECUR_FUNCTION [recalled, File.lua]: index, dumped bytecode (int32 length +
            // bytes), then the upvalue table.  Closures stored in a checkpoint's `opt` or in a module field are irrelevant to the
            // forward pass: parsed and ignored.
            const int idx = i32();
            auto it = memo.find(idx);
            if (it != memo.end()) return it->second;
            v->kind = Val::FUNC; memo[idx] = v;
            (void)rawstr();
            (void)obj(depth + 1);
            return v;
        }
        default:
            ok = false;
            err = "unsupported .t7 type tag " + std::to_string(t) + " at byte " + std::to_string(p - 4) +
                  " (not a Torch7 File.lua binary object)";
            return v;
        }
    }
};

const Val* field(const Val* o, const char* name)
{
    const Val* tab = o;
    if (o && o->kind == Val::OBJECT) tab = o->body.get();
    if (!tab || tab->kind != Val::TABLE) return nullptr;
    for (auto& kv : tab->items)
        if (kv.first->kind == Val::STR && kv.first->str == name) return kv.second.get();
    return nullptr;
}

bool num_field(const Val* o, const char* name, double& out)
{
    const Val* f = field(o, name);
    if (!f || f->kind != Val::NUM) return false;
    out = f->num; return true;
}

// Lua array part of a table: keys 1..n
std::vector<const Val*> array_items(const Val* tab)
{
    std::map<long long, const Val*> m;
    if (tab && tab->kind == Val::TABLE)
        for (auto& kv : tab->items)
            if (kv.first->kind == Val::NUM && std::floor(kv.first->num) == kv.first->num) m[(long long)kv.first->num] = kv.second.get();
    std::vector<const Val*> out;
    for (auto& kv : m) out.push_back(kv.second);
    return out;
}

// strided read of a Float/Double tensor.  Every size, the element count and the extreme reachable storage offsets are
// validated (overflow-checked) BEFORE anything is allocated: a damaged size field must end in `false`, not in
// std::length_error / bad_alloc crossing the C ABI.
constexpr long long MAX_TENSOR_ELEMS = 1ll << 28;      // 1 GiB of fp32: far above any layer of this model family

bool tensor_to_floats(const Val* t, std::vector<float>& out)
{
    out.clear();
    if (!t || t->kind != Val::TENSOR) return false;
    if (t->size.empty() || !t->storage || t->storage->kind != Val::STORAGE) return true;   // empty tensor
    if (t->stride.size() != t->size.size()) return false;
    const Val* st = t->storage.get();
    if (st->stype != 'f' && st->stype != 'd') return false;
    long long total = 1;
    for (auto s : t->size) {
        if (s < 0) return false;
        if (s == 0) return true;
        if (total > MAX_TENSOR_ELEMS / s) return false;
        total *= s;
    }
    const long long cap = (long long)(st->raw.size() / (size_t)st->elem);
    // lowest / highest element offset the index walk can reach
    __int128 lo = t->offset, hi = t->offset;
    for (size_t k = 0; k < t->size.size(); ++k) {
        const __int128 span = (__int128)(t->size[k] - 1) * t->stride[k];
        if (span < 0) lo += span; else hi += span;
    }
    if (lo < 0 || hi >= cap) return false;
    out.resize((size_t)total);
    std::vector<long long> idx(t->size.size(), 0);
    for (long long i = 0; i < total; ++i) {
        long long off = t->offset;
        for (size_t k = 0; k < idx.size(); ++k) off += idx[k] * t->stride[k];
        const uint8_t* p = st->raw.data() + (size_t)off * st->elem;
        float v;
        if (st->stype == 'f') memcpy(&v, p, 4);
        else { double d; memcpy(&d, p, 8); v = (float)d; }
        out[(size_t)i] = v;
        for (int k = (int)idx.size() - 1; k >= 0; --k) { if (++idx[k] < t->size[k]) break; idx[k] = 0; }
    }
    return true;
}

int extract(const Val* seq, std::vector<Layer>& out, int depth);

// depth: a memoised table may refer to itself (nn.Sequential listed in its own `modules`); the parser shares the node, so the
// walk has to be bounded here
constexpr int MAX_MODULE_DEPTH = 32;

int module_to_layers(const Val* m, std::vector<Layer>& out, int depth)
{
    if (depth > MAX_MODULE_DEPTH) { set_error(".t7: modules nested deeper than %d levels (a container that contains itself?)", MAX_MODULE_DEPTH); return FAV_EFORMAT; }
    if (!m || m->kind != Val::OBJECT) { set_error(".t7: expected an nn module object"); return FAV_EFORMAT; }
    const std::string& c = m->str;
    Layer L;
    double a = 0, b = 0, cc = 0, d = 0;
    if (c == "nn.Sequential") {
        auto mods = array_items(field(m, "modules"));
        if (mods.size() == 2 && mods[0]->kind == Val::OBJECT && mods[0]->str == "nn.ConcatTable" &&
            mods[1]->kind == Val::OBJECT && mods[1]->str == "nn.CAddTable") {
            auto br = array_items(field(mods[0], "modules"));          // models_video.lua:41-53
            if (br.size() != 2) { set_error(".t7: residual ConcatTable must have 2 branches"); return FAV_EUNSUPPORTED; }
            L.type = L_RES;
            if (br[1]->kind == Val::OBJECT && br[1]->str == "nn.ShaveImage") {
                if (!num_field(br[1], "size", a)) { set_error(".t7: ShaveImage without size"); return FAV_EFORMAT; }
                L.shave = (int)a;
            } else if (br[1]->kind == Val::OBJECT && br[1]->str == "nn.Identity") {
                L.shave = 0;
            } else { set_error(".t7: unsupported skip branch"); return FAV_EUNSUPPORTED; }
            if (br[0]->kind != Val::OBJECT || br[0]->str != "nn.Sequential") { set_error(".t7: residual branch must be nn.Sequential"); return FAV_EUNSUPPORTED; }
            int rc = extract(br[0], L.block, depth + 1);
            if (rc) return rc;
            out.push_back(std::move(L));
            return FAV_OK;
        }
        return extract(m, out, depth + 1);
    }
    if (c == "nn.SpatialReflectionPadding" || c == "nn.SpatialReplicationPadding") {
        // (both keep pad_l / pad_r / pad_t / pad_b: SpatialReflectionPadding.lua / SpatialReplicationPadding.lua of `nn` [recalled])
        if (!num_field(m, "pad_l", a) || !num_field(m, "pad_r", b) || !num_field(m, "pad_t", cc) || !num_field(m, "pad_b", d)) {
            set_error(".t7: %s without pad_* fields", c.c_str()); return FAV_EFORMAT; }
        if (a < 0 || b < 0 || cc < 0 || d < 0 || a > 4096 || b > 4096 || cc > 4096 || d > 4096) { set_error(".t7: %s with a negative (cropping) or absurd pad", c.c_str()); return FAV_EUNSUPPORTED; }
        L.type = L_PAD; L.pl = (int)a; L.pr = (int)b; L.pt = (int)cc; L.pb = (int)d; L.pad_mode = c == "nn.SpatialReplicationPadding" ? 1 : 0;
    } else if (c == "nn.SpatialConvolution" || c == "cudnn.SpatialConvolution" || c == "nn.SpatialConvolutionMM") {
        double cin, cout, kw, kh, dw = 1, dh = 1, pw = 0, ph = 0;
        if (!num_field(m, "nInputPlane", cin) || !num_field(m, "nOutputPlane", cout) || !num_field(m, "kW", kw) || !num_field(m, "kH", kh)) {
            set_error(".t7: SpatialConvolution without size fields"); return FAV_EFORMAT; }
        num_field(m, "dW", dw); num_field(m, "dH", dh); num_field(m, "padW", pw); num_field(m, "padH", ph);
        if (kw != kh || dw != dh || pw != ph) { set_error(".t7: non-square convolution geometry is unsupported"); return FAV_EUNSUPPORTED; }
        L.type = L_CONV; L.cin = (int)cin; L.cout = (int)cout; L.k = (int)kw; L.stride = (int)dw; L.pad = (int)pw;
        if (!tensor_to_floats(field(m, "weight"), L.w) || L.w.size() != (size_t)L.cin * L.cout * L.k * L.k) {
            set_error(".t7: convolution weight has %zu elements, expected %d", L.w.size(), L.cin * L.cout * L.k * L.k); return FAV_EFORMAT; }
        const Val* bias = field(m, "bias");
        if (bias && bias->kind == Val::TENSOR) {
            if (!tensor_to_floats(bias, L.b) || (!L.b.empty() && L.b.size() != (size_t)L.cout)) { set_error(".t7: bad convolution bias"); return FAV_EFORMAT; }
        }
    } else if (c == "nn.SpatialFullConvolution" || c == "cudnn.SpatialFullConvolution") {
        double cin, cout, kw, kh, dw = 1, dh = 1, pw = 0, ph = 0, aw = 0, ah = 0;
        if (!num_field(m, "nInputPlane", cin) || !num_field(m, "nOutputPlane", cout) || !num_field(m, "kW", kw) || !num_field(m, "kH", kh)) {
            set_error(".t7: SpatialFullConvolution without size fields"); return FAV_EFORMAT; }
        num_field(m, "dW", dw); num_field(m, "dH", dh); num_field(m, "padW", pw); num_field(m, "padH", ph); num_field(m, "adjW", aw); num_field(m, "adjH", ah);
        if (kw != kh || dw != dh || pw != ph || aw != ah) { set_error(".t7: non-square full-convolution geometry is unsupported"); return FAV_EUNSUPPORTED; }
        L.type = L_CONV; L.transposed = 1; L.cin = (int)cin; L.cout = (int)cout; L.k = (int)kw; L.stride = (int)dw; L.pad = (int)pw; L.adj = (int)aw;
        if (!tensor_to_floats(field(m, "weight"), L.w) || L.w.size() != (size_t)L.cin * L.cout * L.k * L.k) {
            set_error(".t7: full-convolution weight has %zu elements, expected %d", L.w.size(), L.cin * L.cout * L.k * L.k); return FAV_EFORMAT; }
        const Val* bias = field(m, "bias");
        if (bias && bias->kind == Val::TENSOR) {
            if (!tensor_to_floats(bias, L.b) || (!L.b.empty() && L.b.size() != (size_t)L.cout)) { set_error(".t7: bad full-convolution bias"); return FAV_EFORMAT; }
        }
    } else if (c == "nn.SpatialBatchNormalization" || c == "cudnn.SpatialBatchNormalization") {
        // evaluate mode (core.lua:47): y = (x - running_mean) / sqrt(running_var + eps) * weight + bias
        L.type = L_BN;
        if (num_field(m, "eps", a)) L.eps = (float)a;
        if (!tensor_to_floats(field(m, "running_mean"), L.mean) || L.mean.empty()) { set_error(".t7: SpatialBatchNormalization without running_mean"); return FAV_EFORMAT; }
        if (!tensor_to_floats(field(m, "running_var"), L.var) || L.var.empty()) {
            std::vector<float> rstd;                     // old checkpoints: running_std = 1/sqrt(var + eps)
            if (!tensor_to_floats(field(m, "running_std"), rstd) || rstd.size() != L.mean.size()) { set_error(".t7: SpatialBatchNormalization without running_var"); return FAV_EFORMAT; }
            L.var.resize(rstd.size());
            for (size_t i = 0; i < rstd.size(); ++i) L.var[i] = (float)(1.0 / ((double)rstd[i] * rstd[i]) - (double)L.eps);
        }
        tensor_to_floats(field(m, "weight"), L.gamma); tensor_to_floats(field(m, "bias"), L.beta);
        if (L.gamma.empty()) L.gamma.assign(L.mean.size(), 1.f);
        if (L.beta.empty()) L.beta.assign(L.mean.size(), 0.f);
        if (L.var.size() != L.mean.size() || L.gamma.size() != L.mean.size() || L.beta.size() != L.mean.size()) { set_error(".t7: bad SpatialBatchNormalization parameters"); return FAV_EFORMAT; }
    } else if (c == "nn.InstanceNormalization") {
        L.type = L_IN;
        if (!tensor_to_floats(field(m, "weight"), L.gamma) || !tensor_to_floats(field(m, "bias"), L.beta) ||
            L.gamma.empty() || L.gamma.size() != L.beta.size()) { set_error(".t7: bad InstanceNormalization parameters"); return FAV_EFORMAT; }
        if (num_field(m, "eps", a)) L.eps = (float)a;
    } else if (c == "nn.ReLU" || c == "cudnn.ReLU") {
        L.type = L_RELU;
    } else if (c == "nn.SpatialUpSamplingNearest") {
        if (!num_field(m, "scale_factor", a)) { set_error(".t7: SpatialUpSamplingNearest without scale_factor"); return FAV_EFORMAT; }
        L.type = L_UP; L.scale = (int)a;
    } else if (c == "nn.Tanh" || c == "cudnn.Tanh") {
        L.type = L_TANH;
    } else if (c == "nn.MulConstant") {
        if (!num_field(m, "constant_scalar", a)) { set_error(".t7: MulConstant without constant_scalar"); return FAV_EFORMAT; }
        L.type = L_MUL; L.mul = (float)a;
    } else if (c == "nn.TotalVariation" || c == "nn.Identity") {
        L.type = L_IDENTITY;                       // TotalVariation.lua:12-15: forward is the identity
    } else {
        set_error(".t7: module %s is outside the supported hot path (see DESIGN.md, 'next' rows)", c.c_str());
        return FAV_EUNSUPPORTED;
    }
    out.push_back(std::move(L));
    return FAV_OK;
}

int extract(const Val* seq, std::vector<Layer>& out, int depth)
{
    for (const Val* m : array_items(field(seq, "modules"))) {
        int rc = module_to_layers(m, out, depth);
        if (rc) return rc;
    }
    return FAV_OK;
}

// ---- blob (de)serialisation --------------------------------------------------------------------
struct W {
    std::vector<uint8_t>& b;
    void i32(int v) { const uint8_t* p = reinterpret_cast<const uint8_t*>(&v); b.insert(b.end(), p, p + 4); }
    void f32(float v) { const uint8_t* p = reinterpret_cast<const uint8_t*>(&v); b.insert(b.end(), p, p + 4); }
    void vec(const std::vector<float>& v) {
        i32((int)v.size());
        const uint8_t* p = reinterpret_cast<const uint8_t*>(v.data()); b.insert(b.end(), p, p + v.size() * 4);
    }
};

void pack_layers(const std::vector<Layer>& ls, W& w)
{
    w.i32((int)ls.size());
    for (const Layer& L : ls) {
        w.i32((int)L.type);
        w.i32(L.pl); w.i32(L.pr); w.i32(L.pt); w.i32(L.pb); w.i32(L.pad_mode);
        w.i32(L.cin); w.i32(L.cout); w.i32(L.k); w.i32(L.stride); w.i32(L.pad);
        w.i32(L.scale); w.i32(L.shave); w.f32(L.mul); w.f32(L.eps); w.i32(L.transposed); w.i32(L.adj);
        w.vec(L.w); w.vec(L.b); w.vec(L.gamma); w.vec(L.beta); w.vec(L.mean); w.vec(L.var);
        pack_layers(L.block, w);
    }
}

struct R {
    const uint8_t* d; size_t n; size_t p = 0; bool ok = true;
    int i32() { if (p + 4 > n) { ok = false; return 0; } int v; memcpy(&v, d + p, 4); p += 4; return v; }
    float f32() { if (p + 4 > n) { ok = false; return 0; } float v; memcpy(&v, d + p, 4); p += 4; return v; }
    void vec(std::vector<float>& v) {
        const int c = i32();
        if (!ok || c < 0 || p + (size_t)c * 4 > n) { ok = false; return; }
        v.resize((size_t)c); memcpy(v.data(), d + p, (size_t)c * 4); p += (size_t)c * 4;
    }
};

void unpack_layers(R& r, std::vector<Layer>& ls, int depth)
{
    const int cnt = r.i32();
    if (!r.ok || cnt < 0 || cnt > 4096 || depth > 8) { r.ok = false; return; }
    ls.resize((size_t)cnt);
    for (Layer& L : ls) {
        L.type = (LayerType)r.i32();
        L.pl = r.i32(); L.pr = r.i32(); L.pt = r.i32(); L.pb = r.i32(); L.pad_mode = r.i32();
        L.cin = r.i32(); L.cout = r.i32(); L.k = r.i32(); L.stride = r.i32(); L.pad = r.i32();
        L.scale = r.i32(); L.shave = r.i32(); L.mul = r.f32(); L.eps = r.f32(); L.transposed = r.i32(); L.adj = r.i32();
        r.vec(L.w); r.vec(L.b); r.vec(L.gamma); r.vec(L.beta); r.vec(L.mean); r.vec(L.var);
        unpack_layers(r, L.block, depth + 1);
        if (!r.ok) return;
    }
}

}  // namespace

int t7_parse_model(const char* path, std::vector<Layer>& out)
{
    out.clear();
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("ERROR: Could not load model from %s", path); return FAV_EIO; }   // core.lua:41
    std::vector<uint8_t> data;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + got);
    fclose(f);
    Reader r{data.data(), data.size()};
    VP root = r.obj();
    if (!r.ok) { set_error(".t7 parse error in %s: %s", path, r.err.c_str()); return FAV_EFORMAT; }
    const Val* model = nullptr;
    if (root->kind == Val::TABLE) model = field(root.get(), "model");     // checkpoint.model (core.lua:46)
    else if (root->kind == Val::OBJECT) model = root.get();               // a bare nn.Sequential
    if (!model || model->kind != Val::OBJECT) { set_error(".t7: %s has no `model` field", path); return FAV_EFORMAT; }
    if (model->str != "nn.Sequential") { set_error(".t7: model is %s, expected nn.Sequential", model->str.c_str()); return FAV_EUNSUPPORTED; }
    return extract(model, out, 0);
}

int blob_pack(const std::vector<Layer>& layers, std::vector<uint8_t>& blob)
{
    blob.clear();
    W w{blob};
    w.i32(0x42564146);   // "FAVB"
    w.i32(3);            // 3: padding layers carry their mode (round 6)
    pack_layers(layers, w);
    return FAV_OK;
}

int blob_unpack(const void* blob, size_t bytes, std::vector<Layer>& out)
{
    R r{static_cast<const uint8_t*>(blob), bytes};
    if (r.i32() != 0x42564146 || r.i32() != 3) { set_error("weight blob: bad magic/version"); return FAV_EFORMAT; }
    unpack_layers(r, out, 0);
    if (!r.ok) { set_error("weight blob: truncated or corrupt"); return FAV_EFORMAT; }
    return FAV_OK;
}

std::string describe_layers(const std::vector<Layer>& layers, int indent)
{
    std::string s;
    char buf[256];
    const std::string pad((size_t)indent * 2, ' ');
    for (const Layer& L : layers) {
        switch (L.type) {
        case L_PAD: snprintf(buf, sizeof buf, "%s %d %d %d %d", L.pad_mode ? "replicate-pad" : "pad", L.pl, L.pr, L.pt, L.pb); break;
        case L_CONV:
            if (L.transposed) snprintf(buf, sizeof buf, "fullconv %d %d %d %d %d adj=%d bias=%d", L.cin, L.cout, L.k, L.stride, L.pad, L.adj, L.b.empty() ? 0 : 1);
            else snprintf(buf, sizeof buf, "conv %d %d %d %d %d bias=%d", L.cin, L.cout, L.k, L.stride, L.pad, L.b.empty() ? 0 : 1);
            break;
        case L_BN: snprintf(buf, sizeof buf, "bn %zu", L.mean.size()); break;
        case L_IN: snprintf(buf, sizeof buf, "in %zu", L.gamma.size()); break;
        case L_RELU: snprintf(buf, sizeof buf, "relu"); break;
        case L_RES: snprintf(buf, sizeof buf, "res shave=%d", L.shave); break;
        case L_UP: snprintf(buf, sizeof buf, "up %d", L.scale); break;
        case L_TANH: snprintf(buf, sizeof buf, "tanh"); break;
        case L_MUL: snprintf(buf, sizeof buf, "mul %g", (double)L.mul); break;
        default: snprintf(buf, sizeof buf, "identity"); break;
        }
        s += pad + buf + "\n";
        if (L.type == L_RES) s += describe_layers(L.block, indent + 1);
    }
    return s;
}

}  // namespace fav
