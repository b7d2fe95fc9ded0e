// config.cpp -- network configuration: defaults of CORE/configs/base.json and a small JSON reader
// for the tiny-cuda-nn schema (NeRF_Model::ReadNetworkConfig, CORE/src/nerf_model.cu:1272-1284 parses
// with comments allowed; ResetNetwork :1286-1342 reads encoding/network/optimizer keys).
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include "model.h"
#include "xorwow.h"

namespace mon {

void set_error(const char* fmt, ...);

// ------------------------------------------------------------------ minimal JSON (objects, arrays, strings, numbers, bools, null, // and /* */ comments)
struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0; bool b = false; std::string str;
    std::vector<JVal> arr; std::map<std::string, JVal> obj;
    const JVal* get(const char* k) const { if (kind != Obj) return nullptr; auto it = obj.find(k); return it == obj.end() ? nullptr : &it->second; }
    double number(const char* k, double dflt) const { const JVal* v = get(k); return (v && v->kind == Num) ? v->num : dflt; }
    std::string string(const char* k, const char* dflt) const { const JVal* v = get(k); return (v && v->kind == Str) ? v->str : std::string(dflt); }
};

struct JParser {
    // nesting is bounded (tcnn configs are 3 deep): no stack exhaustion on hostile input
    const std::string& s; size_t i = 0; bool ok = true; int depth = 0;
    explicit JParser(const std::string& src) : s(src) {}
    void ws() {
        for (;;) {
            while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
            if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '/') { while (i < s.size() && s[i] != '\n') ++i; continue; }
            if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '*') { i += 2; while (i + 1 < s.size() && !(s[i] == '*' && s[i + 1] == '/')) ++i; i += 2;
                continue; }
            break;
        }
    }
    JVal value() {
        ws(); JVal v;
        if (i >= s.size() || depth > 32) { ok = false; return v; }
        struct Depth { int& d; explicit Depth(int& x) : d(x) { ++d; } ~Depth() { --d; } } guard(depth);
        const char c = s[i];
        if (c == '{') {
            v.kind = JVal::Obj; ++i; ws();
            if (i < s.size() && s[i] == '}') { ++i; return v; }
            while (ok) {
                ws(); JVal k = value(); if (k.kind != JVal::Str) { ok = false; break; }
                ws(); if (i >= s.size() || s[i] != ':') { ok = false; break; } ++i;
                v.obj[k.str] = value(); ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == '}') { ++i; break; }
                ok = false;
            }
        } else if (c == '[') {
            v.kind = JVal::Arr; ++i; ws();
            if (i < s.size() && s[i] == ']') { ++i; return v; }
            while (ok) {
                v.arr.push_back(value()); ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == ']') { ++i; break; }
                ok = false;
            }
        } else if (c == '"') {
            v.kind = JVal::Str; ++i;
            while (i < s.size() && s[i] != '"') { if (s[i] == '\\' && i + 1 < s.size()) ++i; v.str.push_back(s[i++]); }
            if (i >= s.size()) ok = false; else ++i;
        } else if (!std::strncmp(s.c_str() + i, "true", 4)) { v.kind = JVal::Bool; v.b = true; i += 4; }
        else if (!std::strncmp(s.c_str() + i, "false", 5)) { v.kind = JVal::Bool; v.b = false; i += 5; }
        else if (!std::strncmp(s.c_str() + i, "null", 4)) { i += 4; }
        else {
            char* end = nullptr; v.num = std::strtod(s.c_str() + i, &end);
            if (end == s.c_str() + i) ok = false; else { v.kind = JVal::Num; i = (size_t)(end - s.c_str()); }
        }
        return v;
    }
};

// ------------------------------------------------------------------ defaults: CORE/configs/base.json + nerf_model.h constants
void config_default(mon_config& c) {
    std::memset(&c, 0, sizeof(c));
    c.n_levels = 16; c.n_features = 2; c.log2_hashmap_size = 16; c.base_resolution = 16;      // base.json:23-29
    c.per_level_scale = 2.0f;                                                                   // tcnn default; not in base.json (SURVEY App.A-1)
    c.n_neurons = 64; c.n_hidden_layers = 1;                                                    // base.json:30-36
    c.rays_per_batch = 4096; c.n_samples = 32; c.loss_scale = 128.0f;                           // nerf_model.h:166,172-175
    c.learning_rate = 1e-2f; c.beta1 = 0.9f; c.beta2 = 0.99f; c.epsilon = 1e-15f; c.l2_reg = 1e-6f;   // base.json:14-21
    c.ema_decay = 0.95f; c.decay_start = 20000; c.decay_interval = 10000; c.decay_base = 0.33f;  // base.json:5-13
    c.param_seed = 1337u; c.sample_seed = 2024ull; c.use_depth = 0;
}

int config_from_json(const char* path, mon_config& c) {
    std::ifstream f(path);
    if (!f) { set_error("config file error: cannot open %s", path); return MON_ERR_IO; }      // nerf_model.cu:1276-1280
    std::stringstream ss; ss << f.rdbuf(); const std::string src = ss.str();
    JParser p(src); JVal root = p.value();
    if (!p.ok || root.kind != JVal::Obj) { set_error("config file error: JSON parse failed in %s", path); return MON_ERR_IO; }
    config_default(c);
    if (const JVal* e = root.get("encoding")) {                                                 // nerf_model.cu:1299-1302
        c.n_features = (int)e->number("n_features_per_level", 2);
        c.n_levels = (int)e->number("n_levels", 16);
        c.base_resolution = (int)e->number("base_resolution", 16);
        c.log2_hashmap_size = (int)e->number("log2_hashmap_size", 15);
        c.per_level_scale = (float)e->number("per_level_scale", 2.0);
        const std::string ot = e->string("otype", "HashGrid");
        if (ot != "HashGrid" && ot != "Grid") { set_error("unsupported encoding otype %s", ot.c_str()); return MON_ERR_ARG; }
    }
    if (const JVal* n = root.get("network")) {
        c.n_neurons = (int)n->number("n_neurons", 64);
        c.n_hidden_layers = (int)n->number("n_hidden_layers", 1);
        const std::string act = n->string("activation", "ReLU"), oact = n->string("output_activation", "None");
        if (act != "ReLU" || oact != "None") { set_error("unsupported activation %s/%s", act.c_str(), oact.c_str()); return MON_ERR_ARG; }
    }
    // optimizer: walk the nesting Ema -> ExponentialDecay -> Adam (base.json:5-22); any level may be absent
    const JVal* o = root.get("optimizer");
    bool have_ema = false, have_decay = false;
    while (o && o->kind == JVal::Obj) {
        const std::string ot = o->string("otype", "");
        if (ot == "Ema") { c.ema_decay = (float)o->number("decay", 0.99); have_ema = true; }
        else if (ot == "ExponentialDecay") {
            c.decay_start = (int)o->number("decay_start", 10000); c.decay_interval = (int)o->number("decay_interval", 10000);
            c.decay_base = (float)o->number("decay_base", 0.33); have_decay = true;
        } else if (ot == "Adam") {
            c.learning_rate = (float)o->number("learning_rate", 1e-3); c.beta1 = (float)o->number("beta1", 0.9);
            c.beta2 = (float)o->number("beta2", 0.999); c.epsilon = (float)o->number("epsilon", 1e-8); c.l2_reg = (float)o->number("l2_reg", 1e-8);
        } else { set_error("unsupported optimizer otype %s", ot.c_str()); return MON_ERR_ARG; }
        o = o->get("nested");
    }
    if (!have_ema) c.ema_decay = 0.0f;             // EMA with decay 0 == plain weights
    if (!have_decay) { c.decay_start = 0x7fffffff; c.decay_interval = 0; c.decay_base = 1.0f; }
    // optional extension block for the constants the reference hard-codes (nerf_model.h:166,172-175)
    if (const JVal* t = root.get("training")) {
        c.rays_per_batch = (int)t->number("rays_per_batch", c.rays_per_batch);
        c.n_samples = (int)t->number("n_samples", c.n_samples);
        c.loss_scale = (float)t->number("loss_scale", c.loss_scale);
        c.sample_seed = (uint64_t)t->number("sample_seed", (double)c.sample_seed);
        c.param_seed = (uint32_t)t->number("param_seed", (double)c.param_seed);
        // "same inputs" mode (include/mon_core.h mon_config::rng_flags), e.g. 17 = XORWOW in cuRAND's flavour + tcnn's init order
        c.rng_flags = (uint32_t)t->number("rng_flags", (double)c.rng_flags);
    }
    return MON_OK;
}

// ------------------------------------------------------------------ level table (tcnn grid.h; SURVEY TCNN-A1/A2/A4)
static uint32_t next_multiple(uint32_t v, uint32_t d) { return ((v + d - 1) / d) * d; }

int level_table_build(const mon_config& c, LevelTable& lt, NetDims& nd, uint32_t& n_grid) {
    if (c.n_levels < 1 || c.n_levels > kMaxLevels || c.n_features != 2) { set_error("n_levels must be 1..%d and n_features 2", kMaxLevels);
        return MON_ERR_ARG; }
    // tcnn FullyFusedMLP's widths (base.json:30-36 is user-editable): 32 and 64 run on the fused MFMA kernels with one or two hidden layers, 128 with one;
    // 16 neurons (half an MFMA tile) and 128 x 2 go through the layer-at-a-time kernels (fused_supported, kernels_fused.hip)
    // Three and four hidden layers (tcnn takes any count) exist on the layer-at-a-time kernels for the widths up to 64.
    const bool width_ok = c.n_neurons == 16 || c.n_neurons == 32 || c.n_neurons == 64 || c.n_neurons == 128;
    const bool depth_ok = c.n_hidden_layers == 1 || c.n_hidden_layers == 2 || ((c.n_hidden_layers == 3 || c.n_hidden_layers == 4) && c.n_neurons <= 64);
    if (!width_ok || !depth_ok) { set_error("n_neurons must be 16|32|64|128, n_hidden_layers 1|2 (3|4 up to 64 neurons)"); return MON_ERR_ARG; }
    if (c.log2_hashmap_size < 4 || c.log2_hashmap_size > 26) { set_error("log2_hashmap_size out of range"); return MON_ERR_ARG; }
    uint32_t off = 0; const float l2 = std::log2(c.per_level_scale);
    for (int l = 0; l < c.n_levels; ++l) {
        const float s = std::exp2((float)l * l2) * (float)c.base_resolution - 1.0f;
        const uint32_t r = (uint32_t)std::ceil(s) + 1u;
        const uint64_t dense = (uint64_t)r * r * r; const uint32_t maxp = 0xffffffffu / 2;
        uint32_t n = dense > maxp ? maxp : (uint32_t)dense;
        n = next_multiple(n, 8u);
        const uint32_t cap = 1u << c.log2_hashmap_size; if (n > cap) n = cap;
        lt.offset[l] = off; lt.scale[l] = s; lt.res[l] = r; off += n;
    }
    for (int l = c.n_levels; l <= kMaxLevels; ++l) lt.offset[l] = off;
    nd.L = c.n_levels; nd.Epad = (int)next_multiple((uint32_t)(c.n_levels * 2), 16u); nd.W = c.n_neurons; nd.NH = c.n_hidden_layers;
    nd.n_mlp = (uint32_t)(nd.W * nd.Epad + (nd.NH - 1) * nd.W * nd.W + kOutPad * nd.W);
    n_grid = off * 2u;
    return MON_OK;
}

// Closed-form per-level index constants (device_common.h:LevelFast) = tcnn's stride loop replayed in uint32.
void level_fast_build(const LevelTable& lt, const NetDims& nd, LevelFast& lf) {
    for (int l = 0; l < kMaxLevels; ++l) { lf.scale[l] = 0.f; lf.size[l] = 1; lf.my[l] = lf.mz[l] = 0; lf.mask[l] = 0; lf.hashed[l] = 0;
        lf.offset[l] = lt.offset[l]; }
    lf.offset[kMaxLevels] = lt.offset[kMaxLevels];
    for (int l = 0; l < nd.L; ++l) {
        const uint32_t size = lt.offset[l + 1] - lt.offset[l], res = lt.res[l];
        uint32_t stride = 1, mult[3] = { 0, 0, 0 };
        for (int d = 0; d < 3 && stride <= size; ++d) { mult[d] = stride; stride *= res; }      // uint32 wrap-around on purpose
        const bool hashed = size < stride;
        lf.scale[l] = lt.scale[l]; lf.size[l] = size; lf.hashed[l] = hashed ? 1u : 0u;
        lf.my[l] = hashed ? 2654435761u : mult[1]; lf.mz[l] = hashed ? 805459861u : mult[2];
        lf.mask[l] = ((size & (size - 1u)) == 0u) ? size - 1u : 0xffffffffu;
    }
}

// ------------------------------------------------------------------ parameter init (SURVEY TCNN-A5)
// pcg32 (tcnn::default_rng_t, seed 1337): MLP Xavier-uniform per matrix, then grid U(-1e-4, 1e-4).
namespace {
struct Pcg32 {
    uint64_t state = 0, inc = 0;
    uint32_t next() { const uint64_t old = state; state = old * 0x5851f42d4c957f2dull + inc;
        const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u); return (xs >> rot) | (xs << ((~rot + 1u) & 31)); }
    void seed(uint64_t initstate, uint64_t initseq) { state = 0; inc = (initseq << 1u) | 1u; next(); state += initstate; next(); }
    float next_float() { const uint32_t u = (next() >> 9) | 0x3f800000u; float f; std::memcpy(&f, &u, 4); return f - 1.0f; }
};
}  // namespace

void init_params_host(const mon_config& c, const NetDims& nd, uint32_t n_params, std::vector<float>& master) {
    master.resize(n_params);
    Pcg32 rng; rng.seed(c.param_seed, 1u);
    uint32_t k = 0;
    if (rng_tcnn_init_order(c.rng_flags)) {
        // TCNN-A5b: tiny-cuda-nn's generate_random_uniform as published -- one launch per tensor (every MLP matrix, then the grid) of ceil(n / 512) blocks of
        // 128 threads; thread i advances the generator by 4 i and writes draw j = 0..3 to element i + n_threads j; the host generator then advances by n. 
        // Element e of a tensor
        // = draw 4 (e mod n_threads) + floor(e / n_threads) of the tensor's stretch of the pcg32 sequence, scaled as draw * (hi - lo) + lo.
        uint64_t base = 0;
        for (int layer = 0; layer <= nd.NH + 1; ++layer) {
            size_t n; float lo, hi;
            if (layer <= nd.NH) { const int rows = (layer == nd.NH) ? kOutPad : nd.W, cols = (layer == 0) ? nd.Epad : nd.W;
                const float sc = std::sqrt(6.0f / (float)(rows + cols)); n = (size_t)rows * cols; lo = -sc; hi = sc; }
            else { n = n_params - nd.n_mlp; lo = -1e-4f; hi = 1e-4f; }
            const size_t n_threads = ((n + 511) / 512) * 128;
            std::vector<float> draws(n_threads * 4);
            Pcg32 r2; r2.seed(c.param_seed, 1u); for (uint64_t a = 0; a < base; ++a) (void)r2.next();
            for (auto& d : draws) d = r2.next_float();
            for (size_t e = 0; e < n; ++e) master[k + e] = draws[4 * (e % n_threads) + e / n_threads] * (hi - lo) + lo;
            k += (uint32_t)n; base += n;
        }
        return;
    }
    for (int layer = 0; layer <= nd.NH; ++layer) {
        const int rows = (layer == nd.NH) ? kOutPad : nd.W, cols = (layer == 0) ? nd.Epad : nd.W;
        const float sc = std::sqrt(6.0f / (float)(rows + cols));
        for (int i = 0; i < rows * cols; ++i, ++k) master[k] = rng.next_float() * (2.0f * sc) - sc;
    }
    for (; k < n_params; ++k) master[k] = rng.next_float() * 2e-4f - 1e-4f;
}

// ---- XORWOW lane states (xorwow.h): lane k = the seed state advanced 2^67 k steps.  The xorshift part of the state is linear over GF(2): its one-step matrix
//      (160 columns of 5 words) is squared 67 times, once per process.
namespace {
struct XwMat { uint32_t col[160][5]; };
void xw_matvec(const XwMat& M, const uint32_t v[5], uint32_t out[5]) {
    uint32_t r[5] = { 0, 0, 0, 0, 0 };
    for (int b = 0; b < 160; ++b) if ((v[b >> 5] >> (b & 31)) & 1u) for (int k = 0; k < 5; ++k) r[k] ^= M.col[b][k];
    for (int k = 0; k < 5; ++k) out[k] = r[k];
}
const XwMat& xw_jump_2pow67() {
    static const XwMat J = [] {
        XwMat M{}, T{};
        for (int b = 0; b < 160; ++b) { XorwowState e{}; e.x[b >> 5] = 1u << (b & 31); (void)xorwow_next(e); for (int k = 0; k < 5; ++k) M.col[b][k] = e.x[k]; }
        for (int q = 0; q < 67; ++q) { for (int b = 0; b < 160; ++b) xw_matvec(M, M.col[b], T.col[b]); M = T; }
        return M;
    }();
    return J;
}
}  // namespace
void xorwow_lane_states(uint64_t seed, int flavour, uint32_t lanes, std::vector<XorwowState>& out) {
    out.resize(lanes); xorwow_seed(out[0], seed, flavour);
    const XwMat& J = xw_jump_2pow67();
    for (uint32_t k = 1; k < lanes; ++k) { out[k].d = out[0].d; xw_matvec(J, out[k - 1].x, out[k].x); }
}

}  // namespace mon
