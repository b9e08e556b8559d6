// libllie_hip.so: context, state_dict repack, workspace arena and the launch sequence of the
// denoiser behind the C ABI of include/llie.h.  Host-only code; kernels live in the *.hip files.
//
// Execution model: one UNet forward is a fixed sequence of kernel launches on the caller's stream.
// All temporaries come from a caller-provided workspace through a deterministic first-fit arena, so
// the same (batch, H, W) always produces the same offsets: llie_workspace_bytes() replays the
// sequence with launches disabled to obtain the high-water mark.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/llie.h"
#include "kernels.h"

using namespace llie;

namespace {

thread_local char g_err[512] = "";
void set_err(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

size_t elem_size(int dt) { return dt == LLIE_F32 ? 4 : 2; }
size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// Parameter table
enum PKind { PK_F32, PK_MAT, PK_CONV3, PK_DW, PK_INIT, PK_FINAL };
struct Param {
  std::string key;
  int64_t numel = 0;
  PKind kind = PK_F32;
  size_t off = 0;                      // byte offset of the destination in the weight blob
  int rows = 0, cols = 0, ld = 0, col0 = 0;  // PK_MAT: [rows][cols] -> dst[r*ld + col0 + c]
  int O = 0, I = 0;
  bool as_t = true;                    // PK_MAT: store as compute dtype (true) or fp32
  bool loaded = false;
  int ndim = 1;
  int64_t shape[4] = {0, 0, 0, 0};     // shape in the reference's state_dict
};

struct IrbW {
  int cin, cout, hid, sq;
  bool skip;
  size_t n1g, n1b, n2g, n2b, w_expand, w_dw, se_w1, se_b1, se_w2, se_b2, w_proj;
  int film_off;  // first row of this block inside the concatenated FiLM projection
};
struct AttnW {
  int c, heads, inner;
  size_t ng, nb, w_qkv, w_out, n2g, n2b;
};
struct ConvW {
  int c;
  size_t w, bias;
};
struct Block {
  int kind;  // 0 irb, 1 attn
  int idx;
};

// ---------------------------------------------------------------------------------------------
// Deterministic first-fit arena over the caller's workspace.
struct Arena {
  struct Blk { size_t off, size; };
  std::vector<Blk> freelist;   // sorted by offset, coalesced
  std::map<size_t, size_t> live;  // off -> size
  size_t cap = 0, high = 0;
  bool failed = false;
  explicit Arena(size_t capacity) : cap(capacity) { freelist.push_back({0, capacity}); }
  size_t alloc(size_t bytes) {
    bytes = align_up(bytes ? bytes : 1, 256);
    for (size_t i = 0; i < freelist.size(); ++i) {
      if (freelist[i].size >= bytes) {
        const size_t off = freelist[i].off;
        freelist[i].off += bytes;
        freelist[i].size -= bytes;
        if (!freelist[i].size) freelist.erase(freelist.begin() + i);
        live[off] = bytes;
        if (off + bytes > high) high = off + bytes;
        return off;
      }
    }
    failed = true;
    return 0;
  }
  void free(size_t off) {
    auto it = live.find(off);
    if (it == live.end()) return;
    Blk b{off, it->second};
    live.erase(it);
    size_t i = 0;
    while (i < freelist.size() && freelist[i].off < b.off) ++i;
    freelist.insert(freelist.begin() + i, b);
    if (i + 1 < freelist.size() && freelist[i].off + freelist[i].size == freelist[i + 1].off) {
      freelist[i].size += freelist[i + 1].size;
      freelist.erase(freelist.begin() + i + 1);
    }
    if (i > 0 && freelist[i - 1].off + freelist[i - 1].size == freelist[i].off) {
      freelist[i - 1].size += freelist[i].size;
      freelist.erase(freelist.begin() + i);
    }
  }
};

// NHWC activation living in the workspace, with the stats slab its producer wrote.
struct Tens {
  size_t off = 0, slab = 0;
  int C = 0, H = 0, W = 0, ntiles = 0;
  bool valid = false;
};

}  // namespace

struct llie_ctx {
  llie_config cfg{};
  int dt = 0;
  std::vector<Param> params;
  std::map<std::string, int> index;
  size_t blob_bytes = 0;
  char* blob = nullptr;
  // topology
  std::vector<IrbW> irbs;
  std::vector<AttnW> attns;
  std::vector<ConvW> downs, ups;
  std::vector<std::vector<Block>> enc, dec;
  std::vector<Block> mid;
  std::vector<int> channels;
  // UNet-level tensors
  size_t t_w1 = 0, t_b1 = 0, t_w3 = 0, t_b3 = 0, freqs = 0, film_w = 0, film_b = 0;
  int film_rows = 0;
  size_t init_wp = 0, fin_wp = 0;  // MFMA-packed init / final conv weights (2-byte compute dtypes)
  size_t init_w = 0, init_b = 0, fin_g = 0, fin_b = 0, fin_w = 0, fin_bias = 0;
  // hipGraph cache of llie_enhance launch sequences (key -> executable graph)
  struct GraphEntry { bool seen = false; hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; };
  std::map<std::string, GraphEntry> graphs;
  hipStream_t cap_stream = nullptr;  // side stream used only to record captures (the legacy null stream cannot capture)
  // per-kernel-class HIP-event profiling (llie_profile_begin / llie_profile_end)
  int prof_mask = 0;
  struct ProfRec { int cls; int64_t bytes; hipEvent_t e0, e1; const char* name; };
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> event_pool;
  hipEvent_t get_event() {
    if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
};

namespace {

// ---------------------------------------------------------------------------------------------
// Topology builder (efficient_unet.py:403-530).  Adds parameters in the reference's registration
// order and lays the weight blob out.
struct Builder {
  llie_ctx* c;
  size_t cursor = 0;
  size_t reserve(size_t bytes) {
    const size_t o = cursor;
    cursor += align_up(bytes, 256);
    return o;
  }
  Param& add(const std::string& key, int64_t numel, PKind kind, size_t off) {
    Param p;
    p.key = key;
    p.numel = numel;
    p.kind = kind;
    p.off = off;
    p.ndim = 1;
    p.shape[0] = numel;
    c->index[key] = (int)c->params.size();
    c->params.push_back(p);
    return c->params.back();
  }
  size_t f32(const std::string& key, int64_t n) {
    const size_t o = reserve((size_t)n * 4);
    add(key, n, PK_F32, o);
    return o;
  }
  size_t es() const { return elem_size(c->dt); }
  // matrix [rows][cols] stored in compute dtype at an existing destination
  void mat_into(const std::string& key, int rows, int cols, size_t off, int ld, int col0, bool as_t = true) {
    Param& p = add(key, (int64_t)rows * cols, PK_MAT, off);
    p.rows = rows; p.cols = cols; p.ld = ld; p.col0 = col0; p.as_t = as_t;
    p.ndim = 4; p.shape[0] = rows; p.shape[1] = cols; p.shape[2] = 1; p.shape[3] = 1;  // 1x1 conv weight
  }
  static void set_shape(Param& p, std::initializer_list<int64_t> dims) {
    p.ndim = (int)dims.size();
    int i = 0;
    for (int64_t d : dims) p.shape[i++] = d;
  }
  size_t mat(const std::string& key, int rows, int cols) {
    const size_t o = reserve((size_t)rows * cols * es());
    mat_into(key, rows, cols, o, cols, 0);
    return o;
  }
  int add_irb(const std::string& p, int cin, int cout, int T, int e) {
    IrbW w{};
    w.cin = cin; w.cout = cout; w.hid = cin * e; w.sq = std::max(1, (int)(w.hid * 0.25));
    w.skip = cin != cout;
    w.n1g = f32(p + ".norm1.weight", cin); w.n1b = f32(p + ".norm1.bias", cin);
    w.n2g = f32(p + ".norm2.weight", w.hid); w.n2b = f32(p + ".norm2.bias", w.hid);
    w.w_expand = mat(p + ".expand.weight", w.hid, cin);
    w.w_dw = reserve((size_t)9 * w.hid * 4);
    { Param& q = add(p + ".depthwise.weight", (int64_t)w.hid * 9, PK_DW, w.w_dw); q.O = w.hid; set_shape(q, {w.hid, 1, 3, 3}); }
    w.se_w1 = mat(p + ".se.fc1.weight", w.sq, w.hid); w.se_b1 = f32(p + ".se.fc1.bias", w.sq);
    w.se_w2 = mat(p + ".se.fc2.weight", w.hid, w.sq); w.se_b2 = f32(p + ".se.fc2.bias", w.hid);
    const int kp = w.hid + (w.skip ? cin : 0);  // project and skip share one K-concatenated matrix
    w.w_proj = reserve((size_t)cout * kp * es());
    mat_into(p + ".project.weight", cout, w.hid, w.w_proj, kp, 0);
    // FiLM Linear: rows appended to the global [F][T] fp32 table (filled in finish())
    w.film_off = c->film_rows;
    c->film_rows += 2 * w.hid;
    film_keys.push_back({p + ".time_mlp.1", 2 * w.hid, w.film_off});
    if (w.skip) pending_skip.push_back({p + ".skip.weight", cout, cin, w.w_proj, kp, w.hid});
    flush_pending();  // registration order: ... project, time_mlp, skip
    c->irbs.push_back(w);
    return (int)c->irbs.size() - 1;
  }
  struct FilmKey { std::string p; int rows, off; };
  struct SkipKey { std::string key; int rows, cols; size_t off; int ld, col0; };
  std::vector<FilmKey> film_keys;
  std::vector<SkipKey> pending_skip;
  void flush_pending() {
    // time_mlp.1.{weight,bias} params are created now (to keep registration order) with offsets
    // patched in finish() once the total FiLM row count is known.
    const FilmKey& fk = film_keys.back();
    Param& pw = add(fk.p + ".weight", (int64_t)fk.rows * c->cfg.time_embed_dim, PK_MAT, 0);
    pw.rows = fk.rows; pw.cols = c->cfg.time_embed_dim; pw.ld = pw.cols; pw.col0 = 0; pw.as_t = false;
    set_shape(pw, {fk.rows, c->cfg.time_embed_dim});  // nn.Linear weight
    add(fk.p + ".bias", fk.rows, PK_F32, 0);
    for (auto& s : pending_skip) mat_into(s.key, s.rows, s.cols, s.off, s.ld, s.col0);
    pending_skip.clear();
  }
  int add_attn(const std::string& p, int ch, int heads) {
    AttnW w{};
    w.c = ch; w.heads = heads; w.inner = heads * 32;
    w.ng = f32(p + ".norm.weight", ch); w.nb = f32(p + ".norm.bias", ch);
    w.w_qkv = mat(p + ".to_qkv.weight", 3 * w.inner, ch);
    w.w_out = mat(p + ".to_out.0.weight", ch, w.inner);
    w.n2g = f32(p + ".to_out.1.weight", ch); w.n2b = f32(p + ".to_out.1.bias", ch);
    c->attns.push_back(w);
    return (int)c->attns.size() - 1;
  }
  ConvW add_conv3(const std::string& p, int ch) {
    ConvW w{};
    w.c = ch;
    w.w = reserve((size_t)9 * ch * ch * es());
    { Param& q = add(p + ".weight", (int64_t)ch * ch * 9, PK_CONV3, w.w); q.O = ch; q.I = ch; set_shape(q, {ch, ch, 3, 3}); }
    w.bias = f32(p + ".bias", ch);
    return w;
  }
  void finish_film() {
    const int T = c->cfg.time_embed_dim;
    c->film_w = reserve((size_t)c->film_rows * T * 4);
    c->film_b = reserve((size_t)c->film_rows * 4);
    for (auto& fk : film_keys) {
      Param& pw = c->params[c->index[fk.p + ".weight"]];
      pw.off = c->film_w + (size_t)fk.off * T * 4;
      Param& pb = c->params[c->index[fk.p + ".bias"]];
      pb.off = c->film_b + (size_t)fk.off * 4;
    }
  }
};

int build_unet(llie_ctx* c) {
  const llie_config& g = c->cfg;
  Builder b{c};
  c->channels.clear();
  for (int i = 0; i < 4; ++i) c->channels.push_back(g.base_channels * g.channel_multipliers[i]);
  const std::vector<int>& ch = c->channels;
  const int T = g.time_embed_dim, e = g.expansion_ratio;
  // GroupNorm(min(32,C), C) must be constructible for every site (efficient_unet.py:170-171,263,528)
  auto gn_ok = [](int x) { return x >= 32 && x % 32 == 0; };
  {
    int in_ch = ch[0];
    for (int l = 0; l < 4; ++l) {
      for (int k = 0; k < g.num_res_blocks; ++k) {
        const int cin = k == 0 ? in_ch : ch[l];
        if (!gn_ok(cin) || !gn_ok(cin * e)) return LLIE_ERR_CONFIG;
      }
      in_ch = ch[l];
    }
    for (int l = 0; l < 4; ++l) {
      const int out = ch[3 - l];
      if (!gn_ok(in_ch + out) || !gn_ok((in_ch + out) * e) || !gn_ok(out) || !gn_ok(out * e)) return LLIE_ERR_CONFIG;
      in_ch = out;
    }
  }
  if (g.base_channels % 2 || g.image_size % 64 || g.image_size < 64) return LLIE_ERR_SHAPE;
  if (g.in_channels < 2 || g.in_channels > 8 || g.out_channels > 4) return LLIE_ERR_SHAPE;

  c->t_w1 = b.reserve((size_t)T * g.base_channels * 4);
  { Param& p = b.add("time_mlp.1.weight", (int64_t)T * g.base_channels, PK_F32, c->t_w1); Builder::set_shape(p, {T, g.base_channels}); }
  c->t_b1 = b.f32("time_mlp.1.bias", T);
  c->t_w3 = b.f32("time_mlp.3.weight", (int64_t)T * T);
  Builder::set_shape(c->params.back(), {T, T});
  c->t_b3 = b.f32("time_mlp.3.bias", T);
  c->init_w = b.reserve((size_t)ch[0] * g.in_channels * 9 * 4);
  { Param& p = b.add("init_conv.weight", (int64_t)ch[0] * g.in_channels * 9, PK_INIT, c->init_w); p.O = ch[0]; p.I = g.in_channels; Builder::set_shape(p, {ch[0], g.in_channels, 3, 3}); }
  c->init_b = b.f32("init_conv.bias", ch[0]);
  c->init_wp = b.reserve((size_t)10 * ch[0] * 8 * 2);

  int res = g.image_size;
  auto is_attn_res = [&](int r) { return r == g.attention_resolutions[0] || r == g.attention_resolutions[1]; };
  int in_ch = ch[0];
  c->enc.assign(4, {});
  for (int l = 0; l < 4; ++l) {
    int k = 0;
    for (int r = 0; r < g.num_res_blocks; ++r) {
      const std::string p = "encoder_blocks." + std::to_string(l) + "." + std::to_string(k++);
      c->enc[l].push_back({0, b.add_irb(p, r == 0 ? in_ch : ch[l], ch[l], T, e)});
      if (is_attn_res(res)) {
        const std::string pa = "encoder_blocks." + std::to_string(l) + "." + std::to_string(k++);
        c->enc[l].push_back({1, b.add_attn(pa, ch[l], g.num_attention_heads)});
      }
    }
    in_ch = ch[l];
    if (l < 3) res /= 2;
  }
  for (int l = 0; l < 3; ++l) c->downs.push_back(b.add_conv3("downsamplers." + std::to_string(l) + ".down", ch[l]));
  c->mid.push_back({0, b.add_irb("mid_block1", ch[3], ch[3], T, e)});
  c->mid.push_back({1, b.add_attn("mid_attn", ch[3], g.num_attention_heads)});
  c->mid.push_back({0, b.add_irb("mid_block2", ch[3], ch[3], T, e)});
  c->dec.assign(4, {});
  for (int l = 0; l < 4; ++l) {
    const int out = ch[3 - l];
    int k = 0;
    for (int r = 0; r < g.num_res_blocks + 1; ++r) {
      const std::string p = "decoder_blocks." + std::to_string(l) + "." + std::to_string(k++);
      c->dec[l].push_back({0, b.add_irb(p, r == 0 ? in_ch + out : out, out, T, e)});
      if (is_attn_res(res)) {
        const std::string pa = "decoder_blocks." + std::to_string(l) + "." + std::to_string(k++);
        c->dec[l].push_back({1, b.add_attn(pa, out, g.num_attention_heads)});
      }
    }
    in_ch = out;
    if (l < 3) res *= 2;
  }
  for (int l = 0; l < 3; ++l) c->ups.push_back(b.add_conv3("upsamplers." + std::to_string(l) + ".conv", ch[3 - l]));
  c->fin_g = b.f32("final_norm.weight", ch[0]);
  c->fin_b = b.f32("final_norm.bias", ch[0]);
  c->fin_w = b.reserve((size_t)9 * ch[0] * 4 * 4);
  { Param& p = b.add("final_conv.weight", (int64_t)g.out_channels * ch[0] * 9, PK_FINAL, c->fin_w); p.O = g.out_channels; p.I = ch[0]; Builder::set_shape(p, {g.out_channels, ch[0], 3, 3}); }
  c->fin_bias = b.f32("final_conv.bias", g.out_channels);
  c->fin_wp = b.reserve((size_t)(ch[0] / 32) * 18 * 2 * 4 * 8 * 2);
  c->freqs = b.reserve((size_t)(g.base_channels / 2) * 4);
  b.finish_film();
  c->blob_bytes = b.cursor;
  return LLIE_OK;
}

int build_module(llie_ctx* c) {
  const llie_config& g = c->cfg;
  Builder b{c};
  auto gn_ok = [](int x) { return x >= 32 && x % 32 == 0; };
  switch (g.kind) {
    case LLIE_IRB:
      if (!gn_ok(g.in_channels) || !gn_ok(g.in_channels * g.expansion_ratio) || g.out_channels % 32) return LLIE_ERR_CONFIG;
      b.add_irb("", g.in_channels, g.out_channels, g.time_embed_dim, g.expansion_ratio);
      // keys of a bare block have no leading dot
      break;
    case LLIE_ATTN:
      if (!gn_ok(g.in_channels)) return LLIE_ERR_CONFIG;
      b.add_attn("", g.in_channels, g.num_attention_heads);
      break;
    case LLIE_DOWN: c->downs.push_back(b.add_conv3("down", g.in_channels)); break;
    case LLIE_UP: c->ups.push_back(b.add_conv3("conv", g.in_channels)); break;
    default: return LLIE_ERR_ARG;
  }
  b.finish_film();
  // strip the leading '.' that an empty prefix leaves on block keys
  c->index.clear();
  for (size_t i = 0; i < c->params.size(); ++i) {
    std::string& k = c->params[i].key;
    if (!k.empty() && k[0] == '.') k = k.substr(1);
    c->index[k] = (int)i;
  }
  c->blob_bytes = b.cursor;
  return LLIE_OK;
}

// ---------------------------------------------------------------------------------------------
// Run helpers.  In dry mode nothing is launched; only the arena is exercised.
// Fused expand+depthwise ("recompute" form, dwx.hip) for 2-byte dtypes.  Numerically equivalent, but as
// built it is slower than the unfused pair on MI355X (6.2 vs 3.6 ms/forward for the depthwise class at
// small@256 B=32 fp16; K1 without stores only drops 5.0 -> 4.6 ms), so it is opt-in: LLIE_DWX=1 or
// llie_tune("dwx", 1).  See DESIGN.md section 7.
bool g_use_dwx = getenv("LLIE_DWX") != nullptr;

struct Run {
  llie_ctx* c;
  Arena* ar;
  hipStream_t s;
  char* ws;
  bool dry;
  int B;
  int dt;
  hipError_t err = hipSuccess;

  template <typename T = void> T* wptr(size_t off) const { return reinterpret_cast<T*>(c->blob + off); }
  template <typename T = void> T* p(size_t off) const { return reinterpret_cast<T*>(ws + off); }
  void chk(hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; }
  size_t es() const { return elem_size(dt); }
  // launch `f` bracketed by HIP events on the launch stream when its class is being profiled
  template <typename F> void timed(int cls, int64_t bytes, F&& f) {
    if (!(c->prof_mask & cls) || c->prof.size() >= 8192) { chk(f()); return; }
    llie_ctx::ProfRec r{cls, bytes, c->get_event(), c->get_event(), ""};
    if (!r.e0 || !r.e1) { chk(f()); return; }
    chk(hipEventRecord(r.e0, s));
    chk(f());
    chk(hipEventRecord(r.e1, s));
    r.name = last_kernel();  // static storage: launchers pass string literals / function-local statics
    c->prof.push_back(r);
  }

  Tens new_tens(int C, int H, int W, int ntiles) {
    Tens t;
    t.C = C; t.H = H; t.W = W; t.ntiles = ntiles; t.valid = true;
    t.off = ar->alloc((size_t)B * H * W * C * es());
    t.slab = ar->alloc((size_t)B * ntiles * 2 * C * 4);
    return t;
  }
  void free_tens(Tens& t) {
    if (!t.valid) return;
    ar->free(t.off);
    ar->free(t.slab);
    t.valid = false;
  }
  StatSrc src(const Tens& t) const { return StatSrc{p<float>(t.slab), t.ntiles, t.C}; }

  // GroupNorm affine of (x0 [+ x1]) -> freshly allocated as/ab [B][C]; returns offsets
  void gn(const Tens& x0, const Tens* x1, size_t gamma, size_t beta, const float* film, int64_t film_stride,
          size_t& as, size_t& ab) {
    const int C = x0.C + (x1 ? x1->C : 0);
    as = ar->alloc((size_t)B * C * 4);
    ab = ar->alloc((size_t)B * C * 4);
    if (dry) return;
    GnFinalizeArgs a{};
    a.src[0] = src(x0);
    if (x1) a.src[1] = src(*x1);
    a.C = C; a.groups = 32; a.P = x0.H * x0.W;
    a.gamma = wptr<float>(gamma); a.beta = wptr<float>(beta);
    a.film = film; a.film_stride = film_stride; a.eps = 1e-5f;
    a.as = p<float>(as); a.ab = p<float>(ab); a.B = B;
    chk(launch_gn_finalize(a, s));
  }

  // InvertedResidualBlock.forward (efficient_unet.py:203-236) as 7 launches.
  Tens irb(const IrbW& w, const Tens& x0, const Tens* x1, const float* film, int64_t film_stride) {
    const int H = x0.H, W = x0.W, P = H * W, M = B * P;
    const int BM = pw_gemm_tile_rows(P);
    size_t as1, ab1;
    gn(x0, x1, w.n1g, w.n1b, nullptr, 0, as1, ab1);
    // Recompute form (2-byte T, narrow inputs): K1 only produces h1's statistics and the fused
    // expand+depthwise kernel rebuilds h1 on the fly, so the 4x-expanded tensor never touches HBM.
    const bool fused = g_use_dwx && dwx_supported(dt, w.cin, w.hid, H, W);
    // K1: expand with norm1 + ReLU6 prologue
    Tens h1;
    h1.C = w.hid; h1.H = H; h1.W = W; h1.ntiles = P / BM; h1.valid = true;
    h1.off = fused ? 0 : ar->alloc((size_t)B * P * w.hid * es());
    h1.slab = ar->alloc((size_t)B * h1.ntiles * 2 * w.hid * 4);
    if (!dry) {
      GemmArgs g{};
      g.seg[0] = GemmSeg{p(x0.off), x0.C, p<float>(as1), p<float>(ab1), w.cin, ACT_RELU6};
      g.nseg = 1;
      if (x1) {
        g.seg[1] = GemmSeg{p(x1->off), x1->C, p<float>(as1) + x0.C, p<float>(ab1) + x0.C, w.cin, ACT_RELU6};
        g.nseg = 2;
      }
      g.w = wptr(w.w_expand); g.out = fused ? nullptr : p(h1.off); g.stats = p<float>(h1.slab);
      g.M = M; g.N = w.hid; g.K = w.cin; g.P = P; g.nostore = fused ? 1 : 0;
      timed(LLIE_K_GEMM, ((int64_t)M * (w.cin + (fused ? 0 : w.hid)) + (int64_t)w.hid * w.cin) * (int64_t)es(),
            [&] { return launch_pw_gemm(dt, g, s); });
    }
    // norm2 + FiLM folded into one affine
    size_t as2, ab2;
    gn(h1, nullptr, w.n2g, w.n2b, film ? film + w.film_off : nullptr, film_stride, as2, ab2);
    // K2: depthwise with affine + ReLU6 prologue and SE pool partials
    const int dnt = dwconv_ntiles(H, W);
    const size_t h2 = ar->alloc((size_t)M * w.hid * es());
    const size_t pool = ar->alloc((size_t)B * dnt * w.hid * 4);
    if (!dry) {
      if (fused) {
        DwxArgs d{};
        d.x0 = p(x0.off); d.c0 = x0.C; d.x1 = x1 ? p(x1->off) : nullptr; d.c1 = x1 ? x1->C : 0;
        d.as1 = p<float>(as1); d.ab1 = p<float>(ab1); d.w1 = wptr(w.w_expand);
        d.as2 = p<float>(as2); d.ab2 = p<float>(ab2); d.wd = wptr<float>(w.w_dw);
        d.out = p(h2); d.pool = p<float>(pool); d.B = B; d.H = H; d.W = W; d.Chid = w.hid;
        timed(LLIE_K_DW, (int64_t)M * (w.cin + w.hid) * (int64_t)es(), [&] { return launch_dwx(dt, d, s); });
      } else {
        DwArgs d{};
        d.in = p(h1.off); d.out = p(h2); d.as = p<float>(as2); d.ab = p<float>(ab2);
        d.w = wptr<float>(w.w_dw); d.pool = p<float>(pool); d.B = B; d.H = H; d.W = W; d.C = w.hid;
        timed(LLIE_K_DW, 2LL * M * w.hid * (int64_t)es(), [&] { return launch_dwconv3x3(dt, d, s); });
      }
    }
    ar->free(as1); ar->free(ab1);
    if (!fused) ar->free(h1.off);
    ar->free(h1.slab);
    ar->free(as2); ar->free(ab2);
    // SE MLP
    const size_t sehid = ar->alloc((size_t)B * w.sq * 4), gate = ar->alloc((size_t)B * w.hid * 4);
    const size_t semean = ar->alloc((size_t)B * w.hid * 4);
    if (!dry) {
      SeArgs e{};
      e.pool = p<float>(pool); e.ntiles = dnt; e.P = P;
      e.w1 = wptr(w.se_w1); e.b1 = wptr<float>(w.se_b1); e.w2 = wptr(w.se_w2); e.b2 = wptr<float>(w.se_b2);
      e.mean = p<float>(semean); e.hid = p<float>(sehid); e.gate = p<float>(gate); e.B = B; e.C = w.hid; e.Cs = w.sq;
      timed(LLIE_K_SE, ((int64_t)B * dnt * w.hid * 4) + 2LL * w.hid * w.sq * (int64_t)es(), [&] {
        hipError_t r1 = launch_se_fc1(dt, e, s);
        return r1 != hipSuccess ? r1 : launch_se_fc2(dt, e, s);
      });
    }
    ar->free(pool); ar->free(sehid); ar->free(semean);
    // K3: project with SE gate prologue (+ skip conv as extra K segments, or identity residual)
    Tens y = new_tens(w.cout, H, W, P / BM);
    if (!dry) {
      GemmArgs g{};
      g.seg[0] = GemmSeg{p(h2), w.hid, p<float>(gate), nullptr, w.hid, ACT_NONE};
      g.nseg = 1;
      g.K = w.hid;
      if (w.skip) {
        g.seg[g.nseg++] = GemmSeg{p(x0.off), x0.C, nullptr, nullptr, 0, ACT_NONE};
        if (x1) g.seg[g.nseg++] = GemmSeg{p(x1->off), x1->C, nullptr, nullptr, 0, ACT_NONE};
        g.K += w.cin;
      } else {
        g.res = p(x0.off);
      }
      g.w = wptr(w.w_proj); g.out = p(y.off); g.stats = p<float>(y.slab);
      g.M = M; g.N = w.cout; g.P = P;
      timed(LLIE_K_GEMM, ((int64_t)M * (g.K + w.cout + (w.skip ? 0 : w.cout)) + (int64_t)w.cout * g.K) * (int64_t)es(),
            [&] { return launch_pw_gemm(dt, g, s); });
    }
    ar->free(h2); ar->free(gate);
    return y;
  }

  // LinearAttention.forward (efficient_unet.py:273-308)
  Tens attn(const AttnW& w, const Tens& x) {
    const int H = x.H, W = x.W, N = H * W, M = B * N;
    const int BM = pw_gemm_tile_rows(N);
    size_t as, ab;
    gn(x, nullptr, w.ng, w.nb, nullptr, 0, as, ab);
    const size_t qkv = ar->alloc((size_t)M * 3 * w.inner * es());
    if (!dry) {
      GemmArgs g{};
      g.seg[0] = GemmSeg{p(x.off), x.C, p<float>(as), p<float>(ab), x.C, ACT_NONE};
      g.nseg = 1; g.w = wptr(w.w_qkv); g.out = p(qkv);
      g.M = M; g.N = 3 * w.inner; g.K = x.C; g.P = N;
      chk(launch_pw_gemm(dt, g, s));
    }
    ar->free(as); ar->free(ab);
    const int nsplit = linattn_nsplit(N);
    const size_t kv = ar->alloc((size_t)nsplit * B * w.heads * 32 * 33 * 4);
    const size_t ao = ar->alloc((size_t)M * w.inner * es());
    if (!dry) {
      AttnArgs a{};
      a.qkv = p(qkv); a.B = B; a.N = N; a.heads = w.heads; a.kv = p<float>(kv); a.out = p(ao); a.nsplit = nsplit;
      chk(launch_linattn_kv(dt, a, s));
      chk(launch_linattn_out(dt, a, s));
    }
    ar->free(qkv); ar->free(kv);
    Tens tmp = new_tens(x.C, H, W, N / BM);
    if (!dry) {
      GemmArgs g{};
      g.seg[0] = GemmSeg{p(ao), w.inner, nullptr, nullptr, 0, ACT_NONE};
      g.nseg = 1; g.w = wptr(w.w_out); g.out = p(tmp.off); g.stats = p<float>(tmp.slab);
      g.M = M; g.N = x.C; g.K = w.inner; g.P = N;
      chk(launch_pw_gemm(dt, g, s));
    }
    ar->free(ao);
    size_t as2, ab2;
    gn(tmp, nullptr, w.n2g, w.n2b, nullptr, 0, as2, ab2);
    Tens y = new_tens(x.C, H, W, N / kAffineTileRows);
    if (!dry) {
      AffineAddArgs a{};
      a.x = p(tmp.off); a.as = p<float>(as2); a.ab = p<float>(ab2); a.res = p(x.off); a.y = p(y.off);
      a.stats = p<float>(y.slab); a.M = M; a.C = x.C; a.P = N;
      chk(launch_affine_add(dt, a, s));
    }
    free_tens(tmp);
    ar->free(as2); ar->free(ab2);
    return y;
  }

  Tens conv3(const ConvW& w, const Tens& x, int mode) {
    const int Ho = mode == 0 ? x.H / 2 : x.H * 2, Wo = mode == 0 ? x.W / 2 : x.W * 2;
    Tens y = new_tens(w.c, Ho, Wo, conv3x3_ntiles(Ho, Wo));
    if (!dry) {
      Conv3Args a{};
      a.in = p(x.off); a.w = wptr(w.w); a.bias = wptr<float>(w.bias); a.out = p(y.off); a.stats = p<float>(y.slab);
      a.B = B; a.Hi = x.H; a.Wi = x.W; a.Cin = w.c; a.Cout = w.c; a.mode = mode;
      timed(LLIE_K_CONV3, ((int64_t)B * w.c * ((int64_t)x.H * x.W + (int64_t)Ho * Wo) + 9LL * w.c * w.c) * (int64_t)es(),
            [&] { return launch_conv3x3(dt, a, s); });
    }
    return y;
  }

  Tens run_blocks(const std::vector<Block>& blocks, Tens h, const Tens* cat, const float* film, int64_t fstride,
                  bool keep_input) {
    bool first = true;
    for (const Block& b : blocks) {
      Tens y = b.kind == 0 ? irb(c->irbs[b.idx], h, first ? cat : nullptr, film, fstride) : attn(c->attns[b.idx], h);
      if (!(first && keep_input)) free_tens(h);
      h = y;
      first = false;
    }
    return h;
  }

  // EfficientUNet.forward (efficient_unet.py:532-606)
  // `fs` (optional): scheduler step fused into the final conv's epilogue (2-byte compute dtypes only)
  struct FusedStep { StepCoef coef; const float* noise; float* prev; float* clamped; };
  void unet(const float* lat, const float* cond, const int64_t* t, int uniform_t, float* eps, const FusedStep* fs = nullptr) {
    const llie_config& g = c->cfg;
    const int S = g.image_size, T = g.time_embed_dim, F = c->film_rows;
    const int rows = uniform_t ? 1 : B;
    const size_t temb = ar->alloc((size_t)rows * T * 4), stemb = ar->alloc((size_t)rows * T * 4);
    const size_t film = ar->alloc((size_t)rows * F * 4);
    if (!dry) {
      TimeArgs ta{};
      ta.t = t; ta.rows = rows; ta.dim = g.base_channels; ta.freqs = wptr<float>(c->freqs); ta.T = T;
      ta.w1 = wptr<float>(c->t_w1); ta.b1 = wptr<float>(c->t_b1); ta.w3 = wptr<float>(c->t_w3); ta.b3 = wptr<float>(c->t_b3);
      ta.temb = p<float>(temb); ta.silu_temb = p<float>(stemb);
      chk(launch_time_embed(ta, s));
      FilmArgs fa{};
      fa.silu_temb = p<float>(stemb); fa.rows = rows; fa.T = T; fa.wf = wptr<float>(c->film_w); fa.bf = wptr<float>(c->film_b);
      fa.film = p<float>(film); fa.F = F;
      chk(launch_film(fa, s));
    }
    const float* filmp = p<float>(film);
    const int64_t fstride = uniform_t ? 0 : F;

    Tens h = new_tens(c->channels[0], S, S, init_conv_ntiles(S, S));
    if (!dry) {
      InitConvArgs a{};
      const int half = g.in_channels / 2;
      a.x0 = lat; a.x1 = cond; a.c0 = half; a.c1 = g.in_channels - half;
      a.w = wptr<float>(c->init_w); a.bias = wptr<float>(c->init_b); a.out = p(h.off); a.stats = p<float>(h.slab);
      a.wp = dt != LLIE_F32 ? wptr(c->init_wp) : nullptr;
      a.B = B; a.H = S; a.W = S; a.Cout = c->channels[0];
      chk(launch_init_conv(dt, a, s));
    }
    Tens skips[4];
    for (int l = 0; l < 4; ++l) {
      h = run_blocks(c->enc[l], h, nullptr, filmp, fstride, false);
      skips[l] = h;  // one skip per level, taken before the downsample (:567)
      if (l < 3) h = conv3(c->downs[l], h, 0);  // the skip stays alive
    }
    // level 3: h aliases skips[3]; mid_block1 must not free it
    h = run_blocks(c->mid, h, nullptr, filmp, fstride, true);
    for (int l = 0; l < 4; ++l) {
      if (l > 0) {
        Tens u = conv3(c->ups[l - 1], h, 1);
        free_tens(h);
        h = u;
      }
      Tens y = run_blocks(c->dec[l], h, &skips[3 - l], filmp, fstride, false);  // cat([h, skip]) (:588)
      free_tens(skips[3 - l]);
      h = y;
    }
    size_t as, ab;
    gn(h, nullptr, c->fin_g, c->fin_b, nullptr, 0, as, ab);
    if (!dry) {
      FinalConvArgs a{};
      a.in = p(h.off); a.as = p<float>(as); a.ab = p<float>(ab); a.w = wptr<float>(c->fin_w); a.bias = wptr<float>(c->fin_bias);
      a.out = eps; a.B = B; a.H = S; a.W = S; a.C = c->channels[0]; a.Cout = g.out_channels;
      a.wp = dt != LLIE_F32 ? wptr(c->fin_wp) : nullptr;
      if (fs) {
        a.fuse_step = 1; a.coef = fs->coef; a.sample = lat; a.noise = fs->noise; a.prev = fs->prev; a.clamped = fs->clamped;
      }
      chk(launch_final_conv(dt, a, s));
    }
    free_tens(h);
    ar->free(as); ar->free(ab);
    ar->free(temb); ar->free(stemb); ar->free(film);
  }

  // single-operator forward: fp32 NCHW in/out
  void module(const float* x, const float* temb, float* y, int H, int W) {
    const llie_config& g = c->cfg;
    const int P = H * W;
    const int split = (g.kind == LLIE_IRB) ? g.base_channels : 0;  // IRB: optional virtual-concat split point
    Tens x0 = new_tens(split ? split : g.in_channels, H, W, P / 64);
    Tens x1;
    if (split) x1 = new_tens(g.in_channels - split, H, W, P / 64);
    if (!dry) {
      chk(launch_nchw_to_nhwc(dt, x, p(x0.off), p<float>(x0.slab), B, x0.C, P, g.in_channels, 0, s));
      if (split) chk(launch_nchw_to_nhwc(dt, x, p(x1.off), p<float>(x1.slab), B, x1.C, P, g.in_channels, split, s));
    }
    Tens out;
    if (g.kind == LLIE_IRB) {
      const int T = g.time_embed_dim, F = c->film_rows;
      const size_t st = ar->alloc((size_t)B * T * 4), film = ar->alloc((size_t)B * F * 4);
      if (!dry) {
        chk(launch_silu_rows(temb, p<float>(st), (int64_t)B * T, s));
        FilmArgs fa{};
        fa.silu_temb = p<float>(st); fa.rows = B; fa.T = T; fa.wf = wptr<float>(c->film_w); fa.bf = wptr<float>(c->film_b);
        fa.film = p<float>(film); fa.F = F;
        chk(launch_film(fa, s));
      }
      out = irb(c->irbs[0], x0, split ? &x1 : nullptr, p<float>(film), F);
      ar->free(st); ar->free(film);
    } else if (g.kind == LLIE_ATTN) {
      out = attn(c->attns[0], x0);
    } else if (g.kind == LLIE_DOWN) {
      out = conv3(c->downs[0], x0, 0);
    } else {
      out = conv3(c->ups[0], x0, 1);
    }
    if (!dry) chk(launch_nhwc_to_nchw(dt, p(out.off), y, B, out.C, out.H * out.W, s));
    free_tens(out);
    free_tens(x0);
    free_tens(x1);
  }
};

int check_loaded(const llie_ctx* c) {
  for (const Param& p : c->params)
    if (!p.loaded) {
      set_err("parameter '%s' was never loaded", p.key.c_str());
      return LLIE_ERR_NOT_LOADED;
    }
  return LLIE_OK;
}

int finish_run(Run& r, int64_t ws_bytes) {
  if (r.ar->failed) {
    set_err("workspace too small: have %lld bytes", (long long)ws_bytes);
    return LLIE_ERR_WORKSPACE;
  }
  if (r.err != hipSuccess) {
    set_err("HIP error %d: %s", (int)r.err, hipGetErrorString(r.err));
    return (int)r.err;
  }
  return LLIE_OK;
}

int shape_ok(const llie_ctx* c, int H, int W) {
  const int k = c->cfg.kind;
  int minside = 8;
  if (k == LLIE_DOWN) minside = 16;
  if (H % 8 || W % 8 || H < minside || W < minside || (H * W) % 64) {
    set_err("unsupported spatial size %dx%d", H, W);
    return LLIE_ERR_SHAPE;
  }
  return LLIE_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

const char* llie_last_error(void) { return g_err; }
const char* llie_version(void) { return "llie-hip 0.1 (gfx950)"; }

int llie_create(const llie_config* cfg, llie_ctx** out) {
  if (!cfg || !out) return LLIE_ERR_ARG;
  if (cfg->compute_dtype < 0 || cfg->compute_dtype > 2) { set_err("bad compute_dtype"); return LLIE_ERR_ARG; }
  llie_ctx* c = new llie_ctx();
  c->cfg = *cfg;
  c->dt = cfg->compute_dtype;
  const int rc = cfg->kind == LLIE_UNET ? build_unet(c) : build_module(c);
  if (rc != LLIE_OK) {
    if (rc == LLIE_ERR_CONFIG) set_err("num_channels must be divisible by num_groups");  // nn.GroupNorm's ValueError
    else set_err("unsupported configuration");
    delete c;
    return rc;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    // No device (CPU-only build container): the handle still describes the state_dict, but cannot
    // hold weights or run.  Loading / forward report LLIE_ERR_NO_DEVICE.
    c->blob = nullptr;
    *out = c;
    return LLIE_OK;
  }
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->blob), c->blob_bytes ? c->blob_bytes : 256);
  if (e != hipSuccess) { set_err("hipMalloc(%zu) failed: %s", c->blob_bytes, hipGetErrorString(e)); delete c; return (int)e; }
  e = hipMemset(c->blob, 0, c->blob_bytes);
  if (e != hipSuccess) { set_err("hipMemset failed"); (void)hipFree(c->blob); delete c; return (int)e; }
  if (cfg->kind == LLIE_UNET) {
    // SinusoidalPosEmb frequencies (efficient_unet.py:70-73), tabulated once
    const int half = cfg->base_channels / 2;
    std::vector<float> f(half);
    // same fp32 operation chain as torch.exp(-math.log(10000) * torch.arange(half) / half)
    const float neg_ln = (float)(-std::log(10000.0));
    for (int i = 0; i < half; ++i) {
      const float q = (neg_ln * (float)i) / (float)half;
      f[i] = (float)std::exp((double)q);
    }
    e = hipMemcpy(c->blob + c->freqs, f.data(), half * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_err("hipMemcpy failed"); (void)hipFree(c->blob); delete c; return (int)e; }
  }
  *out = c;
  return LLIE_OK;
}

void llie_destroy(llie_ctx* c) {
  if (!c) return;
  for (auto& kv : c->graphs) {
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
  }
  if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
  for (auto& r : c->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  if (c->blob) (void)hipFree(c->blob);
  delete c;
}

int llie_num_params(const llie_ctx* c) { return c ? (int)c->params.size() : LLIE_ERR_ARG; }

int llie_param_info(const llie_ctx* c, int i, char* key, size_t cap, int64_t* numel, int* ndim, int64_t* shape4) {
  if (!c || i < 0 || i >= (int)c->params.size()) return LLIE_ERR_ARG;
  if (key && cap) {
    strncpy(key, c->params[i].key.c_str(), cap - 1);
    key[cap - 1] = 0;
  }
  if (numel) *numel = c->params[i].numel;
  if (ndim) *ndim = c->params[i].ndim;
  if (shape4)
    for (int d = 0; d < 4; ++d) shape4[d] = c->params[i].shape[d];
  return LLIE_OK;
}

int llie_load_param(llie_ctx* c, const char* key, const float* src, int64_t numel, llie_stream stream) {
  if (!c || !key || !src) return LLIE_ERR_ARG;
  if (!c->blob) { set_err("no HIP device"); return LLIE_ERR_NO_DEVICE; }
  auto it = c->index.find(key);
  if (it == c->index.end()) { set_err("unexpected key '%s'", key); return LLIE_ERR_KEY; }
  Param& p = c->params[it->second];
  if (p.numel != numel) { set_err("size mismatch for '%s': expected %lld elements, got %lld", key, (long long)p.numel, (long long)numel); return LLIE_ERR_KEY; }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  void* dst = c->blob + p.off;
  hipError_t e = hipSuccess;
  switch (p.kind) {
    case PK_F32: e = hipMemcpyAsync(dst, src, (size_t)numel * 4, hipMemcpyDeviceToDevice, s); break;
    case PK_MAT: e = launch_cvt_rows(p.as_t ? c->dt : 0, src, dst, p.rows, p.cols, p.ld, p.col0, s); break;
    case PK_CONV3: e = launch_repack_conv3x3(c->dt, src, dst, p.O, p.I, s); break;
    case PK_DW: e = launch_repack_dw(src, reinterpret_cast<float*>(dst), p.O, s); break;
    case PK_INIT:
      e = launch_repack_init(src, reinterpret_cast<float*>(dst), p.O, p.I, s);
      if (e == hipSuccess && c->dt != LLIE_F32) e = launch_repack_init_mfma(c->dt, src, c->blob + c->init_wp, p.O, p.I, s);
      break;
    case PK_FINAL:
      e = launch_repack_final(src, reinterpret_cast<float*>(dst), p.O, p.I, s);
      if (e == hipSuccess && c->dt != LLIE_F32) e = launch_repack_final_mfma(c->dt, src, c->blob + c->fin_wp, p.O, p.I, s);
      break;
  }
  if (e != hipSuccess) { set_err("repack of '%s' failed: %s", key, hipGetErrorString(e)); return (int)e; }
  p.loaded = true;
  return LLIE_OK;
}

int llie_params_loaded(const llie_ctx* c) {
  if (!c) return 0;
  for (const Param& p : c->params)
    if (!p.loaded) return 0;
  return 1;
}

int64_t llie_workspace_bytes(llie_ctx* c, int batch, int height, int width) {
  if (!c || batch <= 0) return LLIE_ERR_ARG;
  Arena ar((size_t)1 << 46);
  Run r{c, &ar, nullptr, nullptr, true, batch, c->dt};
  if (c->cfg.kind == LLIE_UNET) {
    r.unet(nullptr, nullptr, nullptr, 0, nullptr);
    // + latents ping-pong and eps buffers for llie_enhance
    const size_t img = align_up((size_t)batch * 3 * c->cfg.image_size * c->cfg.image_size * 4, 256);
    return (int64_t)(ar.high + 3 * img);
  }
  if (shape_ok(c, height, width) != LLIE_OK) return LLIE_ERR_SHAPE;
  r.module(nullptr, nullptr, nullptr, height, width);
  return (int64_t)ar.high;
}

// Workspace for llie_enhance with room for the hipGraph staging area (inputs/outputs of up to
// `max_steps` steps with intermediates and noise predictions).
int64_t llie_enhance_workspace_bytes(llie_ctx* c, int batch, int max_steps) {
  if (!c || batch <= 0 || max_steps <= 0 || c->cfg.kind != LLIE_UNET) return LLIE_ERR_ARG;
  const int64_t core = llie_workspace_bytes(c, batch, 0, 0);
  if (core < 0) return core;
  const size_t img = align_up((size_t)batch * 3 * c->cfg.image_size * c->cfg.image_size * 4, 256);
  return core + (int64_t)((2 + 3 * (size_t)max_steps) * img + align_up((size_t)max_steps * batch * 8, 256));
}

static int unet_forward_impl(llie_ctx* c, const float* lat, const float* cond, const int64_t* t, int uniform_t, float* eps,
                             const Run::FusedStep* fs, int batch, void* ws, int64_t ws_bytes, llie_stream stream);

int llie_unet_forward(llie_ctx* c, const float* lat, const float* cond, const int64_t* t, int uniform_t, float* eps,
                      int batch, void* ws, int64_t ws_bytes, llie_stream stream) {
  if (!eps) return LLIE_ERR_ARG;
  return unet_forward_impl(c, lat, cond, t, uniform_t, eps, nullptr, batch, ws, ws_bytes, stream);
}

static int unet_forward_impl(llie_ctx* c, const float* lat, const float* cond, const int64_t* t, int uniform_t, float* eps,
                             const Run::FusedStep* fs, int batch, void* ws, int64_t ws_bytes, llie_stream stream) {
  if (!c || !lat || !cond || !t || (!eps && !fs) || !ws || batch <= 0 || c->cfg.kind != LLIE_UNET) return LLIE_ERR_ARG;
  if (!c->blob) { set_err("no HIP device"); return LLIE_ERR_NO_DEVICE; }
  int rc = check_loaded(c);
  if (rc) return rc;
  Arena ar((size_t)ws_bytes);
  Run r{c, &ar, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<char*>(ws), false, batch, c->dt};
  {  // capacity check first, so that no kernel ever sees an offset past the workspace
    Arena probe((size_t)1 << 46);
    Run d{c, &probe, nullptr, nullptr, true, batch, c->dt};
    d.unet(nullptr, nullptr, nullptr, uniform_t, nullptr);
    if ((int64_t)probe.high > ws_bytes) { set_err("workspace too small: need %zu, have %lld", probe.high, (long long)ws_bytes); return LLIE_ERR_WORKSPACE; }
  }
  r.unet(lat, cond, t, uniform_t, eps, fs);
  return finish_run(r, ws_bytes);
}

int llie_module_forward(llie_ctx* c, const float* x, const float* temb, float* y, int batch, int H, int W, void* ws,
                        int64_t ws_bytes, llie_stream stream) {
  if (!c || !x || !y || !ws || batch <= 0 || c->cfg.kind == LLIE_UNET) return LLIE_ERR_ARG;
  if (c->cfg.kind == LLIE_IRB && !temb) return LLIE_ERR_ARG;
  if (!c->blob) { set_err("no HIP device"); return LLIE_ERR_NO_DEVICE; }
  int rc = check_loaded(c);
  if (rc) return rc;
  rc = shape_ok(c, H, W);
  if (rc) return rc;
  {
    Arena probe((size_t)1 << 46);
    Run d{c, &probe, nullptr, nullptr, true, batch, c->dt};
    d.module(nullptr, nullptr, nullptr, H, W);
    if ((int64_t)probe.high > ws_bytes) { set_err("workspace too small: need %zu, have %lld", probe.high, (long long)ws_bytes); return LLIE_ERR_WORKSPACE; }
  }
  Arena ar((size_t)ws_bytes);
  Run r{c, &ar, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<char*>(ws), false, batch, c->dt};
  r.module(x, temb, y, H, W);
  return finish_run(r, ws_bytes);
}

int llie_lcm_step(const float* mo, const float* sample, const float* noise, float* prev, float* x0, float* clamped,
                  int64_t n, const llie_step_coef* k, llie_stream stream) {
  if (!mo || !sample || !prev || !k || n <= 0) return LLIE_ERR_ARG;
  if (!k->is_last && !noise) return LLIE_ERR_ARG;
  StepCoef c{k->sqrt_alpha_t, k->sqrt_beta_t, k->sqrt_alpha_prev, k->sqrt_beta_prev, k->is_last, k->v_prediction, k->clamp_x0};
  hipError_t e = launch_lcm_step(mo, sample, noise, prev, x0, clamped, n, c, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) { set_err("lcm_step: %s", hipGetErrorString(e)); return (int)e; }
  return LLIE_OK;
}

int llie_add_noise(const float* x0, const float* noise, const int64_t* t, const float* acp, float* out, int batch,
                   int64_t per, int velocity, llie_stream stream) {
  if (!x0 || !noise || !t || !acp || !out || batch <= 0 || per <= 0) return LLIE_ERR_ARG;
  hipError_t e = launch_add_noise(x0, noise, t, acp, out, batch, per, velocity, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) { set_err("add_noise: %s", hipGetErrorString(e)); return (int)e; }
  return LLIE_OK;
}

// The launch sequence of LowLightDiffusion.enhance (low_light_diffusion.py:204-240): `steps` x
// (UNet forward, scheduler step).  `base` holds two latent ping-pong images and one eps image,
// followed by the UNet workspace.
static int enhance_sequence(llie_ctx* c, const float* low, const float* noise, const int64_t* t_dev,
                            const llie_step_coef* coefs, int steps, float* enhanced, float* inter, float* preds,
                            int batch, char* base, int64_t ws_bytes, llie_stream stream) {
  const int S = c->cfg.image_size;
  const int64_t n = (int64_t)batch * 3 * S * S;
  const size_t img = align_up((size_t)n * 4, 256);
  float* lat[2] = {reinterpret_cast<float*>(base), reinterpret_cast<float*>(base + img)};
  float* eps_ws = reinterpret_cast<float*>(base + 2 * img);
  void* uws = base + 3 * img;
  const int64_t uws_bytes = ws_bytes - (int64_t)(3 * img);
  const float* cur = noise;  // initial latents = first draw (low_light_diffusion.py:208-211)
  const bool fuse = c->dt != LLIE_F32;  // the MFMA output head applies the scheduler step in its epilogue
  for (int i = 0; i < steps; ++i) {
    const bool last = i == steps - 1;
    float* prev = inter ? inter + (size_t)i * n : lat[i & 1];
    const float* nz = coefs[i].is_last ? nullptr : noise + (size_t)(i + 1) * n;
    if (!coefs[i].is_last && i + 1 >= steps) return LLIE_ERR_ARG;  // a non-final step needs a noise draw
    int rc;
    if (fuse) {
      Run::FusedStep fs{StepCoef{coefs[i].sqrt_alpha_t, coefs[i].sqrt_beta_t, coefs[i].sqrt_alpha_prev, coefs[i].sqrt_beta_prev,
                                 coefs[i].is_last, coefs[i].v_prediction, coefs[i].clamp_x0},
                        nz, prev, last ? enhanced : nullptr};
      rc = unet_forward_impl(c, cur, low, t_dev + (size_t)i * batch, 1, preds ? preds + (size_t)i * n : nullptr, &fs, batch,
                             uws, uws_bytes, stream);
      if (rc) return rc;
    } else {
      float* eps = preds ? preds + (size_t)i * n : eps_ws;
      rc = llie_unet_forward(c, cur, low, t_dev + (size_t)i * batch, 1, eps, batch, uws, uws_bytes, stream);
      if (rc) return rc;
      rc = llie_lcm_step(eps, cur, nz, prev, nullptr, last ? enhanced : nullptr, n, &coefs[i], stream);
      if (rc) return rc;
    }
    cur = prev;
  }
  return LLIE_OK;
}

int llie_enhance(llie_ctx* c, const float* low, const float* noise, const int64_t* t_dev, const llie_step_coef* coefs,
                 int steps, float* enhanced, float* inter, float* preds, int batch, void* ws, int64_t ws_bytes,
                 llie_stream stream) {
  if (!c || !low || !noise || !t_dev || !coefs || !enhanced || !ws || steps <= 0 || batch <= 0 || c->cfg.kind != LLIE_UNET)
    return LLIE_ERR_ARG;
  const int S = c->cfg.image_size;
  const int64_t n = (int64_t)batch * 3 * S * S;
  const size_t img = align_up((size_t)n * 4, 256);
  if ((int64_t)(3 * img) > ws_bytes) { set_err("workspace too small"); return LLIE_ERR_WORKSPACE; }
  char* base = reinterpret_cast<char*>(ws);
  hipStream_t hs = reinterpret_cast<hipStream_t>(stream);

  // ---- hipGraph path: the ~800 launches of a 4-step loop are launch-bound in their runs of tiny
  // kernels (GroupNorm finalize, SE MLP).  The sequence is captured once per (shape, schedule,
  // workspace) with every pointer inside the workspace: user tensors are staged in/out by plain
  // async copies around the graph launch.  First use of a key runs eagerly (it also performs the
  // one-time hipFuncSetAttribute calls, which must not happen during capture).
  static const bool no_graph = getenv("LLIE_NO_GRAPH") != nullptr;
  const size_t n_in = 1 + (size_t)steps;                       // low + noise draws
  const size_t n_out = 1 + (inter ? steps : 0) + (preds ? steps : 0);
  const size_t tbytes = align_up((size_t)steps * batch * 8, 256);
  const size_t stage = (n_in + n_out) * img + tbytes;
  const int64_t seq_bytes = ws_bytes - (int64_t)stage;
  bool use_graph = !no_graph && c->prof_mask == 0 && seq_bytes >= llie_workspace_bytes(c, batch, 0, 0);
  if (!use_graph) return enhance_sequence(c, low, noise, t_dev, coefs, steps, enhanced, inter, preds, batch, base, ws_bytes, stream);

  std::string key(reinterpret_cast<const char*>(coefs), sizeof(llie_step_coef) * steps);
  char tail[128];
  snprintf(tail, sizeof tail, "|%d|%d|%d|%d|%p|%lld", batch, steps, inter ? 1 : 0, preds ? 1 : 0, ws, (long long)ws_bytes);
  key += tail;
  llie_ctx::GraphEntry& ge = c->graphs[key];
  if (!ge.seen) {
    ge.seen = true;
    return enhance_sequence(c, low, noise, t_dev, coefs, steps, enhanced, inter, preds, batch, base, ws_bytes, stream);
  }
  // staging area at the tail of the workspace
  char* st = base + seq_bytes;
  float* s_low = reinterpret_cast<float*>(st);
  float* s_noise = reinterpret_cast<float*>(st + img);
  float* s_enh = reinterpret_cast<float*>(st + n_in * img);
  float* s_inter = inter ? reinterpret_cast<float*>(st + (n_in + 1) * img) : nullptr;
  float* s_preds = preds ? reinterpret_cast<float*>(st + (n_in + 1 + (inter ? steps : 0)) * img) : nullptr;
  int64_t* s_t = reinterpret_cast<int64_t*>(st + (n_in + n_out) * img);
  // NB: staged noise / inter / preds are step-major with stride `img` >= n*4; keep them dense (img == n*4 when n*4 % 256 == 0)
  if (img != (size_t)n * 4) return enhance_sequence(c, low, noise, t_dev, coefs, steps, enhanced, inter, preds, batch, base, ws_bytes, stream);
  hipError_t e = hipMemcpyAsync(s_low, low, (size_t)n * 4, hipMemcpyDeviceToDevice, hs);
  if (e == hipSuccess) e = hipMemcpyAsync(s_noise, noise, (size_t)n * 4 * steps, hipMemcpyDeviceToDevice, hs);
  if (e == hipSuccess) e = hipMemcpyAsync(s_t, t_dev, (size_t)steps * batch * 8, hipMemcpyDeviceToDevice, hs);
  if (e != hipSuccess) { set_err("enhance staging: %s", hipGetErrorString(e)); return (int)e; }
  if (!ge.exec) {
    if (!c->cap_stream) {
      e = hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking);
      if (e != hipSuccess) { set_err("hipStreamCreate: %s", hipGetErrorString(e)); return (int)e; }
    }
    e = hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { set_err("hipStreamBeginCapture: %s", hipGetErrorString(e)); return (int)e; }
    const int rc = enhance_sequence(c, s_low, s_noise, s_t, coefs, steps, s_enh, s_inter, s_preds, batch, base, seq_bytes,
                                    reinterpret_cast<llie_stream>(c->cap_stream));
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(c->cap_stream, &g);
    if (rc != LLIE_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess || !g) { set_err("hipStreamEndCapture: %s", hipGetErrorString(e)); return (int)(e ? e : hipErrorUnknown); }
    e = hipGraphInstantiate(&ge.exec, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(g); ge.exec = nullptr; set_err("hipGraphInstantiate: %s", hipGetErrorString(e)); return (int)e; }
    ge.graph = g;
  }
  e = hipGraphLaunch(ge.exec, hs);
  if (e == hipSuccess) e = hipMemcpyAsync(enhanced, s_enh, (size_t)n * 4, hipMemcpyDeviceToDevice, hs);
  if (e == hipSuccess && inter) e = hipMemcpyAsync(inter, s_inter, (size_t)n * 4 * steps, hipMemcpyDeviceToDevice, hs);
  if (e == hipSuccess && preds) e = hipMemcpyAsync(preds, s_preds, (size_t)n * 4 * steps, hipMemcpyDeviceToDevice, hs);
  if (e != hipSuccess) { set_err("enhance graph launch: %s", hipGetErrorString(e)); return (int)e; }
  return LLIE_OK;
}

// ---- kernel-level entry points (unit tests, tuning): thin wrappers over the launch API
int llie_pw_gemm(int dtype, const llie_gemm_seg* segs, int nseg, const void* w, const float* bias, const void* residual,
                 void* out, float* stats, int M, int N, int P, llie_stream stream) {
  if (!segs || nseg < 1 || nseg > 3 || !w || !out || dtype < 0 || dtype > 2) return LLIE_ERR_ARG;
  GemmArgs g{};
  g.nseg = nseg;
  for (int i = 0; i < nseg; ++i) {
    g.seg[i] = GemmSeg{segs[i].ptr, segs[i].channels, segs[i].scale, segs[i].bias, segs[i].affine_ld, segs[i].act};
    g.K += segs[i].channels;
  }
  g.w = w; g.bias = bias; g.res = residual; g.out = out; g.stats = stats; g.M = M; g.N = N; g.P = P;
  hipError_t e = launch_pw_gemm(dtype, g, reinterpret_cast<hipStream_t>(stream));
  if (e == hipErrorInvalidValue) { set_err("pw_gemm: shape outside the kernel contract"); return LLIE_ERR_SHAPE; }
  if (e != hipSuccess) { set_err("pw_gemm: %s", hipGetErrorString(e)); return (int)e; }
  return LLIE_OK;
}

int llie_dwconv3x3(int dtype, const void* in, void* out, const float* scale, const float* bias, const float* w9c,
                   float* pool, int B, int H, int W, int C, llie_stream stream) {
  if (!in || !out || !scale || !bias || !w9c || dtype < 0 || dtype > 2) return LLIE_ERR_ARG;
  DwArgs d{};
  d.in = in; d.out = out; d.as = scale; d.ab = bias; d.w = w9c; d.pool = pool; d.B = B; d.H = H; d.W = W; d.C = C;
  hipError_t e = launch_dwconv3x3(dtype, d, reinterpret_cast<hipStream_t>(stream));
  if (e == hipErrorInvalidValue) { set_err("dwconv3x3: shape outside the kernel contract"); return LLIE_ERR_SHAPE; }
  if (e != hipSuccess) { set_err("dwconv3x3: %s", hipGetErrorString(e)); return (int)e; }
  return LLIE_OK;
}

int llie_preprocess_u8(const uint8_t* img, int batch, int H0, int W0, float* out, int S, llie_stream stream) {
  if (!img || !out) return LLIE_ERR_ARG;
  hipError_t e = launch_preprocess_u8(img, batch, H0, W0, out, S, reinterpret_cast<hipStream_t>(stream));
  if (e == hipErrorInvalidValue) return LLIE_ERR_ARG;
  if (e != hipSuccess) { set_err("preprocess_u8: %s", hipGetErrorString(e)); return (int)e; }
  return LLIE_OK;
}
int llie_postprocess_u8(const float* x, int batch, int S, uint8_t* img, int H0, int W0, llie_stream stream) {
  if (!img || !x) return LLIE_ERR_ARG;
  hipError_t e = launch_postprocess_u8(x, batch, S, img, H0, W0, reinterpret_cast<hipStream_t>(stream));
  if (e == hipErrorInvalidValue) return LLIE_ERR_ARG;
  if (e != hipSuccess) { set_err("postprocess_u8: %s", hipGetErrorString(e)); return (int)e; }
  return LLIE_OK;
}

int llie_dwconv3x3_tiles(int H, int W) { return dwconv_ntiles(H, W); }
int llie_pw_gemm_tile_rows(int P) { return pw_gemm_tile_rows(P); }

int llie_tune(const char* knob, int value) {
  if (!knob) return LLIE_ERR_ARG;
  if (!strcmp(knob, "gemm_bk")) { pw_gemm_force_bk(value); return LLIE_OK; }
  if (!strcmp(knob, "gemm_v2")) { pw_gemm_use_v2(value); return LLIE_OK; }
  if (!strcmp(knob, "dwx")) { g_use_dwx = value != 0; return LLIE_OK; }
  if (!strcmp(knob, "gemm_ablate")) { pw_gemm_debug(value); return LLIE_OK; }
  if (!strcmp(knob, "dw_ablate")) { dwconv_debug(value); return LLIE_OK; }
  return LLIE_ERR_ARG;
}

int llie_profile_begin(llie_ctx* c, int class_mask) {
  if (!c) return LLIE_ERR_ARG;
  for (auto& r : c->prof) { c->event_pool.push_back(r.e0); c->event_pool.push_back(r.e1); }
  c->prof.clear();
  c->prof_mask = class_mask;
  return LLIE_OK;
}

int llie_profile_end(llie_ctx* c, int kernel_class, double* total_ms, int64_t* launches, int64_t* alg_bytes) {
  if (!c) return LLIE_ERR_ARG;
  c->prof_mask = 0;
  double ms = 0.0;
  int64_t n = 0, bytes = 0;
  for (auto& r : c->prof) {
    if (!(r.cls & kernel_class)) continue;
    hipError_t e = hipEventSynchronize(r.e1);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, r.e0, r.e1);
    if (e != hipSuccess) { set_err("profile: %s", hipGetErrorString(e)); return (int)e; }
    ms += t; ++n; bytes += r.bytes;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (alg_bytes) *alg_bytes = bytes;
  return LLIE_OK;
}

int llie_profile_report(llie_ctx* c, char* buf, size_t cap) {
  if (!c || !buf || cap < 2) return LLIE_ERR_ARG;
  c->prof_mask = 0;
  struct Agg { double ms = 0; int64_t n = 0, bytes = 0; };
  std::map<std::string, Agg> agg;
  for (auto& r : c->prof) {
    hipError_t e = hipEventSynchronize(r.e1);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, r.e0, r.e1);
    if (e != hipSuccess) { set_err("profile: %s", hipGetErrorString(e)); return (int)e; }
    Agg& a = agg[r.name ? r.name : "?"];
    a.ms += t; a.n += 1; a.bytes += r.bytes;
  }
  std::string out;
  char line[512];
  for (auto& kv : agg) {
    snprintf(line, sizeof line, "%s\t%.6f\t%lld\t%lld\n", kv.first.c_str(), kv.second.ms, (long long)kv.second.n, (long long)kv.second.bytes);
    out += line;
  }
  if (out.size() + 1 > cap) { set_err("profile report buffer too small"); return LLIE_ERR_ARG; }
  memcpy(buf, out.c_str(), out.size() + 1);
  return LLIE_OK;
}

// SURVEY.md 8d byte model: IRB (2Cin + 4Chid + Cout)P, attention 6CP, dense 3x3 Cin*Pin + Cout*Pout,
// final C0*P + 3P, LCM step 12P fp32; activations at the compute dtype; weights once.
static void count_blocks(const llie_ctx* c, const std::vector<Block>& bl, int64_t P, int64_t& elems, int64_t& flops) {
  for (const Block& b : bl) {
    if (b.kind == 0) {
      const IrbW& w = c->irbs[b.idx];
      elems += (2LL * w.cin + 4LL * w.hid + w.cout) * P;
      flops += 2LL * P * ((int64_t)w.cin * w.hid + 9LL * w.hid + (int64_t)w.hid * w.cout + (w.skip ? (int64_t)w.cin * w.cout : 0));
    } else {
      const AttnW& w = c->attns[b.idx];
      elems += 6LL * w.c * P;
      flops += 2LL * P * ((int64_t)w.c * 3 * w.inner + (int64_t)w.inner * w.c + 2LL * w.inner * 32);
    }
  }
}
static void model_counts(const llie_ctx* c, int64_t& elems, int64_t& flops) {
  elems = flops = 0;
  if (c->cfg.kind != LLIE_UNET) return;
  const int S = c->cfg.image_size;
  int64_t P = (int64_t)S * S;
  const std::vector<int>& ch = c->channels;
  elems += (int64_t)c->cfg.in_channels * P + ch[0] * P;
  flops += 2LL * P * 9 * c->cfg.in_channels * ch[0];
  for (int l = 0; l < 4; ++l) {
    count_blocks(c, c->enc[l], P, elems, flops);
    if (l < 3) {
      elems += ch[l] * P + ch[l] * (P / 4);
      flops += 2LL * (P / 4) * 9 * ch[l] * ch[l];
      P /= 4;
    }
  }
  count_blocks(c, c->mid, P, elems, flops);
  for (int l = 0; l < 4; ++l) {
    if (l > 0) {
      const int cc = ch[4 - l];
      elems += (int64_t)cc * P + (int64_t)cc * P * 4;
      flops += 2LL * (P * 4) * 9 * cc * cc;
      P *= 4;
    }
    count_blocks(c, c->dec[l], P, elems, flops);
  }
  elems += (int64_t)ch[0] * P + 3 * P;
  flops += 2LL * P * 9 * ch[0] * c->cfg.out_channels;
}

int64_t llie_algorithmic_bytes(llie_ctx* c, int batch) {
  if (!c) return LLIE_ERR_ARG;
  int64_t elems, flops;
  model_counts(c, elems, flops);
  int64_t wbytes = 0;
  for (const Param& p : c->params) wbytes += p.numel * (p.kind == PK_F32 || !p.as_t ? 4 : (int64_t)elem_size(c->dt));
  return elems * batch * (int64_t)elem_size(c->dt) + wbytes;
}
int64_t llie_flops(llie_ctx* c, int batch) {
  if (!c) return LLIE_ERR_ARG;
  int64_t elems, flops;
  model_counts(c, elems, flops);
  return flops * batch;
}

}  // extern "C"
