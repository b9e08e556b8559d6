// libllie_hip.so: context, state_dict repack, workspace arena and the launch sequence of the
// denoiser behind the C ABI of include/llie.h.  Host-only code; kernels live in the *.hip files.
//
// Execution model: one UNet forward is a fixed sequence of kernel launches on the caller's stream.
// All temporaries come from a caller-provided workspace through a deterministic first-fit arena, so
// the same (batch, H, W) always produces the same offsets: llie_workspace_bytes() replays the
// sequence with launches disabled to obtain the high-water mark.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
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
  int Op = 0, Ip = 0;                  // padded destination dims of conv / depthwise layouts (0 = O / I)
  bool as_t = true;                    // PK_MAT: store as compute dtype (true) or fp32
  bool loaded = false;
  int ndim = 1;
  int64_t shape[4] = {0, 0, 0, 0};     // shape in the reference's state_dict
  // training: second copy in the layout the input-gradient kernels read (PK_MAT: transposed [cols][rows];
  // PK_DW: taps flipped; PK_CONV3: [8-tap][I][O]); 0 = none.  goff = offset (floats) in the flat gradient buffer.
  bool has_t = false;
  size_t t_off = 0;
  int64_t goff = 0;
  // PK_MAT, 2-byte engines: third copy = the matrix times f_scale in MFMA fragment order (pwx.hip: launch_pack_expand)
  bool has_f = false;
  size_t f_off = 0;
  float f_scale = 1.f;
};

// cin / cout / hid are the PHYSICAL channel counts of the tensors (multiples of 32; hid of 64 for 2-byte types);
// *_r the reference's.  They differ only for the unpinned variants (tiny / base), whose odd channel counts are
// zero-padded at the end of each tensor: zero weights and a zero norm affine keep the padding at exactly zero.
struct IrbW {
  int cin, cout, hid, sq;
  int cin_r, cout_r, hid_r;
  bool skip;
  size_t n1g, n1b, n2g, n2b, w_expand, w_dw, se_w1, se_b1, se_w2, se_b2, w_proj;
  int film_off;  // first row of this block inside the concatenated FiLM projection
  size_t w_expand_t, w_proj_t, w_dw_flip;  // training copies: [cin][hid], [hid (+cin)][cout], flipped taps
  bool has_wf = false;  // expand weights x 6 in MFMA fragment order for the activation-stationary kernel (pwx.hip)
  size_t w_expand_f = 0;
  int p_first;   // index of this block's first parameter (norm1.weight); the rest follow in registration order
};
struct AttnW {
  int c, heads, inner;
  size_t ng, nb, w_qkv, w_out, n2g, n2b;
  size_t w_qkv_t, w_out_t;
  int p_first;
};
struct ConvW {
  int c, c_r;
  size_t w, bias;
  size_t w_t;
  int p_first;
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
  int Cr = 0;  // real channels (<= C; the rest is zero padding)
};

// GroupNorm(min(32, C), C) of the reference; for channel counts it cannot construct (tiny / base: 48, 144 ...) the
// documented deviation: the largest divisor of C that is <= 32.  Identical whenever C is a multiple of 32 or C < 32 | 32.
inline int gn_groups(int c) {
  for (int g = std::min(32, c); g >= 1; --g)
    if (c % g == 0) return g;
  return 1;
}
inline int pad32(int c) { return (c + 31) / 32 * 32; }

// ---------------------------------------------------------------------------------------------
// Training tape: what the forward pass leaves in the workspace for the backward pass (offsets).
struct GnRec { size_t as = 0, ab = 0, mean = 0, rstd = 0; };
struct IrbRec { int w; Tens x0, x1; bool cat; GnRec n1, n2; Tens h1; size_t h2, gate, sehid, semean; Tens y; };
struct AttnRec { int w; Tens x; GnRec n1, n2; size_t qkv, kv, ao; int nsplit; Tens tmp, y; };
struct ConvRec { int w; bool up; Tens x, u, y; };
struct TapeOp { int kind, idx; };  // kind: 0 irb, 1 attn, 2 conv; idx into the vectors below
struct Tape {
  std::vector<IrbRec> irbs;
  std::vector<AttnRec> attns;
  std::vector<ConvRec> convs;
  std::vector<TapeOp> ops;  // forward order
  // UNet level
  size_t temb = 0, stemb = 0, film = 0;
  Tens h0, hlast;
  GnRec fin;
  const float* lat = nullptr; const float* cond = nullptr; const int64_t* t = nullptr;
  int B = 0;
  const void* ws = nullptr;
  bool valid = false;
  void clear() { irbs.clear(); attns.clear(); convs.clear(); ops.clear(); valid = false; }
};

}  // namespace

constexpr int kMaxBranches = 8;
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
  std::vector<int> channels;    // physical (padded) channels per level
  std::vector<int> channels_r;  // the reference's
  bool padded = false;          // some tensor carries zero padding (unpinned variant): inference only
  // UNet-level tensors
  size_t t_w1 = 0, t_b1 = 0, t_w3 = 0, t_b3 = 0, freqs = 0, film_w = 0, film_b = 0;
  int film_rows = 0;
  int64_t grad_numel = 0;
  // batched reload (llie_load_all): device descriptor table + the host pointers it was built for
  LoadDesc* load_descs = nullptr;
  std::vector<const float*> load_srcs;
  unsigned long long* hash_partial = nullptr;  // [n][32] partial content hashes (llie_refresh_params)
  unsigned long long* hash_state = nullptr;    // [0] hash of the last load, [1] "changed" flag read by load_all_kernel
  Tape tape;           // last training forward (llie_unet_train_forward), read by llie_unet_backward
  Arena* train_arena = nullptr;  // arena state after that forward; the backward pass continues in it
  size_t init_wp = 0, fin_wp = 0;  // MFMA-packed init / final conv weights (2-byte compute dtypes)
  size_t init_w = 0, init_b = 0, fin_g = 0, fin_b = 0, fin_w = 0, fin_bias = 0;
  // hipGraph cache of llie_enhance launch sequences (key -> executable graph)
  // bounded: least-recently-used entries beyond kMaxGraphs are destroyed (a server sweeping batch sizes or schedules would
  // otherwise grow it without limit; an evicted key is simply captured again on its second next use)
  struct GraphEntry { bool seen = false; hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; uint64_t used = 0; };
  static constexpr size_t kMaxGraphs = 16;
  uint64_t graph_clock = 0;
  std::map<std::string, GraphEntry> graphs;
  std::map<std::tuple<int, int64_t, int>, size_t> zneed;  // (batch, pixels, knob epoch) -> bytes of zero-initialised totals one forward takes (Run::zbegin)
  hipStream_t cap_stream = nullptr;  // side stream used only to record captures (the legacy null stream cannot capture)
  // backward pass: weight-gradient kernels run on this stream next to the activation-gradient chain (Back::fork/join)
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // concurrent branches of the captured enhance graph (branch 0 is cap_stream)
  hipStream_t branch_stream[kMaxBranches] = {};
  hipEvent_t branch_join[kMaxBranches] = {};
  // per-kernel-class HIP-event profiling (llie_profile_begin / llie_profile_end)
  int prof_mask = 0;
  struct ProfRec { int cls; int64_t bytes; hipEvent_t e0, e1; const char* name; char tag[56]; };
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
  size_t f32(const std::string& key, int64_t n, int64_t n_phys = 0) {  // n_phys: zero-padded length of the destination
    const size_t o = reserve((size_t)std::max(n, n_phys) * 4);
    add(key, n, PK_F32, o);
    return o;
  }
  int padh(int hid) const { return c->dt == LLIE_F32 ? pad32(hid) : (hid + 63) / 64 * 64; }  // depthwise: 64 channels per workgroup
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
  size_t mat(const std::string& key, int rows, int cols, size_t* t_off = nullptr, int rows_p = 0, int cols_p = 0) {
    rows_p = std::max(rows, rows_p); cols_p = std::max(cols, cols_p);
    const size_t o = reserve((size_t)rows_p * cols_p * es());
    mat_into(key, rows, cols, o, cols_p, 0);
    if (t_off) {
      *t_off = reserve((size_t)rows_p * cols_p * es());
      c->params.back().has_t = true;
      c->params.back().t_off = *t_off;
    }
    return o;
  }
  // cin_r / cout_r: the reference's channel counts; x0_r: real channels of the first input segment (== cin_r unless
  // the block reads a virtual concat, whose first segment must be unpadded so that real channels stay contiguous)
  int add_irb(const std::string& p, int cin_r, int cout_r, int T, int e, int x0_r = 0) {
    if (x0_r <= 0) x0_r = cin_r;
    if (x0_r != cin_r && x0_r % 32) bad = true;  // a padded first concat segment would break the real-channel numbering
    IrbW w{};
    w.cin_r = cin_r; w.cout_r = cout_r; w.hid_r = cin_r * e; w.sq = std::max(1, (int)(w.hid_r * 0.25));
    w.cin = x0_r == cin_r ? pad32(cin_r) : x0_r + pad32(cin_r - x0_r);
    w.cout = pad32(cout_r); w.hid = padh(w.hid_r);
    if (w.cin != cin_r || w.cout != cout_r || w.hid != w.hid_r) c->padded = true;
    w.skip = cin_r != cout_r;
    w.p_first = (int)c->params.size();
    w.n1g = f32(p + ".norm1.weight", cin_r, w.cin); w.n1b = f32(p + ".norm1.bias", cin_r, w.cin);
    w.n2g = f32(p + ".norm2.weight", w.hid_r, w.hid); w.n2b = f32(p + ".norm2.bias", w.hid_r, w.hid);
    w.w_expand = mat(p + ".expand.weight", w.hid_r, cin_r, &w.w_expand_t, w.hid, w.cin);
    if (c->dt != LLIE_F32 && w.hid == w.hid_r && w.cin == cin_r && pw_expand_serves_k(w.cin) && w.hid % 64 == 0) {
      // wide blocks of the 2-byte engines: the expand GEMM runs activation-stationary (pwx.hip) from this packed copy
      w.has_wf = true;
      w.w_expand_f = reserve((size_t)w.hid * w.cin * es());
      Param& q = c->params.back();
      q.has_f = true; q.f_off = w.w_expand_f; q.f_scale = 6.f;  // the 6 of ReLU6 carried as clamp01(z / 6), kernels.h
    }
    w.w_dw = reserve((size_t)9 * w.hid * 4);
    w.w_dw_flip = reserve((size_t)9 * w.hid * 4);
    { Param& q = add(p + ".depthwise.weight", (int64_t)w.hid_r * 9, PK_DW, w.w_dw); q.O = w.hid_r; q.Op = w.hid; set_shape(q, {w.hid_r, 1, 3, 3});
      q.has_t = true; q.t_off = w.w_dw_flip; }
    w.se_w1 = mat(p + ".se.fc1.weight", w.sq, w.hid_r, nullptr, w.sq, w.hid); w.se_b1 = f32(p + ".se.fc1.bias", w.sq);
    w.se_w2 = mat(p + ".se.fc2.weight", w.hid_r, w.sq, nullptr, w.hid, w.sq); w.se_b2 = f32(p + ".se.fc2.bias", w.hid_r, w.hid);
    const int kp = w.hid + (w.skip ? w.cin : 0);  // project and skip share one K-concatenated matrix
    w.w_proj = reserve((size_t)w.cout * kp * es());
    w.w_proj_t = reserve((size_t)w.cout * kp * es());  // [kp][cout]: project rows first, then the skip rows
    mat_into(p + ".project.weight", cout_r, w.hid_r, w.w_proj, kp, 0);
    c->params.back().has_t = true; c->params.back().t_off = w.w_proj_t;
    // FiLM Linear: rows appended to the global [F][T] fp32 table (filled in finish())
    w.film_off = c->film_rows;
    c->film_rows += 2 * w.hid_r;
    film_keys.push_back({p + ".time_mlp.1", 2 * w.hid_r, w.film_off});
    if (w.skip) pending_skip.push_back({p + ".skip.weight", cout_r, cin_r, w.w_proj, kp, w.hid, w.w_proj_t + (size_t)w.hid * w.cout * es()});
    flush_pending();  // registration order: ... project, time_mlp, skip
    c->irbs.push_back(w);
    return (int)c->irbs.size() - 1;
  }
  bool bad = false;
  struct FilmKey { std::string p; int rows, off; };
  struct SkipKey { std::string key; int rows, cols; size_t off; int ld, col0; size_t t_off; };
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
    for (auto& s : pending_skip) {
      mat_into(s.key, s.rows, s.cols, s.off, s.ld, s.col0);
      c->params.back().has_t = true; c->params.back().t_off = s.t_off;
    }
    pending_skip.clear();
  }
  int add_attn(const std::string& p, int ch, int heads) {
    if (ch % 32) bad = true;
    AttnW w{};
    w.c = ch; w.heads = heads; w.inner = heads * 32;
    w.p_first = (int)c->params.size();
    w.ng = f32(p + ".norm.weight", ch); w.nb = f32(p + ".norm.bias", ch);
    w.w_qkv = mat(p + ".to_qkv.weight", 3 * w.inner, ch, &w.w_qkv_t);
    w.w_out = mat(p + ".to_out.0.weight", ch, w.inner, &w.w_out_t);
    w.n2g = f32(p + ".to_out.1.weight", ch); w.n2b = f32(p + ".to_out.1.bias", ch);
    c->attns.push_back(w);
    return (int)c->attns.size() - 1;
  }
  ConvW add_conv3(const std::string& p, int ch_r) {
    ConvW w{};
    const int ch = pad32(ch_r);
    if (ch != ch_r) c->padded = true;
    w.c = ch; w.c_r = ch_r;
    w.p_first = (int)c->params.size();
    w.w = reserve((size_t)9 * ch * ch * es());
    w.w_t = reserve((size_t)9 * ch * ch * es());
    { Param& q = add(p + ".weight", (int64_t)ch_r * ch_r * 9, PK_CONV3, w.w); q.O = ch_r; q.I = ch_r; q.Op = ch; q.Ip = ch;
      set_shape(q, {ch_r, ch_r, 3, 3}); q.has_t = true; q.t_off = w.w_t; }
    w.bias = f32(p + ".bias", ch_r, ch);
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

void assign_grad_offsets(llie_ctx* c) {
  int64_t o = 0;
  for (Param& p : c->params) { p.goff = o; o += p.numel; }
  c->grad_numel = o;
}

int build_unet(llie_ctx* c) {
  const llie_config& g = c->cfg;
  Builder b{c};
  c->channels.clear();
  c->channels_r.clear();
  for (int i = 0; i < 4; ++i) {
    c->channels_r.push_back(g.base_channels * g.channel_multipliers[i]);
    c->channels.push_back(pad32(c->channels_r.back()));
  }
  const std::vector<int>& ch = c->channels_r;  // the builder registers parameters with the reference's shapes
  const int T = g.time_embed_dim, e = g.expansion_ratio;
  // GroupNorm(min(32,C), C) must be constructible for every site (efficient_unet.py:170-171,263,528) -- unless the
  // caller opted into the unpinned variants (allow_unpinned: groups = largest divisor <= 32, channels zero-padded)
  auto gn_ok = [](int x) { return x >= 32 && x % 32 == 0; };
  if (g.allow_unpinned) {
    // what the padding scheme needs: the first segment of every virtual concat and every attention input unpadded
    for (int l = 1; l < 4; ++l)
      if (ch[l] % 32) return LLIE_ERR_CONFIG;
    if (g.base_channels < 8 || g.base_channels % 8) return LLIE_ERR_CONFIG;
  } else {
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
  // three stride-2 levels, each followed by a x2 upsample that must restore the size: any multiple of 8, like the reference
  // (edge tiles of the kernels may be partly empty).  Training needs multiples of 64 (checked in the training entry points).
  if (g.base_channels % 2 || g.image_size % 8 || g.image_size < 64) return LLIE_ERR_SHAPE;
  if (g.in_channels < 2 || g.in_channels > 8 || g.out_channels > 4) return LLIE_ERR_SHAPE;

  c->t_w1 = b.reserve((size_t)T * g.base_channels * 4);
  { Param& p = b.add("time_mlp.1.weight", (int64_t)T * g.base_channels, PK_F32, c->t_w1); Builder::set_shape(p, {T, g.base_channels}); }
  c->t_b1 = b.f32("time_mlp.1.bias", T);
  c->t_w3 = b.f32("time_mlp.3.weight", (int64_t)T * T);
  Builder::set_shape(c->params.back(), {T, T});
  c->t_b3 = b.f32("time_mlp.3.bias", T);
  const int c0p = c->channels[0];
  if (c0p != ch[0]) c->padded = true;
  c->init_w = b.reserve((size_t)c0p * g.in_channels * 9 * 4);
  { Param& p = b.add("init_conv.weight", (int64_t)ch[0] * g.in_channels * 9, PK_INIT, c->init_w); p.O = ch[0]; p.I = g.in_channels; p.Op = c0p; Builder::set_shape(p, {ch[0], g.in_channels, 3, 3}); }
  c->init_b = b.f32("init_conv.bias", ch[0], c0p);
  c->init_wp = b.reserve((size_t)10 * c0p * 8 * 2);

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
      c->dec[l].push_back({0, b.add_irb(p, r == 0 ? in_ch + out : out, out, T, e, r == 0 ? in_ch : 0)});
      if (is_attn_res(res)) {
        const std::string pa = "decoder_blocks." + std::to_string(l) + "." + std::to_string(k++);
        c->dec[l].push_back({1, b.add_attn(pa, out, g.num_attention_heads)});
      }
    }
    in_ch = out;
    if (l < 3) res *= 2;
  }
  for (int l = 0; l < 3; ++l) c->ups.push_back(b.add_conv3("upsamplers." + std::to_string(l) + ".conv", ch[3 - l]));
  c->fin_g = b.f32("final_norm.weight", ch[0], c0p);
  c->fin_b = b.f32("final_norm.bias", ch[0], c0p);
  c->fin_w = b.reserve((size_t)9 * c0p * 4 * 4);
  { Param& p = b.add("final_conv.weight", (int64_t)g.out_channels * ch[0] * 9, PK_FINAL, c->fin_w); p.O = g.out_channels; p.I = ch[0]; p.Ip = c0p; Builder::set_shape(p, {g.out_channels, ch[0], 3, 3}); }
  c->fin_bias = b.f32("final_conv.bias", g.out_channels);
  c->fin_wp = b.reserve((size_t)(c0p / 32) * 18 * 2 * 4 * 8 * 2);
  c->freqs = b.reserve((size_t)(g.base_channels / 2) * 4);
  b.finish_film();
  c->blob_bytes = b.cursor;
  assign_grad_offsets(c);
  if (b.bad) return LLIE_ERR_CONFIG;
  return LLIE_OK;
}

int build_module(llie_ctx* c) {
  const llie_config& g = c->cfg;
  Builder b{c};
  auto gn_ok = [](int x) { return x >= 32 && x % 32 == 0; };
  switch (g.kind) {
    case LLIE_IRB:
      if (!gn_ok(g.in_channels) || !gn_ok(g.in_channels * g.expansion_ratio) || g.out_channels % 32) return LLIE_ERR_CONFIG;
      // a two-tensor (virtual concat) input has no single tensor to add as the identity residual: like every
      // concat-fed block of the network (efficient_unet.py:588), such a block needs Cin != Cout (skip conv)
      if (g.base_channels > 0 && (g.in_channels == g.out_channels || g.base_channels % 32 || g.base_channels >= g.in_channels))
        return LLIE_ERR_SHAPE;
      b.add_irb("", g.in_channels, g.out_channels, g.time_embed_dim, g.expansion_ratio, g.base_channels);
      // keys of a bare block have no leading dot
      break;
    case LLIE_ATTN:
      if (!gn_ok(g.in_channels)) return LLIE_ERR_CONFIG;
      b.add_attn("", g.in_channels, g.num_attention_heads);
      break;
    case LLIE_SE: {  // efficient_unet.py:85-94: squeezed = max(1, int(C * 0.25)), both 1x1 convs with bias
      if (g.in_channels % 32) return LLIE_ERR_CONFIG;
      IrbW w{};
      w.hid = w.hid_r = g.in_channels; w.sq = std::max(1, (int)(g.in_channels * 0.25));
      w.se_w1 = b.mat("fc1.weight", w.sq, w.hid); w.se_b1 = b.f32("fc1.bias", w.sq);
      w.se_w2 = b.mat("fc2.weight", w.hid, w.sq); w.se_b2 = b.f32("fc2.bias", w.hid);
      c->irbs.push_back(w);
      break;
    }
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
  assign_grad_offsets(c);
  return LLIE_OK;
}

// ---------------------------------------------------------------------------------------------
// Run helpers.  In dry mode nothing is launched; only the arena is exercised.
// Recompute form (irbx.hip: statistics-only expand + tile-fused expand/depthwise): default for the
// inference path of 2-byte engines wherever irbx_supported(); llie_tune("irbx", 0) restores the unfused pair.
int g_use_irbx = getenv("LLIE_NO_IRBX") ? 0 : 1;
int g_gram = 1;  // norm2 statistics of the recompute form from the Gram matrix of the block input (gram.hip); 0 = expand_stats
int g_ztot = 1;  // SE pool as fixed-point totals + fused gate kernel (llie_tune("ztot", 0): the slab + three launches, as in training)
// Cache policy of the big activation tensors (inference): a tensor of at least g_nt_min_mb MiB (this run's batch) is stored
// non-temporally by its producer (common.h: st_vec_pol).  g_nt_mask picks the producers: 1 expand_dw (h2), 2 pw_expand (h1),
// 4 dwconv3x3 (h2), 8 project / attention GEMM outputs, 16 dense 3x3 conv outputs.  Values never change, only where lines live.
int g_nt_min_mb = 100, g_nt_mask = 1;
int g_se_mfma = 1;  // SE MLP of the wide blocks as two MFMA launches (small.hip: se_fc1_mfma / se_fc2_mfma); 0 = the row-parallel pair
int g_skip_small = 0;  // timing ablation only (results are garbage): bit 0 no gn_finalize launches, bit 1 no SE launches
int g_irbx_mask = 0x7;  // debug: which input widths may take the recompute form (bit 0: 32, bit 1: 64, bit 2: 96 channels)
// Backward pass: run the weight-gradient kernels on a side stream next to the activation-gradient chain
// (llie_tune("bwd_async", 0) puts everything back on the caller's stream).
int g_bwd_async = 1;
// hipGraph path of llie_enhance: batches of 16 and more are captured as two concurrent half-batch branches (no operator
// mixes samples and every kernel is bitwise batch-invariant, so the bits do not change): the launch-bound tail of one
// half (norm finalisation, SE MLP) overlaps the streaming kernels of the other, +4 % at B = 32.  Overlapping kernels
// stretch each other, so per-kernel durations are only meaningful from a single chain: llie_profile_* already forces
// the eager single chain, and LLIE_ENHANCE_SPLIT=0 (or llie_tune("enhance_split", 0)) gives rocprofv3 the same.
int g_enhance_split = getenv("LLIE_ENHANCE_SPLIT") ? atoi(getenv("LLIE_ENHANCE_SPLIT")) : 2;
// Captured graphs bake in the kernel choices of the moment: every llie_tune call starts a new epoch of the graph cache.
int g_tune_epoch = 0;
int tune_epoch() { return g_tune_epoch; }

struct Run {
  llie_ctx* c;
  Arena* ar;
  hipStream_t s;
  char* ws;
  bool dry;
  int B;
  int dt;
  hipError_t err = hipSuccess;
  Tape* tape = nullptr;  // non-null: training forward -- nothing is released, every operator is recorded
  char tag[56] = "";     // label of the operator being launched (llie_profile_dump)
  void rel(size_t off) { if (!tape) ar->free(off); }
  // Zero-initialised totals (inference): fixed-point accumulators that kernels add to with integer atomics (SE pool sums).
  // One block of the arena per forward, cleared by a single memset node at its start and handed out by ztake() in
  // launch order; its size comes from a counting dry run of the same forward (cached per batch and image size).
  size_t zoff = 0, zcur = 0, zcap = 0;
  bool zcount = false;
  template <typename F> void zbegin(int64_t pixels, F&& forward_again) {
    if (tape || zcount) return;
    const auto key = std::make_tuple(B, pixels, tune_epoch());
    auto it = c->zneed.find(key);
    if (it == c->zneed.end()) {
      Arena probe((size_t)1 << 46);
      Run d{c, &probe, nullptr, nullptr, true, B, dt};
      d.zcount = true;
      forward_again(d);
      it = c->zneed.emplace(key, d.zcur).first;
    }
    zcap = it->second;
    if (!zcap) return;
    zoff = ar->alloc(zcap);
    // a kernel of ours, not hipMemsetAsync: as a memset node of the captured graph it stopped clearing the region once another
    // engine context had run between two replays (ROCm 7.2; tests/test_gpu_round2.py::test_inplace_data_writes_are_noticed)
    if (!dry) chk(launch_zero_fill(ws + zoff, (int64_t)zcap, s));
  }
  size_t ztake(size_t bytes) {
    bytes = align_up(bytes, 256);
    const size_t off = zoff + zcur;
    zcur += bytes;
    if (!zcount && zcur > zcap) chk(hipErrorOutOfMemory);
    return off;
  }

  template <typename T = void> T* wptr(size_t off) const { return reinterpret_cast<T*>(c->blob + off); }
  template <typename T = void> T* p(size_t off) const { return reinterpret_cast<T*>(ws + off); }
  void chk(hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; }
  int nt_store(int site, int64_t elems) const {  // see g_nt_min_mb
    return !tape && (g_nt_mask & site) && g_nt_min_mb > 0 && elems * (int64_t)elem_size(dt) >= ((int64_t)g_nt_min_mb << 20) ? 1 : 0;
  }
  size_t es() const { return elem_size(dt); }
  // launch `f` bracketed by HIP events on the launch stream when its class is being profiled
  template <typename F> void timed(int cls, int64_t bytes, F&& f, const char* nm = nullptr) {
    if (!(c->prof_mask & cls) || c->prof.size() >= 8192) { chk(f()); return; }
    llie_ctx::ProfRec r{cls, bytes, c->get_event(), c->get_event(), "", {0}};
    snprintf(r.tag, sizeof r.tag, "%s", tag);
    if (!r.e0 || !r.e1) { chk(f()); return; }
    chk(hipEventRecord(r.e0, s));
    chk(f());
    chk(hipEventRecord(r.e1, s));
    r.name = nm ? nm : last_kernel();  // static storage: launchers pass string literals / function-local statics
    c->prof.push_back(r);
  }

  Tens new_tens(int C, int H, int W, int ntiles, int Cr = 0) {
    Tens t;
    t.C = C; t.H = H; t.W = W; t.ntiles = ntiles; t.valid = true;
    t.Cr = Cr > 0 ? Cr : C;
    t.off = ar->alloc((size_t)B * H * W * C * es());
    t.slab = ar->alloc((size_t)B * ntiles * 2 * C * 4);
    return t;
  }
  void free_tens(Tens& t) {
    if (!t.valid || tape) return;
    ar->free(t.off);
    ar->free(t.slab);
    t.valid = false;
  }
  StatSrc src(const Tens& t) const { return StatSrc{p<float>(t.slab), t.ntiles, t.C}; }

  // GroupNorm affine of (x0 [+ x1]) -> freshly allocated as/ab [B][C]; returns offsets
  void gn(const Tens& x0, const Tens* x1, size_t gamma, size_t beta, const float* film, int64_t film_stride,
          size_t& as, size_t& ab, GnRec* rec = nullptr, float post_scale = 0.f) {
    const int C = x0.C + (x1 ? x1->C : 0);
    const int Creal = x0.Cr + (x1 ? x1->Cr : 0);  // x0 is unpadded whenever x1 exists (checked at build time)
    as = ar->alloc((size_t)B * C * 4);
    ab = ar->alloc((size_t)B * C * 4);
    size_t mo = 0, ro = 0;
    if (tape && rec) {
      mo = ar->alloc((size_t)B * 32 * 4);
      ro = ar->alloc((size_t)B * 32 * 4);
      rec->as = as; rec->ab = ab; rec->mean = mo; rec->rstd = ro;
    }
    if (dry) return;
    GnFinalizeArgs a{};
    a.src[0] = src(x0);
    if (x1) a.src[1] = src(*x1);
    a.C = C; a.Creal = Creal; a.groups = gn_groups(Creal); a.P = x0.H * x0.W;
    a.gamma = wptr<float>(gamma); a.beta = wptr<float>(beta);
    a.film = film; a.film_stride = film_stride; a.eps = 1e-5f;
    a.as = p<float>(as); a.ab = p<float>(ab); a.B = B; a.post_scale = post_scale;
    if (tape && rec) { a.mean_out = p<float>(mo); a.rstd_out = p<float>(ro); }
    if (g_skip_small & 1) return;
    if ((g_skip_small & 4) && film && a.P <= 4096) return;          // bound on a producer-tail finalize behind pw_expand (norm2 + FiLM)
    if ((g_skip_small & 8) && !film && !x1 && a.P <= 4096) return;  // ... behind project GEMMs / convs (norm1, single source)
    if ((g_skip_small & 16) && !film && !x1 && a.P >= 16384) return; // the single-source norm1 finalizes of the high-resolution levels
    if ((g_skip_small & 32) && (film || x1) && a.P >= 16384) return; // the other high-resolution ones (norm2 + FiLM behind expand_stats / pw_expand, concat inputs)
    timed(LLIE_K_OTHER, (int64_t)B * C * 8, [&] { return launch_gn_finalize(a, s); }, "gn_finalize_kernel");
  }

  // InvertedResidualBlock.forward (efficient_unet.py:203-236) as 7 launches.
  Tens irb(const IrbW& w, const Tens& x0, const Tens* x1, const float* film, int64_t film_stride) {
    const int H = x0.H, W = x0.W, P = H * W, M = B * P;
    const int BM = pw_gemm_tile_rows(P);
    size_t as1, ab1;
    IrbRec rec{};
    snprintf(tag, sizeof tag, "irb P=%d %d->%d hid=%d", P, w.cin, w.cout, w.hid);
    // 2-byte inference engines carry norm1's ReLU6 as clamp01(z / 6): the tables come out divided by 6 and the expand
    // GEMM (or the recompute kernels) puts the 6 back (kernels.h: ACT_RELU6_S6).  Training keeps the plain tables.
    const bool s6 = !tape && dt != LLIE_F32;
    gn(x0, x1, w.n1g, w.n1b, nullptr, 0, as1, ab1, &rec.n1, s6 ? 1.f / 6.f : 0.f);
    // Recompute form (2-byte T, narrow inputs): a statistics-only expand pass, then the fused expand + depthwise kernel
    // rebuilds h1 on the fly, so the 4x-expanded tensor never touches HBM (irbx.hip).
    const bool fusedx = !tape && g_use_irbx && w.hid == w.hid_r && w.cin == w.cin_r && ((g_irbx_mask >> (w.cin / 32 - 1)) & 1) &&
                        irbx_supported(dt, w.cin, x0.C, w.hid, H, W);
    // K1: expand with norm1 + ReLU6 prologue
    Tens h1;
    h1.C = w.hid; h1.Cr = w.hid_r; h1.H = H; h1.W = W; h1.ntiles = fusedx ? P / irbx_stats_rows(P) : pw_gemm_ntiles(P); h1.valid = true;
    h1.off = fusedx ? 0 : ar->alloc((size_t)B * P * w.hid * es());
    // norm2's statistics of the recompute form: from the Gram matrix of the activated input (gram.hip) -- the statistics
    // pass is then a plain read of x -- or, knob "gram" = 0, from a second run of the expand GEMM (expand_stats)
    // (from 32 768 pixels per image on: below, the workgroup epilogue and the last-ticket sum outweigh the saved MFMAs --
    // measured at B = 1 and B = 32; the rule must not depend on the batch, it fixes the statistics' summation order)
    const bool gram = fusedx && g_gram && (P >= 32768 || g_gram > 1) && gram_supported(dt, w.cin, x0.C, P);
    h1.slab = gram ? 0 : ar->alloc((size_t)B * h1.ntiles * 2 * w.hid * 4);
    const size_t gpart = gram ? ar->alloc((size_t)B * gram_part_floats(w.cin, P) * 4) : 0;
    const size_t gtot = gram ? ar->alloc((size_t)B * (w.cin * w.cin + w.cin) * 4) : 0;
    const size_t gtick = gram ? ztake((size_t)B * 4) : 0;
    IrbxArgs xa{};
    if (fusedx && !dry) {
      xa.x0 = p(x0.off); xa.c0 = x0.C; xa.x1 = x1 ? p(x1->off) : nullptr; xa.c1 = x1 ? x1->C : 0;
      xa.as1 = p<float>(as1); xa.ab1 = p<float>(ab1); xa.w1 = wptr(w.w_expand); xa.wd = wptr<float>(w.w_dw);
      xa.stats = gram ? nullptr : p<float>(h1.slab); xa.B = B; xa.H = H; xa.W = W; xa.Chid = w.hid;
      if (gram) {
        GramArgs ga{};
        ga.x0 = xa.x0; ga.x1 = xa.x1; ga.c0 = xa.c0; ga.c1 = xa.c1; ga.as1 = xa.as1; ga.ab1 = xa.ab1;
        ga.part = p<float>(gpart); ga.gtot = p<float>(gtot); ga.tickets = p<unsigned int>(gtick); ga.B = B; ga.P = P;
        timed(LLIE_K_GEMM, (int64_t)M * w.cin * (int64_t)es(), [&] { return launch_gram_stats(dt, ga, s); });
      } else {
        timed(LLIE_K_GEMM, ((int64_t)M * w.cin + (int64_t)w.hid * w.cin) * (int64_t)es(), [&] { return launch_expand_stats(dt, xa, s); });
      }
    } else if (!dry) {
      GemmArgs g{};
      const int act1 = s6 ? ACT_RELU6_S6 : ACT_RELU6;
      g.seg[0] = GemmSeg{p(x0.off), x0.C, p<float>(as1), p<float>(ab1), w.cin, act1};
      g.nseg = 1;
      if (x1) {
        g.seg[1] = GemmSeg{p(x1->off), x1->C, p<float>(as1) + x0.C, p<float>(ab1) + x0.C, w.cin, act1};
        g.nseg = 2;
      }
      g.w = wptr(w.w_expand); g.out = p(h1.off); g.stats = p<float>(h1.slab);
      g.M = M; g.N = w.hid; g.K = w.cin; g.P = P;
      const int64_t kbytes = ((int64_t)M * (w.cin + w.hid) + (int64_t)w.hid * w.cin) * (int64_t)es();
      if (s6 && w.has_wf && pw_expand_supported(dt, g.seg, g.nseg, M, w.hid, w.cin, P)) {
        // activation-stationary form: pixels activated once and held in registers, packed weights streamed (pwx.hip)
        ExpandArgs x{};
        for (int i = 0; i < g.nseg; ++i) x.seg[i] = g.seg[i];
        x.nseg = g.nseg; x.wf = wptr(w.w_expand_f); x.out = g.out; x.stats = g.stats;
        x.M = M; x.N = w.hid; x.K = w.cin; x.P = P;
        x.nt = nt_store(2, (int64_t)M * w.hid);
        timed(LLIE_K_GEMM, kbytes, [&] { return launch_pw_expand(dt, x, s); });
      } else if (s6) {
        g.nt = nt_store(2, (int64_t)M * w.hid);
        timed(LLIE_K_GEMM, kbytes, [&] { return launch_pw_gemm(dt, g, s); });
      } else {
        timed(LLIE_K_GEMM, kbytes, [&] { return launch_pw_gemm(dt, g, s); });
      }
    }
    // norm2 + FiLM folded into one affine
    size_t as2, ab2;
    // unfused depthwise of a 2-byte inference engine: tables / 6 and clamp01 in its prologue too (DwArgs::s6); the
    // recompute kernel takes the plain tables (it rescales the shift itself: its accumulators are already / 6)
    const bool s6dw = s6 && !fusedx;
    if (gram) {
      as2 = ar->alloc((size_t)B * w.hid * 4);
      ab2 = ar->alloc((size_t)B * w.hid * 4);
      if (!dry && !(g_skip_small & 1)) {
        GramFinalizeArgs fa{};
        fa.gtot = p<float>(gtot); fa.w1 = wptr(w.w_expand); fa.K = w.cin; fa.Chid = w.hid; fa.groups = gn_groups(w.hid); fa.P = P; fa.B = B;
        fa.gamma = wptr<float>(w.n2g); fa.beta = wptr<float>(w.n2b);
        fa.film = film ? film + w.film_off : nullptr; fa.film_stride = film_stride; fa.eps = 1e-5f;
        fa.as = p<float>(as2); fa.ab = p<float>(ab2); fa.post_scale = 0.f;
        timed(LLIE_K_OTHER, (int64_t)B * w.hid * 8, [&] { return launch_gram_finalize(dt, fa, s); }, "gram_finalize_kernel");
      }
      rel(gpart);
    } else {
      gn(h1, nullptr, w.n2g, w.n2b, film ? film + w.film_off : nullptr, film_stride, as2, ab2, &rec.n2, s6dw ? 1.f / 6.f : 0.f);
    }
    // K2: depthwise with affine + ReLU6 prologue and SE pool partials
    const int dnt = fusedx ? irbx_pool_tiles(H, W) : dwconv_ntiles(H, W);
    const size_t h2 = ar->alloc((size_t)M * w.hid * es());
    // SE pool: inference adds fixed-point channel totals into the zeroed region (one gate kernel follows); training keeps
    // the slab of tile partials (the backward pass and the 3-launch SE path read it)
    const bool ztot = !tape && g_ztot && w.hid % 128 == 0;
    const size_t pool = ztot ? 0 : ar->alloc((size_t)B * dnt * w.hid * 4);
    const size_t ptot = ztot ? ztake((size_t)B * w.hid * 8) : 0;
    if (!dry) {
      if (fusedx) {
        xa.as2 = p<float>(as2); xa.ab2 = p<float>(ab2); xa.out = p(h2);
        xa.pool = ztot ? nullptr : p<float>(pool);
        xa.pool_tot = ztot ? p<unsigned long long>(ptot) : nullptr;
        xa.nt = w.cin <= 64 ? nt_store(1, (int64_t)M * w.hid) : 0;  // 96 -> 384: the kernel itself loses more than its consumer gains
        timed(LLIE_K_DW, (int64_t)M * (w.cin + w.hid) * (int64_t)es(), [&] { return launch_expand_dw(dt, xa, s); });
      } else {
        DwArgs d{};
        d.in = p(h1.off); d.out = p(h2); d.as = p<float>(as2); d.ab = p<float>(ab2);
        d.w = wptr<float>(w.w_dw); d.pool = ztot ? nullptr : p<float>(pool);
        d.pool_tot = ztot ? p<unsigned long long>(ptot) : nullptr; d.B = B; d.H = H; d.W = W; d.C = w.hid; d.s6 = s6dw ? 1 : 0;
        d.nt = nt_store(4, (int64_t)M * w.hid);
        timed(LLIE_K_DW, 2LL * M * w.hid * (int64_t)es(), [&] { return launch_dwconv3x3(dt, d, s); });
      }
    }
    rel(as1); rel(ab1);
    if (!fusedx) rel(h1.off);
    if (gram) rel(gtot); else rel(h1.slab);
    rel(as2); rel(ab2);
    // SE MLP
    // (wide blocks of the 2-byte inference engines: fc1's pre-activations accumulate as integers in the zero-initialised region)
    const bool sepre_ok = ztot && dt != LLIE_F32 && g_se_mfma && w.hid >= 768 && w.hid % 256 == 0 && w.sq % 64 == 0 && w.sq <= 512;
    const size_t sepre = sepre_ok ? ztake((size_t)B * w.sq * 8) : 0;
    const size_t sehid = ar->alloc((size_t)B * w.sq * 4), gate = ar->alloc((size_t)B * w.hid * 4);
    const size_t semean = ar->alloc((size_t)B * w.hid * 4);
    if (!dry) {
      SeArgs e{};
      e.pool = ztot ? nullptr : p<float>(pool); e.ntiles = dnt; e.P = P;
      e.w1 = wptr(w.se_w1); e.b1 = wptr<float>(w.se_b1); e.w2 = wptr(w.se_w2); e.b2 = wptr<float>(w.se_b2);
      e.mean = p<float>(semean); e.hid = p<float>(sehid); e.gate = p<float>(gate); e.B = B; e.C = w.hid; e.Cs = w.sq;
      if (ztot) e.tot = p<unsigned long long>(ptot);
      if (sepre_ok) e.pre = p<long long>(sepre);
      if (sepre_ok && g_se_mfma && se_mlp_mfma_supported(dt, e)) {
        if (!(g_skip_small & 2)) timed(LLIE_K_SE, (int64_t)B * w.hid * 12 + 2LL * w.hid * w.sq * (int64_t)es(), [&] { return launch_se_mlp_mfma(dt, e, s); });
      } else if (ztot && w.hid <= 384) {
        if (!(g_skip_small & 2)) timed(LLIE_K_SE, (int64_t)B * w.hid * 12 + 2LL * w.hid * w.sq * (int64_t)es(), [&] { return launch_se_gate(dt, e, s); });
      } else if (!(g_skip_small & 2)) timed(LLIE_K_SE, ((int64_t)B * dnt * w.hid * 4) + 2LL * w.hid * w.sq * (int64_t)es(), [&] {
        hipError_t r1 = launch_se_fc1(dt, e, s);
        return r1 != hipSuccess ? r1 : launch_se_fc2(dt, e, s);
      });
    }
    if (!ztot) rel(pool);
    rel(sehid); rel(semean);
    // K3: project with SE gate prologue (+ skip conv as extra K segments, or identity residual)
    Tens y = new_tens(w.cout, H, W, pw_gemm_ntiles(P), w.cout_r);
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
      g.nt = nt_store(8, (int64_t)M * w.cout);
      timed(LLIE_K_GEMM, ((int64_t)M * (g.K + w.cout + (w.skip ? 0 : w.cout)) + (int64_t)w.cout * g.K) * (int64_t)es(),
            [&] { return launch_pw_gemm(dt, g, s); });
    }
    rel(h2); rel(gate);
    if (tape) {
      rec.w = (int)(&w - c->irbs.data());
      rec.x0 = x0; rec.cat = x1 != nullptr;
      if (x1) rec.x1 = *x1;
      rec.h1 = h1; rec.h2 = h2; rec.gate = gate; rec.sehid = sehid; rec.semean = semean; rec.y = y;
      tape->ops.push_back({0, (int)tape->irbs.size()});
      tape->irbs.push_back(rec);
    }
    return y;
  }

  // LinearAttention.forward (efficient_unet.py:273-308)
  Tens attn(const AttnW& w, const Tens& x) {
    const int H = x.H, W = x.W, N = H * W, M = B * N;
    const int BM = pw_gemm_tile_rows(N);
    size_t as, ab;
    AttnRec rec{};
    snprintf(tag, sizeof tag, "attn N=%d C=%d", N, x.C);
    gn(x, nullptr, w.ng, w.nb, nullptr, 0, as, ab, &rec.n1);
    const size_t qkv = ar->alloc((size_t)M * 3 * w.inner * es());
    if (!dry) {
      GemmArgs g{};
      g.seg[0] = GemmSeg{p(x.off), x.C, p<float>(as), p<float>(ab), x.C, ACT_NONE};
      g.nseg = 1; g.w = wptr(w.w_qkv); g.out = p(qkv);
      g.M = M; g.N = 3 * w.inner; g.K = x.C; g.P = N;
      timed(LLIE_K_GEMM, ((int64_t)M * (x.C + 3 * w.inner) + 3LL * w.inner * x.C) * (int64_t)es(), [&] { return launch_pw_gemm(dt, g, s); });
    }
    rel(as); rel(ab);
    const int nsplit = linattn_nsplit(N);
    const size_t kv = ar->alloc((size_t)nsplit * B * w.heads * 32 * 33 * 4);
    const size_t ao = ar->alloc((size_t)M * w.inner * es());
    if (!dry) {
      AttnArgs a{};
      a.qkv = p(qkv); a.B = B; a.N = N; a.heads = w.heads; a.kv = p<float>(kv); a.out = p(ao); a.nsplit = nsplit;
      timed(LLIE_K_OTHER, (int64_t)M * 2 * w.inner * (int64_t)es(), [&] { return launch_linattn_kv(dt, a, s); }, "linattn_kv_kernel");
      timed(LLIE_K_OTHER, (int64_t)M * 2 * w.inner * (int64_t)es(), [&] { return launch_linattn_out(dt, a, s); }, "linattn_out_kernel");
    }
    rel(qkv); rel(kv);
    Tens tmp = new_tens(x.C, H, W, pw_gemm_ntiles(N));
    if (!dry) {
      GemmArgs g{};
      g.seg[0] = GemmSeg{p(ao), w.inner, nullptr, nullptr, 0, ACT_NONE};
      g.nseg = 1; g.w = wptr(w.w_out); g.out = p(tmp.off); g.stats = p<float>(tmp.slab);
      g.M = M; g.N = x.C; g.K = w.inner; g.P = N;
      timed(LLIE_K_GEMM, ((int64_t)M * (x.C + w.inner) + (int64_t)w.inner * x.C) * (int64_t)es(), [&] { return launch_pw_gemm(dt, g, s); });
    }
    rel(ao);
    size_t as2, ab2;
    gn(tmp, nullptr, w.n2g, w.n2b, nullptr, 0, as2, ab2, &rec.n2);
    Tens y = new_tens(x.C, H, W, (N + kAffineTileRows - 1) / kAffineTileRows);
    if (!dry) {
      AffineAddArgs a{};
      a.x = p(tmp.off); a.as = p<float>(as2); a.ab = p<float>(ab2); a.res = p(x.off); a.y = p(y.off);
      a.stats = p<float>(y.slab); a.M = M; a.C = x.C; a.P = N;
      timed(LLIE_K_OTHER, 3LL * M * x.C * (int64_t)es(), [&] { return launch_affine_add(dt, a, s); }, "affine_add_kernel");
    }
    free_tens(tmp);
    rel(as2); rel(ab2);
    if (tape) {
      rec.w = (int)(&w - c->attns.data());
      rec.x = x; rec.qkv = qkv; rec.kv = kv; rec.ao = ao; rec.nsplit = nsplit; rec.tmp = tmp; rec.y = y;
      tape->ops.push_back({1, (int)tape->attns.size()});
      tape->attns.push_back(rec);
    }
    return y;
  }

  Tens conv3(const ConvW& w, const Tens& x, int mode) {
    const int Ho = mode == 0 ? x.H / 2 : x.H * 2, Wo = mode == 0 ? x.W / 2 : x.W * 2;
    Tens y = new_tens(w.c, Ho, Wo, conv3x3_ntiles(Ho, Wo), w.c_r);
    Tens u;
    snprintf(tag, sizeof tag, "conv3 mode=%d C=%d %dx%d", mode, w.c, x.H, x.W);
    if (tape && mode == 1) {
      // training: keep the upsampled tensor (the weight gradient reads it) and run the plain stride-1 conv on it
      u.C = w.c; u.H = Ho; u.W = Wo; u.valid = true;
      u.off = ar->alloc((size_t)B * Ho * Wo * w.c * es());
      if (!dry) {
        chk(launch_upsample2x(dt, p(x.off), p(u.off), B, x.H, x.W, w.c, s));
        Conv3Args a{};
        a.in = p(u.off); a.w = wptr(w.w); a.bias = wptr<float>(w.bias); a.out = p(y.off); a.stats = p<float>(y.slab);
        a.B = B; a.Hi = Ho; a.Wi = Wo; a.Cin = w.c; a.Cout = w.c; a.mode = 2;
        chk(launch_conv3x3(dt, a, s));
      }
    } else if (!dry) {
      Conv3Args a{};
      a.in = p(x.off); a.w = wptr(w.w); a.bias = wptr<float>(w.bias); a.out = p(y.off); a.stats = p<float>(y.slab);
      a.B = B; a.Hi = x.H; a.Wi = x.W; a.Cin = w.c; a.Cout = w.c; a.mode = mode;
      a.nt = nt_store(16, (int64_t)B * Ho * Wo * w.c);
      timed(LLIE_K_CONV3, ((int64_t)B * w.c * ((int64_t)x.H * x.W + (int64_t)Ho * Wo) + 9LL * w.c * w.c) * (int64_t)es(),
            [&] { return launch_conv3x3(dt, a, s); });
    }
    if (tape) {
      ConvRec rec{};
      rec.w = mode == 0 ? (int)(&w - c->downs.data()) : (int)(&w - c->ups.data());
      rec.up = mode != 0; rec.x = x; rec.u = u; rec.y = y;
      tape->ops.push_back({2, (int)tape->convs.size()});
      tape->convs.push_back(rec);
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
    zbegin((int64_t)S * S, [&](Run& d) { d.unet(nullptr, nullptr, nullptr, uniform_t, nullptr); });
    const size_t temb = ar->alloc((size_t)rows * T * 4), stemb = ar->alloc((size_t)rows * T * 4);
    const size_t film = ar->alloc((size_t)rows * F * 4);
    if (!dry) {
      TimeArgs ta{};
      ta.t = t; ta.rows = rows; ta.dim = g.base_channels; ta.freqs = wptr<float>(c->freqs); ta.T = T;
      ta.w1 = wptr<float>(c->t_w1); ta.b1 = wptr<float>(c->t_b1); ta.w3 = wptr<float>(c->t_w3); ta.b3 = wptr<float>(c->t_b3);
      ta.temb = p<float>(temb); ta.silu_temb = p<float>(stemb);
      snprintf(tag, sizeof tag, "time");
      timed(LLIE_K_OTHER, 0, [&] { return launch_time_embed(ta, s); }, "time_embed_kernel");
      FilmArgs fa{};
      fa.silu_temb = p<float>(stemb); fa.rows = rows; fa.T = T; fa.wf = wptr<float>(c->film_w); fa.bf = wptr<float>(c->film_b);
      fa.film = p<float>(film); fa.F = F;
      timed(LLIE_K_OTHER, (int64_t)F * T * 4, [&] { return launch_film(fa, s); }, "film_kernel");
    }
    const float* filmp = p<float>(film);
    const int64_t fstride = uniform_t ? 0 : F;

    Tens h = new_tens(c->channels[0], S, S, init_conv_ntiles(S, S, dt != LLIE_F32), c->channels_r[0]);
    if (!dry) {
      InitConvArgs a{};
      const int half = g.in_channels / 2;
      a.x0 = lat; a.x1 = cond; a.c0 = half; a.c1 = g.in_channels - half;
      a.w = wptr<float>(c->init_w); a.bias = wptr<float>(c->init_b); a.out = p(h.off); a.stats = p<float>(h.slab);
      a.wp = dt != LLIE_F32 ? wptr(c->init_wp) : nullptr;
      a.B = B; a.H = S; a.W = S; a.Cout = c->channels[0];
      snprintf(tag, sizeof tag, "init_conv");
      timed(LLIE_K_OTHER, (int64_t)B * S * S * (g.in_channels * 4 + c->channels[0] * (int64_t)es()), [&] { return launch_init_conv(dt, a, s); }, "init_conv_kernel");
    }
    if (tape) {
      tape->temb = temb; tape->stemb = stemb; tape->film = film; tape->h0 = h;
      tape->lat = lat; tape->cond = cond; tape->t = t; tape->B = B;
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
    GnRec finrec{};
    snprintf(tag, sizeof tag, "final_norm");
    gn(h, nullptr, c->fin_g, c->fin_b, nullptr, 0, as, ab, &finrec);
    if (tape) { tape->fin = finrec; tape->hlast = h; }
    if (!dry) {
      FinalConvArgs a{};
      a.in = p(h.off); a.as = p<float>(as); a.ab = p<float>(ab); a.w = wptr<float>(c->fin_w); a.bias = wptr<float>(c->fin_bias);
      a.out = eps; a.B = B; a.H = S; a.W = S; a.C = c->channels[0]; a.Cout = g.out_channels;
      a.wp = dt != LLIE_F32 ? wptr(c->fin_wp) : nullptr;
      if (fs) {
        a.fuse_step = 1; a.coef = fs->coef; a.sample = lat; a.noise = fs->noise; a.prev = fs->prev; a.clamped = fs->clamped;
      }
      snprintf(tag, sizeof tag, "final_conv");
      timed(LLIE_K_OTHER, (int64_t)B * S * S * (c->channels[0] * (int64_t)es() + 3 * 4 * (fs ? 4 : 1)), [&] { return launch_final_conv(dt, a, s); }, "final_conv_kernel");
    }
    free_tens(h);
    rel(as); rel(ab);
    rel(temb); rel(stemb); rel(film);
  }

  // single-operator forward: fp32 NCHW in/out
  void module(const float* x, const float* temb, float* y, int H, int W) {
    const llie_config& g = c->cfg;
    const int P = H * W;
    const int split = (g.kind == LLIE_IRB) ? g.base_channels : 0;  // IRB: optional virtual-concat split point
    zbegin((int64_t)P, [&](Run& d) { d.module(nullptr, nullptr, nullptr, H, W); });
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
      rel(st); rel(film);
      if (tape) { tape->stemb = st; tape->film = film; }
    } else if (g.kind == LLIE_ATTN) {
      out = attn(c->attns[0], x0);
    } else if (g.kind == LLIE_SE) {
      // SqueezeExcitation.forward (efficient_unet.py:96-100): the 64-pixel (sum, sum of squares) slab of the layout
      // conversion doubles as the pool partials (every second entry), then the block's own SE kernels and x * gate
      const IrbW& w = c->irbs[0];
      const int C = w.hid;
      const size_t sehid = ar->alloc((size_t)B * w.sq * 4), gate = ar->alloc((size_t)B * C * 4);
      const size_t semean = ar->alloc((size_t)B * C * 4), zero = ar->alloc((size_t)B * C * 4);
      out = new_tens(C, H, W, P / kAffineTileRows);
      if (!dry) {
        SeArgs e{};
        e.pool = p<float>(x0.slab); e.ntiles = P / 64; e.pool_stride = 2 * C; e.P = P;
        e.w1 = wptr(w.se_w1); e.b1 = wptr<float>(w.se_b1); e.w2 = wptr(w.se_w2); e.b2 = wptr<float>(w.se_b2);
        e.mean = p<float>(semean); e.hid = p<float>(sehid); e.gate = p<float>(gate); e.B = B; e.C = C; e.Cs = w.sq;
        chk(launch_se_fc1(dt, e, s));
        chk(launch_se_fc2(dt, e, s));
        chk(launch_fill_zero(p(zero), (int64_t)B * C * 4, s));
        AffineAddArgs a{};
        a.x = p(x0.off); a.as = p<float>(gate); a.ab = p<float>(zero); a.res = nullptr; a.y = p(out.off);
        a.stats = p<float>(out.slab); a.M = B * P; a.C = C; a.P = P;
        chk(launch_affine_add(dt, a, s));
      }
      rel(sehid); rel(gate); rel(semean); rel(zero);
    } else if (g.kind == LLIE_DOWN) {
      out = conv3(c->downs[0], x0, 0);
    } else {
      out = conv3(c->ups[0], x0, 1);
    }
    if (!dry && y) chk(launch_nhwc_to_nchw(dt, p(out.off), y, B, out.C, out.H * out.W, s));
    if (tape) { tape->h0 = x0; tape->hlast = out; tape->B = B; mod_x1 = x1; }
    free_tens(out);
    free_tens(x0);
    free_tens(x1);
  }
  Tens mod_x1;  // training, IRB module with a virtual-concat input: the second segment
};

// ---------------------------------------------------------------------------------------------
// Backward pass over the tape.  Reverse-mode over the recorded operators: every forward
// tensor's gradient lives in the same workspace (NHWC T), keyed by the tensor's offset; an operator takes
// the gradient of its output, writes / accumulates the gradients of its inputs and the fp32 parameter
// gradients (reference state_dict layout, flat buffer `grads` at Param::goff).
struct Back {
  llie_ctx* c;
  Arena* ar;
  hipStream_t s;
  char* ws;
  bool dry;
  int B;
  int dt;
  Tape* tp;
  float* grads;
  hipError_t err = hipSuccess;
  std::map<size_t, size_t> gmap;  // forward tensor offset -> gradient offset
  // Weight gradients do not feed the activation-gradient chain, so they run on a side stream (`async`): fork() makes
  // the side stream wait for what the main stream has enqueued so far, defer() keeps a buffer the side stream may still
  // read until join(), where the main stream waits for the side stream and the deferred buffers are released.  The dry
  // run follows the same release order, so the workspace plan accounts for the longer lifetimes.
  bool async = false;
  hipStream_t s2 = nullptr;
  std::vector<size_t> deferred;
  hipStream_t side() const { return (async && s2) ? s2 : s; }
  void fork() {
    if (!async || dry || !s2) return;
    chk(hipEventRecord(c->ev_fork, s));
    chk(hipStreamWaitEvent(s2, c->ev_fork, 0));
  }
  void defer(size_t off) {
    if (async) deferred.push_back(off);
    else ar->free(off);
  }
  void join() {
    if (async && !dry && s2) {
      chk(hipEventRecord(c->ev_join, s2));
      chk(hipStreamWaitEvent(s, c->ev_join, 0));
    }
    for (size_t off : deferred) ar->free(off);
    deferred.clear();
  }

  template <typename T = void> T* wptr(size_t off) const { return reinterpret_cast<T*>(c->blob + off); }
  template <typename T = void> T* p(size_t off) const { return reinterpret_cast<T*>(ws + off); }
  void chk(hipError_t e) { if (e != hipSuccess && err == hipSuccess) err = e; }
  size_t es() const { return elem_size(dt); }
  size_t alloc(size_t bytes) { return ar->alloc(bytes); }
  float* gp(int param) const { return grads + c->params[param].goff; }
  int pidx(const char* key) const { return c->index.at(key); }

  size_t take_grad(const Tens& t) {
    auto it = gmap.find(t.off);
    if (it == gmap.end()) { if (err == hipSuccess) err = hipErrorInvalidValue; return 0; }
    const size_t g = it->second;
    gmap.erase(it);
    return g;
  }
  // gradient buffer of a forward tensor: the existing one (existed = true) or a fresh allocation
  size_t grad_of(const Tens& t, bool& existed) {
    auto it = gmap.find(t.off);
    existed = it != gmap.end();
    if (existed) return it->second;
    const size_t g = alloc((size_t)B * t.H * t.W * t.C * es());
    gmap[t.off] = g;
    return g;
  }
  void add_grad(const Tens& t, size_t g) {  // hand over `g` as (part of) t's gradient
    auto it = gmap.find(t.off);
    if (it == gmap.end()) { gmap[t.off] = g; return; }
    if (!dry) chk(launch_add_into(dt, p(it->second), p(g), (int64_t)B * t.H * t.W * t.C, s));
    ar->free(g);
  }

  // out[M][N] = in[M][K] * W[N][K]^T on the forward GEMM kernel (W = a transposed weight copy)
  void gemm(size_t in, int K, const void* w, size_t out, int N, int M, int P, size_t slab = 0, size_t dot = 0, bool with_dot = false) {
    if (dry) return;
    GemmArgs g{};
    g.seg[0] = GemmSeg{p(in), K, nullptr, nullptr, 0, ACT_NONE};
    g.nseg = 1; g.w = w; g.out = p(out); g.M = M; g.N = N; g.K = K; g.P = P;
    if (with_dot) { g.stats = p<float>(slab); g.dot = p(dot); }  // slab[b][tile][0][n] = sum over the tile's rows of out*dot
    chk(launch_pw_gemm(dt, g, s));
  }
  struct Geo { int Ho, Wo, Hi, Wi, stride, dy, dx; };
  void wgrad(size_t g, int N, const GemmSeg* segs, int nseg, int K, Geo geo, float* out, int64_t ldn, int64_t ldk, int64_t off,
             int ntap = 1, int nstore = 0, int kstore = 0) {
    const int M = B * geo.Ho * geo.Wo;
    const int ms = wgrad_msplit(dt, M, N, K, ntap);
    const size_t part = alloc((size_t)ms * ntap * N * K * 4);
    if (!dry) {
      WgradArgs a{};
      a.g = p(g); a.N = N; a.nseg = nseg; a.K = K;
      for (int i = 0; i < nseg; ++i) a.seg[i] = segs[i];
      a.B = B; a.Ho = geo.Ho; a.Wo = geo.Wo; a.Hi = geo.Hi; a.Wi = geo.Wi; a.stride = geo.stride; a.dy = geo.dy; a.dx = geo.dx;
      a.partial = p<float>(part); a.out = out; a.ldn = ldn; a.ldk = ldk; a.off = off; a.msplit = ms; a.ntap = ntap; a.nstore = nstore; a.kstore = kstore;
      chk(launch_wgrad(dt, a, side()));
    }
    defer(part);
  }

  // activation backward + GroupNorm backward coefficients + norm parameter gradients at one norm site.
  //   g: gradient w.r.t. the activation output act(norm(x)) [M][C]; dz is written over g when act != none.
  struct Coef { size_t A, Bq, Cq; };
  //   pre_slab / pre_tiles: the producer already applied the activation derivative and wrote the partial sums
  //   (depthwise backward epilogue); then only the reduction and the coefficient kernels run here.
  Coef gn_site(size_t g, const Tens& x0, const Tens* x1, const GnRec& rec, int act, size_t gamma, size_t beta,
               float* dgamma, float* dbeta, const float* film, int64_t fstride, float* dfilm, int64_t dfstride,
               size_t pre_slab = 0, int pre_tiles = 0) {
    const int C = x0.C + (x1 ? x1->C : 0), P = x0.H * x0.W, M = B * P, nt = pre_tiles ? pre_tiles : P / 64;
    const size_t slab = pre_tiles ? pre_slab : alloc((size_t)B * nt * 2 * C * 4);
    const size_t S = alloc((size_t)B * 2 * C * 4);
    Coef k{alloc((size_t)B * C * 4), alloc((size_t)B * C * 4), alloc((size_t)B * C * 4)};
    const size_t dG = alloc((size_t)B * C * 4), dBc = alloc((size_t)B * C * 4);
    if (!dry) {
      BwdMaskArgs m{};
      m.g = p(g); m.x0 = p(x0.off); m.c0 = x0.C; m.x1 = x1 ? p(x1->off) : nullptr; m.c1 = x1 ? x1->C : 0;
      m.as = p<float>(rec.as); m.ab = p<float>(rec.ab); m.act = act; m.dz = act == ACT_NONE ? nullptr : p(g);
      m.slab = p<float>(slab); m.M = M; m.C = C; m.P = P;
      if (!pre_tiles) chk(launch_bwd_mask_reduce(dt, m, s));
      GnBwdArgs a{};
      if (C / 32 <= 64) {  // the coefficient kernel sums the tile partials of its group itself (no slab_reduce launch)
        a.slab = p<float>(slab); a.ntiles = nt;
      } else {
        chk(launch_slab_reduce(p<float>(slab), p<float>(S), B, nt, 2, 2, C, s));
        a.S = p<float>(S);
      }
      a.mean = p<float>(rec.mean); a.rstd = p<float>(rec.rstd); a.gamma = wptr<float>(gamma);
      a.film = film; a.film_stride = fstride; a.C = C; a.groups = 32; a.P = P; a.B = B;
      a.A = p<float>(k.A); a.Bq = p<float>(k.Bq); a.Cq = p<float>(k.Cq); a.dG = p<float>(dG); a.dBc = p<float>(dBc);
      chk(launch_gn_bwd_coef(a, s));
      GnParamGradArgs q{};
      q.dG = p<float>(dG); q.dBc = p<float>(dBc); q.gamma = wptr<float>(gamma); q.beta = wptr<float>(beta);
      q.film = film; q.film_stride = fstride; q.dgamma = dgamma; q.dbeta = dbeta; q.dfilm = dfilm; q.dfilm_stride = dfstride;
      q.B = B; q.C = C;
      chk(launch_gn_param_grad(q, s));
    }
    if (!pre_tiles) ar->free(slab);
    ar->free(S); ar->free(dG); ar->free(dBc);
    return k;
  }
  void free_coef(const Coef& k) { ar->free(k.A); ar->free(k.Bq); ar->free(k.Cq); }
  void apply(size_t dz, const Tens& x0, const Tens* x1, const Coef& k, size_t add0, bool has_add0, size_t a10, bool has10,
             size_t a11, bool has11, size_t dx0, size_t dx1) {
    if (dry) return;
    GnApplyArgs a{};
    a.dz = p(dz); a.x0 = p(x0.off); a.c0 = x0.C; a.x1 = x1 ? p(x1->off) : nullptr; a.c1 = x1 ? x1->C : 0;
    a.A = p<float>(k.A); a.Bq = p<float>(k.Bq); a.Cq = p<float>(k.Cq);
    a.add0 = has_add0 ? p(add0) : nullptr; a.add1_0 = has10 ? p(a10) : nullptr; a.add1_1 = has11 ? p(a11) : nullptr;
    a.dx0 = p(dx0); a.dx1 = x1 ? p(dx1) : nullptr; a.M = B * x0.H * x0.W; a.P = x0.H * x0.W;
    chk(launch_gn_bwd_apply(dt, a, s));
  }

  // ---- InvertedResidualBlock
  void irb_bwd(const IrbRec& r, size_t dfilm, int F) {
    const IrbW& w = c->irbs[r.w];
    const int H = r.x0.H, W = r.x0.W, P = H * W, M = B * P, hid = w.hid, cin = w.cin, cout = w.cout, pf = w.p_first;
    const Tens* x1 = r.cat ? &r.x1 : nullptr;
    const size_t dY = take_grad(r.y);
    const Geo g11{H, W, H, W, 1, 0, 0};
    // project (+ skip) input gradients
    const size_t da3 = alloc((size_t)M * hid * es());
    const int gtiles = P / pw_gemm_tile_rows(P);
    const size_t gslab = alloc((size_t)B * gtiles * 2 * hid * 4);  // d(gate) partials from the GEMM's epilogue: sum_px da3*h2
    gemm(dY, cout, wptr(w.w_proj_t), da3, hid, M, P, gslab, r.h2, true);
    size_t dxs = 0;
    if (w.skip) {
      dxs = alloc((size_t)M * cin * es());
      gemm(dY, cout, wptr<char>(w.w_proj_t) + (size_t)hid * cout * es(), dxs, cin, M, P);
    }
    {  // project / skip weight gradients (side stream: they only need dY, which the previous operator produced)
      fork();
      GemmSeg sg[2];
      sg[0] = GemmSeg{dry ? nullptr : p(r.h2), hid, dry ? nullptr : p<float>(r.gate), nullptr, hid, ACT_NONE};
      wgrad(dY, cout, sg, 1, hid, g11, gp(pf + 10), hid, 1, 0);
      if (w.skip) {
        sg[0] = GemmSeg{dry ? nullptr : p(r.x0.off), r.x0.C, nullptr, nullptr, 0, ACT_NONE};
        if (x1) sg[1] = GemmSeg{dry ? nullptr : p(x1->off), x1->C, nullptr, nullptr, 0, ACT_NONE};
        wgrad(dY, cout, sg, x1 ? 2 : 1, cin, g11, gp(pf + 13), cin, 1, 0);
      }
    }
    // SE: dgate = sum_px da3*h2, then the two-layer MLP backwards to d(mean)
    const size_t dgate = alloc((size_t)B * hid * 4), dpre2 = alloc((size_t)B * hid * 4), dr = alloc((size_t)B * w.sq * 4);
    const size_t dmean = alloc((size_t)B * hid * 4);
    {
      const size_t sescr = alloc((size_t)std::max(linear_dx_chunks(hid) * w.sq, linear_dx_chunks(w.sq) * hid) * B * 4);
      if (!dry) {
        chk(launch_slab_reduce(p<float>(gslab), p<float>(dgate), B, gtiles, 2, 1, hid, s));
        chk(launch_sigmoid_bwd(p<float>(dgate), p<float>(r.gate), p<float>(dpre2), (int64_t)B * hid, s));
        chk(launch_linear_dw(p<float>(dpre2), hid, p<float>(r.sehid), gp(pf + 8), gp(pf + 9), B, hid, w.sq, s));
        chk(launch_linear_dx(dt, p<float>(dpre2), hid, wptr(w.se_w2), p<float>(dr), B, hid, w.sq, s, p<float>(sescr)));
        chk(launch_relu6_bwd(p<float>(dr), p<float>(r.sehid), p<float>(dr), (int64_t)B * w.sq, s));
        chk(launch_linear_dw(p<float>(dr), w.sq, p<float>(r.semean), gp(pf + 6), gp(pf + 7), B, w.sq, hid, s));
        chk(launch_linear_dx(dt, p<float>(dr), w.sq, wptr(w.se_w1), p<float>(dmean), B, w.sq, hid, s, p<float>(sescr)));
        chk(launch_scale_rows(p<float>(dmean), p<float>(dmean), (int64_t)B * hid, 1.f / (float)P, s));
      }
      ar->free(gslab); ar->free(sescr);
    }
    // depthwise: input gradient (same kernel, flipped taps, prologue dh2 = da3*gate + dmean/P) and weight gradient
    const size_t da2 = alloc((size_t)M * hid * es());
    const int dztiles = dwconv_ntiles(H, W);
    const size_t dzslab = alloc((size_t)B * dztiles * 2 * hid * 4);
    {
      const size_t part = alloc((size_t)B * dw_wgrad_strips(H, W) * 9 * hid * 4);
      if (!dry) {
        DwArgs d{};  // writes dz2 = da2 * relu6'(norm2(h1)) and the (sum dz, sum dz*h1) partials in its epilogue
        d.in = p(da3); d.out = p(da2); d.as = p<float>(r.gate); d.ab = p<float>(dmean); d.w = wptr<float>(w.w_dw_flip);
        d.pool = nullptr; d.B = B; d.H = H; d.W = W; d.C = hid; d.no_act = 1;
        d.bx = p(r.h1.off); d.bas = p<float>(r.n2.as); d.bab = p<float>(r.n2.ab); d.bslab = p<float>(dzslab);
        chk(launch_dwconv3x3(dt, d, s));
        fork();  // da3 and d(mean) are enqueued: the depthwise weight gradient may run beside the rest of the chain
        DwWgradArgs q{};
        q.g = p(da3); q.gs = p<float>(r.gate); q.gb = p<float>(dmean); q.h = p(r.h1.off); q.as = p<float>(r.n2.as);
        q.ab = p<float>(r.n2.ab); q.partial = p<float>(part); q.out = gp(pf + 5); q.B = B; q.H = H; q.W = W; q.C = hid;
        chk(launch_dw_wgrad(dt, q, side()));
      }
      defer(part);
    }
    defer(da3); ar->free(dgate); ar->free(dpre2); ar->free(dr); defer(dmean);
    // norm2 + FiLM + ReLU6
    const float* film = dry ? nullptr : p<float>(tp->film) + w.film_off;
    float* dfl = dry ? nullptr : p<float>(dfilm) + w.film_off;
    Coef k2 = gn_site(da2, r.h1, nullptr, r.n2, ACT_RELU6, w.n2g, w.n2b, gp(pf + 2), gp(pf + 3), film, F, dfl, F, dzslab, dztiles);
    ar->free(dzslab);
    apply(da2, r.h1, nullptr, k2, 0, false, 0, false, 0, false, da2, 0);  // dh1, in place
    free_coef(k2);
    // expand
    const size_t da1 = alloc((size_t)M * cin * es());
    gemm(da2, hid, wptr(w.w_expand_t), da1, cin, M, P);
    {
      GemmSeg sg[2];
      sg[0] = GemmSeg{dry ? nullptr : p(r.x0.off), r.x0.C, dry ? nullptr : p<float>(r.n1.as), dry ? nullptr : p<float>(r.n1.ab), cin, ACT_RELU6};
      if (x1) sg[1] = GemmSeg{dry ? nullptr : p(x1->off), x1->C, dry ? nullptr : p<float>(r.n1.as) + r.x0.C,
                              dry ? nullptr : p<float>(r.n1.ab) + r.x0.C, cin, ACT_RELU6};
      fork();  // dh1 (in da2) is complete
      wgrad(da2, hid, sg, x1 ? 2 : 1, cin, g11, gp(pf + 4), cin, 1, 0);
    }
    defer(da2);
    // norm1 + ReLU6, then the block input (residual / skip-conv gradient added, existing gradients accumulated)
    Coef k1 = gn_site(da1, r.x0, x1, r.n1, ACT_RELU6, w.n1g, w.n1b, gp(pf + 0), gp(pf + 1), nullptr, 0, nullptr, 0);
    bool e0 = false, e1 = false;
    const size_t g0 = grad_of(r.x0, e0);
    const size_t g1 = x1 ? grad_of(*x1, e1) : 0;
    apply(da1, r.x0, x1, k1, w.skip ? dxs : dY, true, g0, e0, g1, e1, g0, g1);
    free_coef(k1);
    ar->free(da1);
    if (w.skip) ar->free(dxs);
    defer(dY);
    join();
  }

  // ---- LinearAttention
  void attn_bwd(const AttnRec& r) {
    const AttnW& w = c->attns[r.w];
    const int H = r.x.H, W = r.x.W, N = H * W, M = B * N, C = w.c, inner = w.inner, pf = w.p_first;
    const size_t dY = take_grad(r.y);
    const Geo g11{H, W, H, W, 1, 0, 0};
    // y = norm2(tmp) + x
    Coef k2 = gn_site(dY, r.tmp, nullptr, r.n2, ACT_NONE, w.n2g, w.n2b, gp(pf + 4), gp(pf + 5), nullptr, 0, nullptr, 0);
    const size_t dtmp = alloc((size_t)M * C * es());
    apply(dY, r.tmp, nullptr, k2, 0, false, 0, false, 0, false, dtmp, 0);
    free_coef(k2);
    // to_out
    const size_t dao = alloc((size_t)M * inner * es());
    gemm(dtmp, C, wptr(w.w_out_t), dao, inner, M, N);
    {
      GemmSeg sg{dry ? nullptr : p(r.ao), inner, nullptr, nullptr, 0, ACT_NONE};
      fork();
      wgrad(dtmp, C, &sg, 1, inner, g11, gp(pf + 3), inner, 1, 0);
    }
    defer(dtmp);
    // attention core
    const size_t dqkv = alloc((size_t)M * 3 * inner * es());
    {
      const int nt = N / 64;
      const size_t part = alloc((size_t)B * w.heads * nt * 32 * 33 * 4), tot = alloc((size_t)B * w.heads * 32 * 33 * 4);
      if (!dry) {
        AttnBwdArgs a{};
        a.qkv = p(r.qkv); a.dout = p(dao); a.dqkv = p(dqkv); a.kv = p<float>(r.kv); a.nsplit = r.nsplit;
        a.dkv = p<float>(part); a.B = B; a.N = N; a.heads = w.heads;
        chk(launch_linattn_bwd_q(dt, a, s));
        chk(launch_slab_reduce(p<float>(part), p<float>(tot), B * w.heads, nt, 1, 1, 32 * 33, s));
        a.dkv = p<float>(tot);
        chk(launch_linattn_bwd_kv(dt, a, s));
      }
      ar->free(part); ar->free(tot);
    }
    ar->free(dao);
    // to_qkv
    const size_t dxn = alloc((size_t)M * C * es());
    gemm(dqkv, 3 * inner, wptr(w.w_qkv_t), dxn, C, M, N);
    {
      GemmSeg sg{dry ? nullptr : p(r.x.off), C, dry ? nullptr : p<float>(r.n1.as), dry ? nullptr : p<float>(r.n1.ab), C, ACT_NONE};
      fork();
      wgrad(dqkv, 3 * inner, &sg, 1, C, g11, gp(pf + 2), C, 1, 0);
    }
    defer(dqkv);
    // norm (no activation) + residual
    Coef k1 = gn_site(dxn, r.x, nullptr, r.n1, ACT_NONE, w.ng, w.nb, gp(pf + 0), gp(pf + 1), nullptr, 0, nullptr, 0);
    bool e0 = false;
    const size_t g0 = grad_of(r.x, e0);
    apply(dxn, r.x, nullptr, k1, dY, true, g0, e0, 0, false, g0, 0);
    free_coef(k1);
    ar->free(dxn);
    ar->free(dY);
    join();
  }

  // ---- Downsample / Upsample convolutions
  void conv_bwd(const ConvRec& r) {
    const ConvW& w = r.up ? c->ups[r.w] : c->downs[r.w];
    const int C = w.c, Ho = r.y.H, Wo = r.y.W, Mo = B * Ho * Wo, pf = w.p_first;
    const size_t dY = take_grad(r.y);
    {  // bias gradient: column sums of dY
      const int nt = Ho * Wo / 64;
      const size_t slab = alloc((size_t)B * nt * 2 * C * 4), S = alloc((size_t)B * C * 4);
      if (!dry) {
        BwdMaskArgs m{};
        m.g = p(dY); m.act = ACT_NONE; m.slab = p<float>(slab); m.M = Mo; m.C = C; m.P = Ho * Wo;
        chk(launch_bwd_mask_reduce(dt, m, s));
        chk(launch_slab_reduce(p<float>(slab), p<float>(S), B, nt, 2, 1, C, s));
        chk(launch_batch_sum(p<float>(S), gp(pf + 1), B, C, C, s));
      }
      ar->free(slab); ar->free(S);
    }
    const Tens& src = r.up ? r.u : r.x;  // what the conv itself read
    {
      GemmSeg sg{dry ? nullptr : p(src.off), C, nullptr, nullptr, 0, ACT_NONE};
      const Geo geo{Ho, Wo, src.H, src.W, r.up ? 1 : 2, 0, 0};
      fork();
      wgrad(dY, C, &sg, 1, C, geo, gp(pf + 0), (int64_t)C * 9, 9, 0, 9);
    }
    // input gradient: stride-1 conv with flipped / transposed weights over dY (zero-dilated for the stride-2 conv)
    size_t din = dY;
    if (!r.up) {
      din = alloc((size_t)B * r.x.H * r.x.W * C * es());
      if (!dry) chk(launch_dilate2x(dt, p(dY), p(din), B, Ho, Wo, C, s));
    }
    const size_t dsrc = alloc((size_t)B * src.H * src.W * C * es());
    if (!dry) {
      Conv3Args a{};
      a.in = p(din); a.w = wptr(w.w_t); a.bias = nullptr; a.out = p(dsrc); a.stats = nullptr;
      a.B = B; a.Hi = src.H; a.Wi = src.W; a.Cin = C; a.Cout = C; a.mode = 2;
      chk(launch_conv3x3(dt, a, s));
    }
    if (!r.up) ar->free(din);
    defer(dY);
    if (r.up) {
      const size_t dx = alloc((size_t)B * r.x.H * r.x.W * C * es());
      if (!dry) chk(launch_upsample2x_bwd(dt, p(dsrc), p(dx), B, r.x.H, r.x.W, C, s));
      ar->free(dsrc);
      add_grad(r.x, dx);
    } else {
      add_grad(r.x, dsrc);
    }
    join();
  }

  void run_ops(size_t dfilm, int F) {
    for (int i = (int)tp->ops.size() - 1; i >= 0; --i) {
      const TapeOp& op = tp->ops[i];
      if (op.kind == 0) irb_bwd(tp->irbs[op.idx], dfilm, F);
      else if (op.kind == 1) attn_bwd(tp->attns[op.idx]);
      else conv_bwd(tp->convs[op.idx]);
    }
  }
  // FiLM Linear of every block: weight / bias gradients, and d(silu(temb)) summed over all FiLM rows
  void film_bwd(size_t dfilm, int F, int T, size_t dstemb) {
    const size_t scratch = alloc((size_t)linear_dx_chunks(F) * B * T * 4);
    if (!dry) {
      for (const IrbW& w : c->irbs)
        chk(launch_linear_dw(p<float>(dfilm) + w.film_off, F, p<float>(tp->stemb), gp(w.p_first + 11), gp(w.p_first + 12), B,
                             2 * w.hid, T, s));
      chk(launch_linear_dx(0, p<float>(dfilm), F, wptr(c->film_w), p<float>(dstemb), B, F, T, s, p<float>(scratch)));
    }
    ar->free(scratch);
  }

  // ---- whole UNet
  void unet(const float* deps) {
    const llie_config& g = c->cfg;
    const int S = g.image_size, C0 = c->channels[0], P = S * S, M = B * P, T = g.time_embed_dim, F = c->film_rows;
    const size_t dfilm = alloc((size_t)B * F * 4);
    // output head
    const size_t da = alloc((size_t)M * C0 * es());
    {
      // weight gradient of the head on the MFMA weight-gradient GEMM: d(eps) packed to [M][32] NHWC is the "g"
      // operand (3 real rows), silu(norm(h)) recomputed in the prologue the other; bias = plane sums of d(eps).
      // It only needs d(eps) and forward tensors: side stream, joined with the first operator.
      const size_t g32 = alloc((size_t)M * 32 * es());
      const int nt = P / 64;
      const size_t bslab = alloc((size_t)B * nt * 2 * 32 * 4), bS = alloc((size_t)B * 32 * 4);
      if (!dry) {
        FinalBwdArgs a{};
        a.deps = deps; a.w = wptr<float>(c->fin_w); a.da = p(da);
        a.B = B; a.H = S; a.W = S; a.C = C0; a.Cout = g.out_channels;
        chk(launch_final_bwd_data(dt, a, s));
        fork();
        chk(launch_pack_planes(dt, deps, nullptr, g.out_channels, 0, p(g32), B, P, side()));
        BwdMaskArgs m{};  // bias gradient = column sums of the packed d(eps)
        m.g = p(g32); m.act = ACT_NONE; m.slab = p<float>(bslab); m.M = M; m.C = 32; m.P = P;
        chk(launch_bwd_mask_reduce(dt, m, side()));
        chk(launch_slab_reduce(p<float>(bslab), p<float>(bS), B, nt, 2, 1, 32, side()));
        chk(launch_batch_sum(p<float>(bS), gp(pidx("final_conv.bias")), B, 32, g.out_channels, side()));
      }
      defer(bslab); defer(bS);
      GemmSeg sg{dry ? nullptr : p(tp->hlast.off), C0, dry ? nullptr : p<float>(tp->fin.as), dry ? nullptr : p<float>(tp->fin.ab), C0, ACT_SILU};
      const Geo geo{S, S, S, S, 1, 0, 0};
      wgrad(g32, 32, &sg, 1, C0, geo, gp(pidx("final_conv.weight")), (int64_t)C0 * 9, 9, 0, 9, g.out_channels, 0);
      defer(g32);  // released at the first operator's join
    }
    Coef kf = gn_site(da, tp->hlast, nullptr, tp->fin, ACT_SILU, c->fin_g, c->fin_b, gp(pidx("final_norm.weight")),
                      gp(pidx("final_norm.bias")), nullptr, 0, nullptr, 0);
    apply(da, tp->hlast, nullptr, kf, 0, false, 0, false, 0, false, da, 0);
    free_coef(kf);
    gmap[tp->hlast.off] = da;
    run_ops(dfilm, F);
    // input conv
    {
      // dW[co][ci][tap] on the same GEMM: g = d(h0) [M][C0], the other operand the two fp32 input planes packed to
      // [M][32] NHWC (6 real channels); bias = column sums of d(h0)
      const size_t g0 = take_grad(tp->h0);
      const int half = g.in_channels / 2, nt = P / 64;
      const size_t x32 = alloc((size_t)M * 32 * es());
      const size_t slab = alloc((size_t)B * nt * 2 * C0 * 4), S1 = alloc((size_t)B * C0 * 4);
      if (!dry) {
        chk(launch_pack_planes(dt, tp->lat, tp->cond, half, g.in_channels - half, p(x32), B, P, s));
        BwdMaskArgs m{};
        m.g = p(g0); m.act = ACT_NONE; m.slab = p<float>(slab); m.M = M; m.C = C0; m.P = P;
        chk(launch_bwd_mask_reduce(dt, m, s));
        chk(launch_slab_reduce(p<float>(slab), p<float>(S1), B, nt, 2, 1, C0, s));
        chk(launch_batch_sum(p<float>(S1), gp(pidx("init_conv.bias")), B, C0, c->channels_r[0], s));
      }
      GemmSeg sg{dry ? nullptr : p(x32), 32, nullptr, nullptr, 0, ACT_NONE};
      const Geo geo{S, S, S, S, 1, 0, 0};
      fork();
      wgrad(g0, C0, &sg, 1, 32, geo, gp(pidx("init_conv.weight")), (int64_t)g.in_channels * 9, 9, 0, 9, c->channels_r[0], g.in_channels);
      ar->free(slab); ar->free(S1);
      defer(x32); defer(g0);
      join();
    }
    // time embedding MLP (efficient_unet.py:412-417): temb = W3 silu(W1 emb + b1) + b3, FiLM reads silu(temb)
    const int dim = g.base_channels;
    const size_t dtemb = alloc((size_t)B * T * 4), emb = alloc((size_t)B * dim * 4), z1 = alloc((size_t)B * T * 4);
    const size_t a1 = alloc((size_t)B * T * 4), dh = alloc((size_t)B * T * 4);
    film_bwd(dfilm, F, T, dtemb);
    if (!dry) {
      chk(launch_silu_bwd(p<float>(dtemb), p<float>(tp->temb), p<float>(dtemb), (int64_t)B * T, s));
      chk(launch_sin_embed(tp->t, wptr<float>(c->freqs), p<float>(emb), B, dim, s));
      FilmArgs fa{};
      fa.silu_temb = p<float>(emb); fa.rows = B; fa.T = dim; fa.wf = wptr<float>(c->t_w1); fa.bf = wptr<float>(c->t_b1);
      fa.film = p<float>(z1); fa.F = T;
      chk(launch_film(fa, s));
      chk(launch_silu_rows(p<float>(z1), p<float>(a1), (int64_t)B * T, s));
      chk(launch_linear_dw(p<float>(dtemb), T, p<float>(a1), gp(pidx("time_mlp.3.weight")), gp(pidx("time_mlp.3.bias")), B, T, T, s));
      chk(launch_linear_dx(0, p<float>(dtemb), T, wptr(c->t_w3), p<float>(dh), B, T, T, s));
      chk(launch_silu_bwd(p<float>(dh), p<float>(z1), p<float>(dh), (int64_t)B * T, s));
      chk(launch_linear_dw(p<float>(dh), T, p<float>(emb), gp(pidx("time_mlp.1.weight")), gp(pidx("time_mlp.1.bias")), B, T, dim, s));
    }
    ar->free(dtemb); ar->free(emb); ar->free(z1); ar->free(a1); ar->free(dh); ar->free(dfilm);
  }

  // ---- single operator: dy / dx fp32 NCHW, dtemb [B][T] (IRB only)
  void module(const Tens& x1, const float* temb, const float* dy, float* dx, float* dtemb) {
    const llie_config& g = c->cfg;
    const Tens& out = tp->hlast;
    const Tens& x0 = tp->h0;
    const int T = g.time_embed_dim, F = c->film_rows;
    const size_t gy = alloc((size_t)B * out.H * out.W * out.C * es());
    if (!dry) chk(launch_nchw_to_nhwc(dt, dy, p(gy), nullptr, B, out.C, out.H * out.W, out.C, 0, s));
    gmap[out.off] = gy;
    const size_t dfilm = g.kind == LLIE_IRB ? alloc((size_t)B * F * 4) : 0;
    run_ops(dfilm, F);
    if (g.kind == LLIE_IRB) {
      const size_t ds = alloc((size_t)B * T * 4);
      film_bwd(dfilm, F, T, ds);
      if (!dry) chk(launch_silu_bwd(p<float>(ds), temb, dtemb, (int64_t)B * T, s));
      ar->free(ds); ar->free(dfilm);
    }
    const size_t g0 = take_grad(x0);
    if (!dry) chk(launch_nhwc_to_nchw(dt, p(g0), dx, B, x0.C, x0.H * x0.W, s, g.in_channels, 0));
    ar->free(g0);
    if (x1.valid) {
      const size_t g1 = take_grad(x1);
      if (!dry) chk(launch_nhwc_to_nchw(dt, p(g1), dx, B, x1.C, x1.H * x1.W, s, g.in_channels, x0.C));
      ar->free(g1);
    }
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
  if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
  for (int i = 0; i < kMaxBranches; ++i) {
    if (c->branch_stream[i]) (void)hipStreamDestroy(c->branch_stream[i]);
    if (c->branch_join[i]) (void)hipEventDestroy(c->branch_join[i]);
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  for (auto& r : c->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  if (c->blob) (void)hipFree(c->blob);
  delete c->train_arena;
  if (c->load_descs) (void)hipFree(c->load_descs);
  if (c->hash_partial) (void)hipFree(c->hash_partial);
  if (c->hash_state) (void)hipFree(c->hash_state);
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
    case PK_MAT:
      e = launch_cvt_rows(p.as_t ? c->dt : 0, src, dst, p.rows, p.cols, p.ld, p.col0, s);
      // transposed copies feed the backward pass, which the padded (unpinned) variants do not have
      if (e == hipSuccess && p.has_t && !c->padded) e = launch_cvt_rows_t(c->dt, src, c->blob + p.t_off, p.rows, p.cols, s);
      if (e == hipSuccess && p.has_f) e = launch_pack_expand(c->dt, src, c->blob + p.f_off, p.rows, p.cols, p.f_scale, s);
      break;
    case PK_CONV3:
      e = launch_repack_conv3x3(c->dt, src, dst, p.O, p.I, s, p.Op, p.Ip);
      if (e == hipSuccess && p.has_t && !c->padded) e = launch_repack_conv3x3_t(c->dt, src, c->blob + p.t_off, p.O, p.I, s);
      break;
    case PK_DW:
      e = launch_repack_dw(src, reinterpret_cast<float*>(dst), p.O, s, p.Op);
      if (e == hipSuccess && p.has_t) e = launch_repack_dw_flip(src, reinterpret_cast<float*>(c->blob + p.t_off), p.O, s, p.Op);
      break;
    case PK_INIT:
      e = launch_repack_init(src, reinterpret_cast<float*>(dst), p.O, p.I, s, p.Op);
      if (e == hipSuccess && c->dt != LLIE_F32) e = launch_repack_init_mfma(c->dt, src, c->blob + c->init_wp, p.O, p.I, s, p.Op);
      break;
    case PK_FINAL:
      e = launch_repack_final(src, reinterpret_cast<float*>(dst), p.O, p.I, s, p.Ip);
      if (e == hipSuccess && c->dt != LLIE_F32) e = launch_repack_final_mfma(c->dt, src, c->blob + c->fin_wp, p.O, p.I, s, p.Ip);
      break;
  }
  if (e != hipSuccess) { set_err("repack of '%s' failed: %s", key, hipGetErrorString(e)); return (int)e; }
  p.loaded = true;
  return LLIE_OK;
}

// Reload every parameter from `srcs[i]` (device fp32, llie_param_info order) -- what an optimiser step needs.  Everything
// goes through one kernel driven by a descriptor table that is rebuilt only when a source pointer changes.
// conditional != 0 (llie_refresh_params): the reload happens on the device only if the parameters' content hash differs
// from the one of the last load -- no host round trip, ~3 small launches when nothing changed.
static int load_all_impl(llie_ctx* c, const float* const* srcs, int n, llie_stream stream, int conditional) {
  if (!c || !srcs || n != (int)c->params.size()) return LLIE_ERR_ARG;
  if (!c->blob) { set_err("no HIP device"); return LLIE_ERR_NO_DEVICE; }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int i = 0; i < n; ++i)
    if (!srcs[i]) return LLIE_ERR_ARG;
  bool rebuild = !c->load_descs || (int)c->load_srcs.size() != n;
  for (int i = 0; !rebuild && i < n; ++i) rebuild = c->load_srcs[i] != srcs[i];
  if (rebuild) {
    std::vector<LoadDesc> d;
    for (int i = 0; i < n; ++i) {
      const Param& p = c->params[i];
      LoadDesc e{};
      e.src = srcs[i]; e.numel = p.numel; e.dst = (long long)p.off; e.dst_t = p.has_t ? (long long)p.t_off : -1;
      e.as_t = p.as_t ? 1 : 0; e.rows = p.rows; e.cols = p.cols; e.ld = p.ld; e.col0 = p.col0; e.O = p.O; e.I = p.I;
      e.Op = p.Op > 0 ? p.Op : p.O; e.Ip = p.Ip > 0 ? p.Ip : p.I;
      e.dst_f = p.has_f ? (long long)p.f_off : -1; e.fscale = p.f_scale;
      if (c->padded && p.kind != PK_DW) e.dst_t = -1;  // no backward pass for the padded variants (see llie_load_param)
      switch (p.kind) {
        case PK_F32: e.kind = 0; break;
        case PK_MAT: e.kind = 1; break;
        case PK_CONV3: e.kind = 2; break;
        case PK_DW: e.kind = 3; break;
        case PK_INIT: e.kind = 4; e.dst_t = c->dt != LLIE_F32 ? (long long)c->init_wp : -1; break;
        case PK_FINAL: e.kind = 5; e.dst_t = c->dt != LLIE_F32 ? (long long)c->fin_wp : -1; break;
      }
      d.push_back(e);
    }
    hipError_t e = hipSuccess;
    if (!c->load_descs) e = hipMalloc(reinterpret_cast<void**>(&c->load_descs), sizeof(LoadDesc) * c->params.size());
    if (e == hipSuccess && !c->hash_partial) e = hipMalloc(reinterpret_cast<void**>(&c->hash_partial), sizeof(unsigned long long) * 32 * c->params.size());
    if (e == hipSuccess && !c->hash_state) {
      e = hipMalloc(reinterpret_cast<void**>(&c->hash_state), 2 * sizeof(unsigned long long));
      if (e == hipSuccess) e = hipMemsetAsync(c->hash_state, 0, 2 * sizeof(unsigned long long), s);
    }
    // pageable host memory: the copy is staged before the call returns, so the vector may go out of scope
    if (e == hipSuccess) e = hipMemcpyAsync(c->load_descs, d.data(), sizeof(LoadDesc) * d.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { set_err("llie_load_all: %s", hipGetErrorString(e)); return (int)e; }
    c->load_srcs.assign(srcs, srcs + n);
    // the input / output convolutions' zero padding is written by their own repack kernels, once per source set
    for (int i = 0; i < n; ++i) {
      const PKind k = c->params[i].kind;
      if (k == PK_INIT || k == PK_FINAL) {
        const int rc = llie_load_param(c, c->params[i].key.c_str(), srcs[i], c->params[i].numel, stream);
        if (rc) return rc;
      }
    }
    conditional = 0;
  }
  hipError_t e = launch_params_hash(c->load_descs, n, c->hash_partial, c->hash_state, conditional ? 0 : 1, s);
  if (e == hipSuccess) e = launch_load_all(c->dt, c->load_descs, n, c->blob, s, c->hash_state);
  if (e != hipSuccess) { set_err("llie_load_all: %s", hipGetErrorString(e)); return (int)e; }
  for (int i = 0; i < n; ++i) c->params[i].loaded = true;
  return LLIE_OK;
}
int llie_load_all(llie_ctx* c, const float* const* srcs, int n, llie_stream stream) { return load_all_impl(c, srcs, n, stream, 0); }
int llie_refresh_params(llie_ctx* c, const float* const* srcs, int n, llie_stream stream) { return load_all_impl(c, srcs, n, stream, 1); }

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

// ---------------------------------------------------------------------------------------------
// Training: forward that keeps its activations + reverse pass over the tape.
// side stream + events of the backward pass (created on first use); dry runs only copy the flag
static int setup_async(llie_ctx* c, Back& b) {
  b.async = g_bwd_async != 0;
  if (!b.async || b.dry) return LLIE_OK;
  hipError_t e = hipSuccess;
  if (!c->side_stream) e = hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking);
  if (e == hipSuccess && !c->ev_fork) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess && !c->ev_join) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
  if (e != hipSuccess) { set_err("backward side stream: %s", hipGetErrorString(e)); return (int)e; }
  b.s2 = c->side_stream;
  return LLIE_OK;
}

int64_t llie_grad_numel(const llie_ctx* c) { return c ? c->grad_numel : LLIE_ERR_ARG; }
int64_t llie_param_grad_offset(const llie_ctx* c, int i) {
  if (!c || i < 0 || i >= (int)c->params.size()) return LLIE_ERR_ARG;
  return c->params[i].goff;
}

int64_t llie_train_workspace_bytes(llie_ctx* c, int batch, int height, int width) {
  if (!c || batch <= 0) return LLIE_ERR_ARG;
  if (c->padded) { set_err("the unpinned variants (tiny / base, zero-padded channels) are inference-only"); return LLIE_ERR_CONFIG; }
  if (c->cfg.kind == LLIE_UNET && c->cfg.image_size % 64) { set_err("training needs an image_size that is a multiple of 64 (inference: 32)"); return LLIE_ERR_SHAPE; }
  Arena ar((size_t)1 << 46);
  Tape tape;
  Run r{c, &ar, nullptr, nullptr, true, batch, c->dt};
  r.tape = &tape;
  Back b{c, &ar, nullptr, nullptr, true, batch, c->dt, &tape, nullptr};
  setup_async(c, b);
  if (c->cfg.kind == LLIE_UNET) {
    r.unet(nullptr, nullptr, nullptr, 0, nullptr);
    b.unet(nullptr);
  } else {
    if (shape_ok(c, height, width) != LLIE_OK) return LLIE_ERR_SHAPE;
    r.module(nullptr, nullptr, nullptr, height, width);
    b.module(r.mod_x1, nullptr, nullptr, nullptr, nullptr);
  }
  if (b.err != hipSuccess) { set_err("training plan is inconsistent"); return LLIE_ERR_ARG; }
  return (int64_t)ar.high;
}

int llie_unet_train_forward(llie_ctx* c, const float* lat, const float* cond, const int64_t* t, float* eps, int batch,
                            void* ws, int64_t ws_bytes, llie_stream stream) {
  if (!c || !lat || !cond || !t || !eps || !ws || batch <= 0 || c->cfg.kind != LLIE_UNET) return LLIE_ERR_ARG;
  if (c->padded) { set_err("the unpinned variants (tiny / base, zero-padded channels) are inference-only"); return LLIE_ERR_CONFIG; }
  if (c->cfg.image_size % 64) { set_err("training needs an image_size that is a multiple of 64 (inference: 32)"); return LLIE_ERR_SHAPE; }
  if (!c->blob) { set_err("no HIP device"); return LLIE_ERR_NO_DEVICE; }
  int rc = check_loaded(c);
  if (rc) return rc;
  const int64_t need = llie_train_workspace_bytes(c, batch, 0, 0);
  if (need < 0) return (int)need;
  if (need > ws_bytes) { set_err("workspace too small: need %lld, have %lld", (long long)need, (long long)ws_bytes); return LLIE_ERR_WORKSPACE; }
  delete c->train_arena;
  c->train_arena = new Arena((size_t)ws_bytes);
  c->tape.clear();
  Run r{c, c->train_arena, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<char*>(ws), false, batch, c->dt};
  r.tape = &c->tape;
  r.unet(lat, cond, t, 0, eps, nullptr);
  rc = finish_run(r, ws_bytes);
  if (rc) return rc;
  c->tape.valid = true;
  c->tape.ws = ws;
  return LLIE_OK;
}

int llie_unet_backward(llie_ctx* c, const float* d_eps, float* grads, int batch, void* ws, int64_t ws_bytes, llie_stream stream) {
  if (!c || !d_eps || !grads || !ws || c->cfg.kind != LLIE_UNET) return LLIE_ERR_ARG;
  if (!c->tape.valid || c->tape.ws != ws || c->tape.B != batch || !c->train_arena || (int64_t)c->train_arena->cap != ws_bytes) {
    set_err("llie_unet_backward needs the workspace of the preceding llie_unet_train_forward (same batch)");
    return LLIE_ERR_ARG;
  }
  Back b{c, c->train_arena, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<char*>(ws), false, batch, c->dt, &c->tape, grads};
  { const int rc = setup_async(c, b); if (rc) return rc; }
  b.unet(d_eps);
  if (c->train_arena->failed) { set_err("workspace too small for the backward pass"); return LLIE_ERR_WORKSPACE; }
  if (b.err != hipSuccess) { set_err("HIP error %d: %s", (int)b.err, hipGetErrorString(b.err)); return (int)b.err; }
  return LLIE_OK;
}

int llie_module_backward(llie_ctx* c, const float* x, const float* temb, const float* dy, float* dx, float* dtemb, float* grads,
                         int batch, int H, int W, void* ws, int64_t ws_bytes, llie_stream stream) {
  if (!c || !x || !dy || !dx || !grads || !ws || batch <= 0 || c->cfg.kind == LLIE_UNET || c->cfg.kind == LLIE_SE) return LLIE_ERR_ARG;
  if (c->cfg.kind == LLIE_IRB && (!temb || !dtemb)) return LLIE_ERR_ARG;
  if (!c->blob) { set_err("no HIP device"); return LLIE_ERR_NO_DEVICE; }
  int rc = check_loaded(c);
  if (rc) return rc;
  rc = shape_ok(c, H, W);
  if (rc) return rc;
  const int64_t need = llie_train_workspace_bytes(c, batch, H, W);
  if (need < 0) return (int)need;
  if (need > ws_bytes) { set_err("workspace too small: need %lld, have %lld", (long long)need, (long long)ws_bytes); return LLIE_ERR_WORKSPACE; }
  Arena ar((size_t)ws_bytes);
  Tape tape;
  Run r{c, &ar, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<char*>(ws), false, batch, c->dt};
  r.tape = &tape;
  r.module(x, temb, nullptr, H, W);
  rc = finish_run(r, ws_bytes);
  if (rc) return rc;
  Back b{c, &ar, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<char*>(ws), false, batch, c->dt, &tape, grads};
  { const int rc2 = setup_async(c, b); if (rc2) return rc2; }
  b.module(r.mod_x1, temb, dy, dx, dtemb);
  if (ar.failed) { set_err("workspace too small for the backward pass"); return LLIE_ERR_WORKSPACE; }
  if (b.err != hipSuccess) { set_err("HIP error %d: %s", (int)b.err, hipGetErrorString(b.err)); return (int)b.err; }
  return LLIE_OK;
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

int llie_add_noise(const float* x0, const float* noise, const int64_t* t, const float* acp, int table_len, float* out, int batch,
                   int64_t per, int velocity, llie_stream stream) {
  if (!x0 || !noise || !t || !acp || !out || batch <= 0 || per <= 0 || table_len <= 0) return LLIE_ERR_ARG;
  hipError_t e = launch_add_noise(x0, noise, t, acp, out, batch, per, velocity, table_len, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) { set_err("add_noise: %s", hipGetErrorString(e)); return (int)e; }
  return LLIE_OK;
}

// The launch sequence of LowLightDiffusion.enhance (low_light_diffusion.py:204-240): `steps` x
// (UNet forward, scheduler step).  `base` holds two latent ping-pong images and one eps image,
// followed by the UNet workspace.
// `step_batch`: images per step in the noise / inter / preds / timestep arrays (>= batch when this call handles a
// slice of a larger batch; 0 = batch)
static int enhance_sequence(llie_ctx* c, const float* low, const float* noise, const int64_t* t_dev,
                            const llie_step_coef* coefs, int steps, float* enhanced, float* inter, float* preds,
                            int batch, char* base, int64_t ws_bytes, llie_stream stream, int step_batch = 0) {
  const int S = c->cfg.image_size;
  const int64_t n = (int64_t)batch * 3 * S * S;
  if (step_batch <= 0) step_batch = batch;
  const int64_t sn = (int64_t)step_batch * 3 * S * S;  // elements between consecutive steps
  const size_t img = align_up((size_t)n * 4, 256);
  float* lat[2] = {reinterpret_cast<float*>(base), reinterpret_cast<float*>(base + img)};
  float* eps_ws = reinterpret_cast<float*>(base + 2 * img);
  void* uws = base + 3 * img;
  const int64_t uws_bytes = ws_bytes - (int64_t)(3 * img);
  const float* cur = noise;  // initial latents = first draw (low_light_diffusion.py:208-211)
  const bool fuse = c->dt != LLIE_F32;  // the MFMA output head applies the scheduler step in its epilogue
  for (int i = 0; i < steps; ++i) {
    const bool last = i == steps - 1;
    float* prev = inter ? inter + (size_t)i * sn : lat[i & 1];
    const float* nz = coefs[i].is_last ? nullptr : noise + (size_t)(i + 1) * sn;
    if (!coefs[i].is_last && i + 1 >= steps) return LLIE_ERR_ARG;  // a non-final step needs a noise draw
    int rc;
    if (fuse) {
      Run::FusedStep fs{StepCoef{coefs[i].sqrt_alpha_t, coefs[i].sqrt_beta_t, coefs[i].sqrt_alpha_prev, coefs[i].sqrt_beta_prev,
                                 coefs[i].is_last, coefs[i].v_prediction, coefs[i].clamp_x0},
                        nz, prev, last ? enhanced : nullptr};
      rc = unet_forward_impl(c, cur, low, t_dev + (size_t)i * step_batch, 1, preds ? preds + (size_t)i * sn : nullptr, &fs, batch,
                             uws, uws_bytes, stream);
      if (rc) return rc;
    } else {
      float* eps = preds ? preds + (size_t)i * sn : eps_ws;
      rc = llie_unet_forward(c, cur, low, t_dev + (size_t)i * step_batch, 1, eps, batch, uws, uws_bytes, stream);
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
  snprintf(tail, sizeof tail, "|%d|%d|%d|%d|%p|%lld|%d|%d", batch, steps, inter ? 1 : 0, preds ? 1 : 0, ws, (long long)ws_bytes,
           g_enhance_split, tune_epoch());
  key += tail;
  if (c->graphs.find(key) == c->graphs.end() && c->graphs.size() >= llie_ctx::kMaxGraphs) {  // evict the least recently used entry
    auto lru = c->graphs.begin();
    for (auto it = c->graphs.begin(); it != c->graphs.end(); ++it)
      if (it->second.used < lru->second.used) lru = it;
    if (lru->second.exec || lru->second.graph) (void)hipDeviceSynchronize();  // a replay of it may still be in flight (on any stream); evictions are rare
    if (lru->second.exec) (void)hipGraphExecDestroy(lru->second.exec);
    if (lru->second.graph) (void)hipGraphDestroy(lru->second.graph);
    c->graphs.erase(lru);
  }
  llie_ctx::GraphEntry& ge = c->graphs[key];
  ge.used = ++c->graph_clock;
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
    int rc = LLIE_OK;
    // The batch as g_enhance_split (default 2) concurrent branches of the graph: no operator mixes samples and every kernel
    // is bitwise batch-invariant, so the result is unchanged; memory-bound kernels of one branch overlap with the
    // latency / MFMA-bound ones and the launch boundaries of the others (llie_tune("enhance_split", 0 or 1): a single chain).
    int nbr = g_enhance_split < 2 ? 1 : (g_enhance_split > kMaxBranches ? kMaxBranches : g_enhance_split);
    while (nbr > 1 && batch / nbr < 8) --nbr;  // branches of fewer than 8 images lose more in kernel efficiency than they hide
    int hb[kMaxBranches];
    size_t woff[kMaxBranches];
    int64_t wsz[kMaxBranches];
    size_t wtot = 0;
    for (int i = 0; i < nbr; ++i) {
      hb[i] = batch / nbr + (i < batch % nbr ? 1 : 0);
      wsz[i] = llie_workspace_bytes(c, hb[i], 0, 0);
      if (wsz[i] <= 0) { nbr = 1; break; }
      woff[i] = wtot;
      wtot += align_up((size_t)wsz[i], 256);
    }
    if (nbr > 1 && (int64_t)wtot > seq_bytes) nbr = 1;
    if (nbr > 1) {
      hipError_t e2 = hipSuccess;
      if (!c->ev_fork) e2 = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
      for (int i = 1; i < nbr && e2 == hipSuccess; ++i) {
        if (!c->branch_stream[i]) e2 = hipStreamCreateWithFlags(&c->branch_stream[i], hipStreamNonBlocking);
        if (e2 == hipSuccess && !c->branch_join[i]) e2 = hipEventCreateWithFlags(&c->branch_join[i], hipEventDisableTiming);
      }
      if (e2 == hipSuccess) e2 = hipEventRecord(c->ev_fork, c->cap_stream);
      for (int i = 1; i < nbr && e2 == hipSuccess; ++i) e2 = hipStreamWaitEvent(c->branch_stream[i], c->ev_fork, 0);  // joins the capture
      if (e2 != hipSuccess) { hipGraph_t gd = nullptr; (void)hipStreamEndCapture(c->cap_stream, &gd); if (gd) (void)hipGraphDestroy(gd);
                              set_err("enhance split: %s", hipGetErrorString(e2)); return (int)e2; }
      size_t img0 = 0;  // first image of the branch
      for (int i = 0; i < nbr; ++i) {
        const size_t off = img0 * 3 * S * S;
        hipStream_t bs = i == 0 ? c->cap_stream : c->branch_stream[i];
        const int rci = enhance_sequence(c, s_low + off, s_noise + off, s_t + img0, coefs, steps, s_enh + off, s_inter ? s_inter + off : nullptr,
                                         s_preds ? s_preds + off : nullptr, hb[i], base + woff[i], wsz[i], reinterpret_cast<llie_stream>(bs), batch);
        if (rc == LLIE_OK) rc = rci;
        img0 += hb[i];
      }
      for (int i = 1; i < nbr; ++i) {
        e2 = hipEventRecord(c->branch_join[i], c->branch_stream[i]);
        if (e2 == hipSuccess) e2 = hipStreamWaitEvent(c->cap_stream, c->branch_join[i], 0);
        if (e2 != hipSuccess && rc == LLIE_OK) { set_err("enhance split join: %s", hipGetErrorString(e2)); rc = (int)e2; }
      }
    } else {
      rc = enhance_sequence(c, s_low, s_noise, s_t, coefs, steps, s_enh, s_inter, s_preds, batch, base, seq_bytes,
                            reinterpret_cast<llie_stream>(c->cap_stream));
    }
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

int llie_gram_stats(int dtype, const void* x0, int c0, const void* x1, int c1, const float* scale, const float* bias, int batch, int pixels,
                    float* part, float* gtot, unsigned int* tickets, llie_stream stream) {
  GramArgs a{};
  a.x0 = x0; a.x1 = x1; a.c0 = c0; a.c1 = c1; a.as1 = scale; a.ab1 = bias; a.part = part; a.gtot = gtot; a.tickets = tickets; a.B = batch; a.P = pixels;
  if (!gram_supported(dtype, c0 + c1, c0, pixels)) { set_err("gram_stats: K in {32, 64, 96}, 2-byte dtype, pixels a multiple of 512"); return LLIE_ERR_SHAPE; }
  hipError_t e = launch_gram_stats(dtype, a, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) { set_err("gram_stats: %s", hipGetErrorString(e)); return e == hipErrorInvalidValue ? LLIE_ERR_ARG : (int)e; }
  return LLIE_OK;
}
int64_t llie_gram_part_floats(int K, int pixels) { return (K == 32 || K == 64 || K == 96) && pixels > 0 && pixels % 512 == 0 ? (int64_t)gram_part_floats(K, pixels) : LLIE_ERR_ARG; }

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

int llie_pw_expand(int dtype, const llie_gemm_seg* segs, int nseg, const float* w32, void* wpack, void* out, float* stats,
                   int M, int N, int P, llie_stream stream) {
  if (!segs || nseg < 1 || nseg > 3 || !wpack || !out || !stats || dtype < 1 || dtype > 2) return LLIE_ERR_ARG;
  ExpandArgs x{};
  x.nseg = nseg;
  for (int i = 0; i < nseg; ++i) {
    x.seg[i] = GemmSeg{segs[i].ptr, segs[i].channels, segs[i].scale, segs[i].bias, segs[i].affine_ld, segs[i].act};
    x.K += segs[i].channels;
  }
  x.wf = wpack; x.out = out; x.stats = stats; x.M = M; x.N = N; x.P = P;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = !pw_expand_supported(dtype, x.seg, nseg, M, N, x.K, P) ? hipErrorInvalidValue
                 : (w32 ? launch_pack_expand(dtype, w32, wpack, N, x.K, 6.f, s) : hipSuccess);
  if (e == hipSuccess) e = launch_pw_expand(dtype, x, s);
  if (e == hipErrorInvalidValue) { set_err("pw_expand: shape outside the kernel contract"); return LLIE_ERR_SHAPE; }
  if (e != hipSuccess) { set_err("pw_expand: %s", hipGetErrorString(e)); return (int)e; }
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

// ---- the remaining kernel-level entry points of SURVEY.md 8b (GroupNorm finalize, dense 3x3, linear attention, SE MLP, FiLM)
static int kerr(const char* what, hipError_t e) {
  if (e == hipSuccess) return LLIE_OK;
  if (e == hipErrorInvalidValue) { set_err("%s: shape outside the kernel contract", what); return LLIE_ERR_SHAPE; }
  set_err("%s: %s", what, hipGetErrorString(e));
  return (int)e;
}
int llie_groupnorm_finalize(const float* slab0, int ntiles0, int ch0, const float* slab1, int ntiles1, int ch1, int groups, int pixels,
                            const float* gamma, const float* beta, const float* film, int64_t film_stride, float eps, float post_scale,
                            int batch, float* scale_out, float* shift_out, llie_stream stream) {
  if (!slab0 || !gamma || !beta || !scale_out || !shift_out || batch <= 0 || ch0 <= 0 || (slab1 && ch1 <= 0) || groups <= 0 || pixels <= 0 ||
      ntiles0 <= 0 || (slab1 && ntiles1 <= 0) || (ch0 + (slab1 ? ch1 : 0)) % groups)
    return LLIE_ERR_ARG;
  GnFinalizeArgs a{};
  a.src[0] = StatSrc{slab0, ntiles0, ch0};
  if (slab1) a.src[1] = StatSrc{slab1, ntiles1, ch1};
  a.C = ch0 + (slab1 ? ch1 : 0); a.groups = groups; a.P = pixels; a.gamma = gamma; a.beta = beta; a.film = film; a.film_stride = film_stride;
  a.eps = eps; a.as = scale_out; a.ab = shift_out; a.B = batch; a.post_scale = post_scale;
  return kerr("groupnorm_finalize", launch_gn_finalize(a, reinterpret_cast<hipStream_t>(stream)));
}
// GroupNorm-2 + FiLM affine of the recompute form from the Gram totals llie_gram_stats leaves (gram.hip: gram_finalize_kernel)
int llie_gram_finalize(int dtype, const float* gram_totals, const void* w_expand, int K, int pixels, const float* gamma, const float* beta,
                       const float* film, int64_t film_stride, float eps, float post_scale, int batch, float* scale_out, float* shift_out,
                       llie_stream stream) {
  if (!gram_totals || !w_expand || !gamma || !beta || !scale_out || !shift_out || batch <= 0 || pixels <= 0 || (K != 32 && K != 64 && K != 96) ||
      (dtype != 1 && dtype != 2))
    return LLIE_ERR_ARG;
  GramFinalizeArgs a{};
  a.gtot = gram_totals; a.w1 = w_expand; a.K = K; a.Chid = 4 * K; a.groups = 32; a.P = pixels; a.B = batch; a.gamma = gamma; a.beta = beta;
  a.film = film; a.film_stride = film_stride; a.eps = eps; a.as = scale_out; a.ab = shift_out; a.post_scale = post_scale;
  return kerr("gram_finalize", launch_gram_finalize(dtype, a, reinterpret_cast<hipStream_t>(stream)));
}
int llie_conv3x3(int dtype, int mode, const void* in, const void* w, const float* bias, void* out, float* stats, int batch, int Hi, int Wi,
                 int Cin, int Cout, llie_stream stream) {
  if (!in || !w || !out || dtype < 0 || dtype > 2 || (mode != 0 && mode != 1)) return LLIE_ERR_ARG;
  Conv3Args a{};
  a.in = in; a.w = w; a.bias = bias; a.out = out; a.stats = stats; a.B = batch; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.mode = mode;
  return kerr("conv3x3", launch_conv3x3(dtype, a, reinterpret_cast<hipStream_t>(stream)));
}
int llie_conv3x3_tiles(int Ho, int Wo) { return conv3x3_ntiles(Ho, Wo); }
int llie_linattn_splits(int N) { return linattn_nsplit(N); }
int llie_linattn(int dtype, const void* qkv, float* kv_scratch, void* out, int batch, int N, int heads, llie_stream stream) {
  if (!qkv || !kv_scratch || !out || dtype < 0 || dtype > 2 || batch <= 0 || N <= 0 || heads <= 0) return LLIE_ERR_ARG;
  AttnArgs a{};
  a.qkv = qkv; a.B = batch; a.N = N; a.heads = heads; a.kv = kv_scratch; a.out = out; a.nsplit = linattn_nsplit(N);
  hipError_t e = launch_linattn_kv(dtype, a, reinterpret_cast<hipStream_t>(stream));
  if (e == hipSuccess) e = launch_linattn_out(dtype, a, reinterpret_cast<hipStream_t>(stream));
  return kerr("linattn", e);
}
int llie_se_mlp(int dtype, const float* pool_sums, int pixels, const void* w1, const float* b1, const void* w2, const float* b2, float* mean_scratch,
                float* hidden_scratch, float* gate, int batch, int C, int Cs, llie_stream stream) {
  if (!pool_sums || !w1 || !b1 || !w2 || !b2 || !mean_scratch || !hidden_scratch || !gate || dtype < 0 || dtype > 2 || batch <= 0 || C <= 0 || Cs <= 0)
    return LLIE_ERR_ARG;
  SeArgs a{};
  a.pool = pool_sums; a.ntiles = 1; a.P = pixels; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.mean = mean_scratch; a.hid = hidden_scratch; a.gate = gate;
  a.B = batch; a.C = C; a.Cs = Cs;
  hipError_t e = launch_se_fc1(dtype, a, reinterpret_cast<hipStream_t>(stream));
  if (e == hipSuccess) e = launch_se_fc2(dtype, a, reinterpret_cast<hipStream_t>(stream));
  return kerr("se_mlp", e);
}
int llie_film(const float* silu_temb, const float* wf, const float* bf, float* film, int rows, int T, int F, llie_stream stream) {
  if (!silu_temb || !wf || !bf || !film || rows <= 0 || T <= 0 || F <= 0) return LLIE_ERR_ARG;
  FilmArgs a{};
  a.silu_temb = silu_temb; a.rows = rows; a.T = T; a.wf = wf; a.bf = bf; a.film = film; a.F = F;
  return kerr("film", launch_film(a, reinterpret_cast<hipStream_t>(stream)));
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

int llie_time_embed(llie_ctx* c, const int64_t* t, int rows, float* emb, float* temb, float* silu_temb, llie_stream stream) {
  if (!c || !t || !temb || !silu_temb || rows <= 0 || c->cfg.kind != LLIE_UNET) return LLIE_ERR_ARG;
  if (int rc = check_loaded(c)) return rc;
  const llie_config& g = c->cfg;
  TimeArgs ta{};
  ta.t = t; ta.rows = rows; ta.dim = g.base_channels; ta.T = g.time_embed_dim;
  ta.freqs = reinterpret_cast<const float*>(c->blob + c->freqs);
  ta.w1 = reinterpret_cast<const float*>(c->blob + c->t_w1); ta.b1 = reinterpret_cast<const float*>(c->blob + c->t_b1);
  ta.w3 = reinterpret_cast<const float*>(c->blob + c->t_w3); ta.b3 = reinterpret_cast<const float*>(c->blob + c->t_b3);
  ta.temb = temb; ta.silu_temb = silu_temb; ta.emb_out = emb;
  hipError_t e = launch_time_embed(ta, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) { set_err("time_embed: %s", hipGetErrorString(e)); return (int)e; }
  return LLIE_OK;
}

// ---- optimiser step (training): one object per parameter set, tables resident on the device
struct llie_optimizer {
  int device = 0;
  OptTensor* tensors = nullptr;
  OptChunk* chunks = nullptr;
  double* partial = nullptr;
  int count = 0, nchunks = 0;
  int64_t numel = 0;
};

int llie_optimizer_create(const llie_opt_tensor* tensors, int count, llie_optimizer** out) {
  if (!tensors || count <= 0 || !out) return LLIE_ERR_ARG;
  std::vector<OptTensor> tt((size_t)count);
  std::vector<OptChunk> cc;
  int64_t total = 0;
  for (int i = 0; i < count; ++i) {
    const llie_opt_tensor& t = tensors[i];
    if (!t.param || !t.exp_avg || !t.exp_avg_sq || t.numel <= 0 || t.grad_offset < 0 || t.numel > (int64_t)INT32_MAX) {
      set_err("optimizer_create: tensor %d: null pointer, empty tensor or negative gradient offset", i);
      return LLIE_ERR_ARG;
    }
    tt[(size_t)i] = OptTensor{t.param, t.exp_avg, t.exp_avg_sq, t.ema, (long long)t.grad_offset, (long long)t.numel};
    for (int64_t o = 0; o < t.numel; o += kOptChunk) cc.push_back(OptChunk{i, (int)o});
    total += t.numel;
  }
  if (cc.size() > (size_t)INT32_MAX) return LLIE_ERR_ARG;
  auto* o = new llie_optimizer();
  o->count = count;
  o->nchunks = (int)cc.size();
  o->numel = total;
  hipError_t e = hipGetDevice(&o->device);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&o->tensors), tt.size() * sizeof(OptTensor));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&o->chunks), cc.size() * sizeof(OptChunk));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&o->partial), cc.size() * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(o->tensors, tt.data(), tt.size() * sizeof(OptTensor), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(o->chunks, cc.data(), cc.size() * sizeof(OptChunk), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    set_err("optimizer_create: %s", hipGetErrorString(e));
    llie_optimizer_destroy(o);
    return e == hipErrorNoDevice ? LLIE_ERR_NO_DEVICE : (int)e;
  }
  *out = o;
  return LLIE_OK;
}

void llie_optimizer_destroy(llie_optimizer* o) {
  if (!o) return;
  if (o->tensors) (void)hipFree(o->tensors);
  if (o->chunks) (void)hipFree(o->chunks);
  if (o->partial) (void)hipFree(o->partial);
  delete o;
}

int64_t llie_optimizer_numel(const llie_optimizer* o) { return o ? o->numel : (int64_t)LLIE_ERR_ARG; }

int llie_optimizer_step(llie_optimizer* o, const float* grad_base, const llie_opt_hyper* h, float* stats3, llie_stream stream) {
  if (!o || !grad_base || !h || !stats3) return LLIE_ERR_ARG;
  OptStepArgs a{};
  a.tensors = o->tensors; a.chunks = o->chunks; a.nchunks = o->nchunks;
  a.gbase = grad_base; a.partial = o->partial; a.stats = stats3;
  a.lr = h->lr; a.beta1 = h->beta1; a.beta2 = h->beta2; a.eps = h->eps; a.weight_decay = h->weight_decay;
  a.max_grad_norm = h->max_grad_norm; a.ema_decay = h->ema_decay; a.grad_scale = h->grad_scale;
  a.step = h->step; a.skip_nonfinite = h->skip_nonfinite;
  hipError_t e = launch_optimizer_step(a, reinterpret_cast<hipStream_t>(stream));
  if (e == hipErrorInvalidValue) { set_err("optimizer_step: hyper-parameters outside their ranges (lr, eps, weight_decay >= 0; 0 <= beta < 1; ema_decay <= 1; step >= 1)"); return LLIE_ERR_ARG; }
  if (e != hipSuccess) { set_err("optimizer_step: %s", hipGetErrorString(e)); return (int)e; }
  return LLIE_OK;
}

int llie_copy_probe(const void* src, void* dst, int64_t bytes, llie_stream stream) {
  if (!src || !dst || bytes <= 0) return LLIE_ERR_ARG;
  hipError_t e = launch_copy_probe(src, dst, bytes, reinterpret_cast<hipStream_t>(stream));
  if (e == hipErrorInvalidValue) return LLIE_ERR_ARG;
  if (e != hipSuccess) { set_err("copy_probe: %s", hipGetErrorString(e)); return (int)e; }
  return LLIE_OK;
}

int llie_rw_probe(const void* src, void* dst, int64_t units, int reads, int writes, int nontemporal, llie_stream stream) {
  if (!src || !dst || units <= 0) return LLIE_ERR_ARG;
  hipError_t e = launch_rw_probe(src, dst, units, reads, writes, nontemporal, reinterpret_cast<hipStream_t>(stream));
  if (e == hipErrorInvalidValue) return LLIE_ERR_ARG;
  if (e != hipSuccess) { set_err("rw_probe: %s", hipGetErrorString(e)); return (int)e; }
  return LLIE_OK;
}

int llie_dwconv3x3_tiles(int H, int W) { return dwconv_ntiles(H, W); }
int llie_pw_gemm_tile_rows(int P) { return pw_gemm_tile_rows(P); }

int llie_tune(const char* knob, int value) {
  if (!knob) return LLIE_ERR_ARG;
  ++g_tune_epoch;
  if (!strcmp(knob, "gemm_bk")) { pw_gemm_force_bk(value); return LLIE_OK; }
  if (!strcmp(knob, "gemm_bk128")) { pw_gemm_bk128(value); return LLIE_OK; }
  if (!strcmp(knob, "skip_small")) { g_skip_small = value; return LLIE_OK; }
  if (!strcmp(knob, "se_mfma")) { g_se_mfma = value; return LLIE_OK; }
  if (!strcmp(knob, "nt_min_mb")) { g_nt_min_mb = value; return LLIE_OK; }
  if (!strcmp(knob, "nt_mask")) { g_nt_mask = value; return LLIE_OK; }
  if (!strcmp(knob, "ztot")) { g_ztot = value; return LLIE_OK; }
  if (!strcmp(knob, "gram")) { g_gram = value; return LLIE_OK; }
  if (!strcmp(knob, "irbx")) { g_use_irbx = value != 0; return LLIE_OK; }
  if (!strcmp(knob, "irbx_dbuf")) { irbx_tune(value, 0); return LLIE_OK; }
  if (!strcmp(knob, "irbx_tiles")) { irbx_tune(-1, value); return LLIE_OK; }
  if (!strcmp(knob, "irbx_stamp")) { irbx_stamp(value); return LLIE_OK; }
  if (!strcmp(knob, "irbx_mask")) { g_irbx_mask = value; return LLIE_OK; }
  if (!strcmp(knob, "irbx_ablate")) { irbx_ablate(value); return LLIE_OK; }
  if (!strcmp(knob, "irbx_dwv")) { irbx_dwv(value); return LLIE_OK; }
  if (!strcmp(knob, "irbx_grid")) { irbx_grid(0, value); return LLIE_OK; }
  if (!strcmp(knob, "irbx_grid2")) { irbx_grid(2, value); return LLIE_OK; }
  if (!strcmp(knob, "irbx_grid4")) { irbx_grid(4, value); return LLIE_OK; }
  if (!strcmp(knob, "irbx_grid6")) { irbx_grid(6, value); return LLIE_OK; }
  if (!strcmp(knob, "irbx_var")) { irbx_var(value); return LLIE_OK; }
  if (!strcmp(knob, "conv_stamp")) { conv3x3_stamp(value); return LLIE_OK; }
  if (!strcmp(knob, "gemm_stamp")) { pw_gemm_stamp(value); return LLIE_OK; }
  if (!strcmp(knob, "pwx")) { pw_expand_enable(value); return LLIE_OK; }
  if (!strcmp(knob, "pwx_ablate")) { pw_expand_debug(value, -1); return LLIE_OK; }
  if (!strcmp(knob, "pwx_stamp")) { pw_expand_debug(-1, value); return LLIE_OK; }
  if (!strcmp(knob, "pwx_nbw")) { pw_expand_debug(-1, -1, value); return LLIE_OK; }
  if (!strcmp(knob, "gemm_ablate")) { pw_gemm_debug(value); return LLIE_OK; }
  if (!strcmp(knob, "dw_ablate")) { dwconv_debug(value); return LLIE_OK; }
  if (!strcmp(knob, "dw_swap")) { dwconv_swap(value); return LLIE_OK; }
  if (!strcmp(knob, "bwd_async")) { g_bwd_async = value; return LLIE_OK; }
  if (!strcmp(knob, "enhance_split")) { g_enhance_split = value; return LLIE_OK; }
  if (!strcmp(knob, "wgrad_target")) { wgrad_set_target(value); return LLIE_OK; }
  return LLIE_ERR_ARG;
}

int llie_debug_gemm_stamps(double* out3) {
  if (!out3) return LLIE_ERR_ARG;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = pw_gemm_stamp_fetch(out3);
  return e == hipSuccess ? LLIE_OK : LLIE_ERR_ARG;
}
int llie_debug_pwx_stamps(double* out4) {
  if (!out4) return LLIE_ERR_ARG;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = pw_expand_stamp_fetch(out4);
  return e == hipSuccess ? LLIE_OK : LLIE_ERR_ARG;
}

// diagnostic: mean per-wave cycles of the last stamped up-sampling conv launch (llie_tune("conv_stamp", 1)); synchronises
int llie_debug_conv_stamps(double* out8) {
  if (!out8) return LLIE_ERR_ARG;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = conv3x3_stamp_fetch(out8);
  return e == hipSuccess ? LLIE_OK : LLIE_ERR_ARG;
}

// number of entries in the context's hipGraph cache (bounded by kMaxGraphs, least recently used evicted)
int llie_graph_cache_entries(const llie_ctx* c) { return c ? (int)c->graphs.size() : LLIE_ERR_ARG; }

// diagnostic: mean per-wave cycles of the last stamped expand_dw launch (llie_tune("irbx_stamp", 1)); synchronises
int llie_debug_irbx_stamps(double* out10) {
  if (!out10) return LLIE_ERR_ARG;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = irbx_stamp_fetch(out10);
  return e == hipSuccess ? LLIE_OK : LLIE_ERR_ARG;
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

// Every recorded launch, in launch order: "class\tkernel\ttag\tms\talgorithmic_bytes\n" (tools/gpu_layers.py).
int llie_profile_dump(llie_ctx* c, char* buf, size_t cap) {
  if (!c || !buf || cap < 2) return LLIE_ERR_ARG;
  c->prof_mask = 0;
  std::string out;
  char line[640];
  for (auto& r : c->prof) {
    hipError_t e = hipEventSynchronize(r.e1);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, r.e0, r.e1);
    if (e != hipSuccess) { set_err("profile: %s", hipGetErrorString(e)); return (int)e; }
    snprintf(line, sizeof line, "%d\t%s\t%s\t%.6f\t%lld\n", r.cls, r.name ? r.name : "?", r.tag, t, (long long)r.bytes);
    out += line;
  }
  if (out.size() + 1 > cap) { set_err("profile dump buffer too small"); return LLIE_ERR_ARG; }
  memcpy(buf, out.c_str(), out.size() + 1);
  return LLIE_OK;
}

// SURVEY.md 8d byte model: IRB (2Cin + 4Chid + Cout)P, attention 6CP, dense 3x3 Cin*Pin + Cout*Pout,
// final C0*P + 3P, LCM step 12P fp32; activations at the compute dtype; weights once.
// engine != 0: what the engine's own kernel selection has to move -- blocks that run in the recompute form (irbx.hip)
// read x three times and never store h1: (3Cin + 2Chid + Cout) P (SURVEY.md 8d "recompute variant").  x0c = channels of
// the first input segment of the first block (virtual concat), 0 = none.
static void count_blocks(const llie_ctx* c, const std::vector<Block>& bl, int64_t P, int64_t& elems, int64_t& flops, int engine = 0,
                         int x0c = 0) {
  bool first = true;
  for (const Block& b : bl) {
    if (b.kind == 0) {
      const IrbW& w = c->irbs[b.idx];
      const int S = (int)std::lround(std::sqrt((double)P));
      const bool fx = engine && g_use_irbx && w.hid == w.hid_r && w.cin == w.cin_r &&
                      irbx_supported(c->dt, w.cin, (first && x0c) ? x0c : w.cin, w.hid, S, S);
      first = false;
      if (fx) elems += (3LL * w.cin + 2LL * w.hid + w.cout) * P;
      else elems += (2LL * w.cin + 4LL * w.hid + w.cout) * P;
      flops += 2LL * P * ((int64_t)w.cin * w.hid + 9LL * w.hid + (int64_t)w.hid * w.cout + (w.skip ? (int64_t)w.cin * w.cout : 0));
    } else {
      const AttnW& w = c->attns[b.idx];
      elems += 6LL * w.c * P;
      flops += 2LL * P * ((int64_t)w.c * 3 * w.inner + (int64_t)w.inner * w.c + 2LL * w.inner * 32);
    }
  }
}
static void model_counts(const llie_ctx* c, int64_t& elems, int64_t& flops, int engine = 0) {
  elems = flops = 0;
  if (c->cfg.kind != LLIE_UNET) return;
  const int S = c->cfg.image_size;
  int64_t P = (int64_t)S * S;
  const std::vector<int>& ch = c->channels;
  elems += (int64_t)c->cfg.in_channels * P + ch[0] * P;
  flops += 2LL * P * 9 * c->cfg.in_channels * ch[0];
  for (int l = 0; l < 4; ++l) {
    count_blocks(c, c->enc[l], P, elems, flops, engine);
    if (l < 3) {
      elems += ch[l] * P + ch[l] * (P / 4);
      flops += 2LL * (P / 4) * 9 * ch[l] * ch[l];
      P /= 4;
    }
  }
  count_blocks(c, c->mid, P, elems, flops, engine);
  for (int l = 0; l < 4; ++l) {
    if (l > 0) {
      const int cc = ch[4 - l];
      elems += (int64_t)cc * P + (int64_t)cc * P * 4;
      flops += 2LL * (P * 4) * 9 * cc * cc;
      P *= 4;
    }
    count_blocks(c, c->dec[l], P, elems, flops, engine, l == 0 ? ch[3] : ch[4 - l]);
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
int64_t llie_path_bytes(llie_ctx* c, int batch) {
  if (!c) return LLIE_ERR_ARG;
  int64_t elems, flops;
  model_counts(c, elems, flops, 1);
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
