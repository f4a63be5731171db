// C-ABI glue of libsnet_hip.so: error state, compiled-shape registry, conv plans, scratch.
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#include <dlfcn.h>

#include "snet_common.h"

namespace snet {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

static std::vector<const ConvKernels *> &registry() {
  static std::vector<const ConvKernels *> r;
  return r;
}
void register_conv(const ConvKernels *k) { registry().push_back(k); }

static std::vector<const FusedKernels *> &fused_registry() {
  static std::vector<const FusedKernels *> r;
  return r;
}
void register_fused(const FusedKernels *k) { fused_registry().push_back(k); }
const FusedKernels *find_fused(const char *tag) {
  for (const FusedKernels *k : fused_registry())
    if (std::string(k->tag) == tag) return k;
  return nullptr;
}

// Small scratch for deterministic two-stage reductions, per host thread and device (grown on
// demand, never shrunk).  Per-thread because the partial-sum kernel and its final-sum kernel are
// two launches: two host threads sharing one stream must not interleave on one buffer.
double *reduce_scratch(int64_t n_doubles, hipStream_t st) {
  (void)st;
  static thread_local std::map<int, std::pair<double *, int64_t>> bufs;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  auto &b = bufs[dev];
  if (b.second < n_doubles) {
    if (b.first) (void)hipFree(b.first);
    const int64_t n = n_doubles < 4096 ? 4096 : n_doubles;
    if (hipMalloc(&b.first, sizeof(double) * n) != hipSuccess) {
      b = {nullptr, 0};
      return nullptr;
    }
    b.second = n;
  }
  return b.first;
}

}  // namespace snet

struct snet_conv_plan {
  const snet::ConvKernels *k;
};
struct snet_fused_plan {
  const snet::FusedKernels *k;
  int terms;
  void *slabs;    // device: W2 as pre-split MFMA fragments in the FORWARD kernel's sub-step order
  void *slabs_b;  // the same in the REVERSE kernel's order, then the hidden-layer tail
  snet::MlpHidden hidden;  // w0 == nullptr: no tail (g_h2 is always written)
  int32_t exps[3];         // mode 4: power-of-two scales of W2 / W1 / W0 in the fragment stream
};

extern "C" {

int snet_abi_version(void) { return SNET_ABI_VERSION; }
const char *snet_last_error(void) { return snet::g_err.c_str(); }
int snet_conv_num_shapes(void) { return (int)snet::registry().size(); }
const char *snet_conv_shape_tag(int i) {
  auto &r = snet::registry();
  return (i >= 0 && i < (int)r.size()) ? r[i]->tag : nullptr;
}

int snet_conv_plan_create(const char *tag, snet_conv_plan **plan) {
  SNET_REQUIRE(tag != nullptr && plan != nullptr, "snet_conv_plan_create: null argument");
  for (const snet::ConvKernels *k : snet::registry()) {
    if (std::string(k->tag) == tag) {
      *plan = new snet_conv_plan{k};
      return 0;
    }
  }
  snet::set_error(std::string("snet_conv_plan_create: tensor-product shape '") + tag +
                  "' is not compiled into libsnet_hip.so (rebuild with the model's config: "
                  "sevennet_amd.build.build(extra_configs=[config]), or register it in sevennet_amd/shapes.py)");
  return 3;
}
void snet_conv_plan_destroy(snet_conv_plan *plan) { delete plan; }
int snet_conv_register_library(const char *path) {
  SNET_REQUIRE(path != nullptr, "snet_conv_register_library: null path");
  const size_t before = snet::registry().size() + snet::fused_registry().size();
  // the library's static registrars (generated conv_<tag>.hip / convf_<tag>.hip) run inside dlopen
  void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (h == nullptr) {
    const char *why = dlerror();
    snet::set_error(std::string("snet_conv_register_library: dlopen(") + path + ") failed: " + (why ? why : "?"));
    return 2;
  }
  if (snet::registry().size() + snet::fused_registry().size() == before) {
    snet::set_error(std::string("snet_conv_register_library: ") + path + " registered no tensor-product shape");
    return 3;
  }
  return 0;
}
int snet_conv_plan_dims(const snet_conv_plan *plan, int32_t *dx, int32_t *dout, int32_t *nsh, int32_t *wn) {
  SNET_REQUIRE(plan != nullptr, "snet_conv_plan_dims: null plan");
  if (dx) *dx = plan->k->dx;
  if (dout) *dout = plan->k->dout;
  if (nsh) *nsh = plan->k->nsh;
  if (wn) *wn = plan->k->wn;
  return 0;
}

int snet_conv_fwd(const snet_conv_plan *plan, const float *x, const float *sh, const float *w, const int32_t *w_row,
                  const int32_t *row_ptr, const int32_t *src, int64_t n_dst, float scale, float *out, void *stream) {
  SNET_REQUIRE(plan != nullptr, "snet_conv_fwd: null plan");
  SNET_REQUIRE(n_dst < (1ll << 31), "snet_conv_fwd: too many nodes");
  if (n_dst <= 0) return 0;
  plan->k->fwd(x, sh, w, w_row, row_ptr, src, n_dst, scale, out, static_cast<hipStream_t>(stream));
  SNET_CHECK_LAUNCH("snet_conv_fwd");
  return 0;
}
int snet_conv_plan_transposed(const snet_conv_plan *plan, char *tag, float *col_scale, int32_t *dead, int32_t dead_capacity,
                              int32_t *n_dead) {
  SNET_REQUIRE(plan != nullptr && tag != nullptr, "snet_conv_plan_transposed: null argument");
  const snet::ConvKernels *k = plan->k;
  tag[0] = 0;
  if (n_dead) *n_dead = 0;
  if (k->t_tag == nullptr) return 0;  // a non-scalar output: no transposed form
  SNET_REQUIRE(dead == nullptr || dead_capacity >= 2 * k->t_ndead, "snet_conv_plan_transposed: dead[] too small");
  snprintf(tag, 13, "%s", k->t_tag);
  if (col_scale) memcpy(col_scale, k->t_col_scale, (size_t)k->wn * 4);
  if (dead) memcpy(dead, k->t_dead, (size_t)k->t_ndead * 8);
  if (n_dead) *n_dead = k->t_ndead;
  return 0;
}

int snet_conv_fused_available(const snet_conv_plan *plan) {
  return plan != nullptr && snet::find_fused(plan->k->tag) != nullptr;
}


int snet_fused_plan_create(const snet_conv_plan *plan, const snet_mlp_plan *mlp, int32_t terms, snet_fused_plan **out) {
  SNET_REQUIRE(plan != nullptr && mlp != nullptr && out != nullptr, "snet_fused_plan_create: null argument");
  SNET_REQUIRE(terms >= 1 && terms <= 4,
               "snet_fused_plan_create: terms must be 1 (bf16), 2 (bf16x3), 3 (bf16x6) or 4 (f16x3: two fp16 terms)");
  const snet::FusedKernels *k = snet::find_fused(plan->k->tag);
  SNET_REQUIRE(k != nullptr, "snet_fused_plan_create: this tensor-product shape has no fused kernels (channel "
                             "multiplicities must be multiples of 16)");
  SNET_REQUIRE(snet::mlp_plan_wn(mlp) == k->wn && snet::mlp_plan_w2_host(mlp) != nullptr,
               "snet_fused_plan_create: the radial-MLP plan does not match the shape's weight_numel");
  void *dev = nullptr;
  // the reverse kernels can run the MLP's two hidden layers backwards themselves when the basis fits one
  // 16-row operand tile and its rows are whole float4s
  snet::MlpHidden hid = snet::mlp_plan_hidden(mlp);
  if (hid.w0 != nullptr && !(hid.nb <= 16 && hid.nb % 4 == 0 && (hid.act == 0 || hid.act == 1))) hid.w0 = nullptr;   // (the tail's fast activation pair: silu / tanh)
  int32_t exps[3] = {0, 0, 0}, exps_f[3];
  void *dev_b = nullptr;
  if (snet::pack_fused_slabs(snet::mlp_plan_w2_host(mlp), k->wn, k->n_sub, k->sub_cols, terms, nullptr, &dev, exps_f) != 0 ||
      snet::pack_fused_slabs(snet::mlp_plan_w2_host(mlp), k->wn, k->n_sub_b, k->sub_cols_b, terms,
                             hid.w0 ? &hid : nullptr, &dev_b, exps) != 0) {
    if (dev) (void)hipFree(dev);
    snet::set_error("snet_fused_plan_create: device allocation / upload of the W2 fragment stream failed");
    return 1;
  }
  *out = new snet_fused_plan{k, terms, dev, dev_b, hid, {exps[0], exps[1], exps[2]}};
  return 0;
}
int snet_fused_plan_tile_mode(const snet_fused_plan *p) { return p ? p->k->tile_mode : 0; }
void snet_fused_plan_destroy(snet_fused_plan *p) {
  if (!p) return;
  if (p->slabs) (void)hipFree(p->slabs);
  if (p->slabs_b) (void)hipFree(p->slabs_b);
  delete p;
}

int snet_conv_fwd_fused(const snet_fused_plan *fp, const float *x, const float *sh, const float *h2,
                        const int32_t *w_row, const int32_t *row_ptr, const int32_t *src, int64_t n_dst, float scale,
                        float *out, void *stream) {
  SNET_REQUIRE(fp != nullptr, "snet_conv_fwd_fused: null plan");
  SNET_REQUIRE(n_dst < (1ll << 31) / 4, "snet_conv_fwd_fused: too many nodes");
  if (n_dst <= 0) return 0;
  fp->k->fwd(fp->terms, x, sh, h2, w_row, row_ptr, src, n_dst, fp->slabs, scale, out, fp->exps[0],
             static_cast<hipStream_t>(stream));
  SNET_CHECK_LAUNCH("snet_conv_fwd_fused");
  return 0;
}
int snet_fused_plan_gxe_chunks(const snet_fused_plan *fp, int32_t *chunk_pos, int32_t capacity) {
  SNET_REQUIRE(fp != nullptr && chunk_pos != nullptr, "snet_fused_plan_gxe_chunks: null argument");
  const int n = fp->k->dx / 16;
  SNET_REQUIRE(capacity >= n, "snet_fused_plan_gxe_chunks: capacity below dx / 16");
  for (int i = 0; i < n; ++i) chunk_pos[i] = fp->k->gxe_chunk[i];
  return 0;
}
int snet_fused_plan_has_mlp_tail(const snet_fused_plan *fp) { return fp != nullptr && fp->hidden.w0 != nullptr; }

static int conv_bwd_fused_impl(const char *who, const snet_fused_plan *fp, const float *x, const float *sh, const float *dsh,
                               const float *h2, const int32_t *w_row, const int32_t *row_ptr, const int32_t *src,
                               const int32_t *tile_ptr, const int32_t *tile_node, int64_t n_tiles, float scale, const float *g_out,
                               float *g_xe, float *g_h2, const float *emb, float *g_emb, float *g_vec, float *g_sh,
                               const float *x_rowmax, const float *g_rowmax, void *stream) {
  SNET_REQUIRE(fp != nullptr, std::string(who) + ": null plan");
  SNET_REQUIRE(fp->terms != 4 || (x_rowmax != nullptr && g_rowmax != nullptr),
               std::string(who) + ": terms = 4 (fp16 operands) needs x_rowmax and g_rowmax (snet_row_absmax of x and g_out)");
  SNET_REQUIRE(n_tiles < (1ll << 31), std::string(who) + ": too many tiles");
  if (n_tiles <= 0) return 0;
  SNET_REQUIRE(tile_ptr != nullptr && tile_node != nullptr, std::string(who) + ": null tile list");
  SNET_REQUIRE((g_h2 != nullptr) != (g_emb != nullptr), std::string(who) + ": exactly one of g_h2 / g_emb is the output");
  SNET_REQUIRE(g_emb == nullptr || (fp->hidden.w0 != nullptr && emb != nullptr),
               std::string(who) + ": g_emb needs emb and a plan with the hidden-layer tail (snet_fused_plan_has_mlp_tail)");
  SNET_REQUIRE((dsh != nullptr) == (g_vec != nullptr), std::string(who) + ": dsh and g_vec go together");
  SNET_REQUIRE(g_vec != nullptr || g_sh != nullptr, std::string(who) + ": no output for the harmonics' gradient (g_vec or g_sh)");
  const snet::FusedTail tail{emb, g_emb, fp->hidden.nb, fp->hidden.act, fp->hidden.cst, fp->exps[0], fp->exps[1], fp->exps[2],
                             x_rowmax, g_rowmax, g_sh};
  fp->k->bwd(fp->terms, x, sh, dsh, h2, w_row, row_ptr, src, tile_ptr, tile_node, n_tiles, fp->slabs_b, scale, g_out,
             g_xe, g_h2, g_vec, tail, static_cast<hipStream_t>(stream));
  SNET_CHECK_LAUNCH(who);
  return 0;
}
int snet_conv_bwd_fused(const snet_fused_plan *fp, const float *x, const float *sh, const float *dsh, const float *h2,
                        const int32_t *w_row, const int32_t *row_ptr, const int32_t *src, const int32_t *tile_ptr,
                        const int32_t *tile_node, int64_t n_tiles, float scale, const float *g_out, float *g_xe,
                        float *g_h2, const float *emb, float *g_emb, float *g_vec, const float *x_rowmax,
                        const float *g_rowmax, void *stream) {
  return conv_bwd_fused_impl("snet_conv_bwd_fused", fp, x, sh, dsh, h2, w_row, row_ptr, src, tile_ptr, tile_node, n_tiles, scale, g_out,
                             g_xe, g_h2, emb, g_emb, g_vec, nullptr, x_rowmax, g_rowmax, stream);
}
int snet_conv_bwd_fused_sh(const snet_fused_plan *fp, const float *x, const float *sh, const float *h2, const int32_t *w_row,
                           const int32_t *row_ptr, const int32_t *src, const int32_t *tile_ptr, const int32_t *tile_node,
                           int64_t n_tiles, float scale, const float *g_out, float *g_xe, float *g_h2, const float *emb,
                           float *g_emb, float *g_sh, const float *x_rowmax, const float *g_rowmax, void *stream) {
  SNET_REQUIRE(g_sh != nullptr, "snet_conv_bwd_fused_sh: null g_sh");
  return conv_bwd_fused_impl("snet_conv_bwd_fused_sh", fp, x, sh, nullptr, h2, w_row, row_ptr, src, tile_ptr, tile_node, n_tiles, scale,
                             g_out, g_xe, g_h2, emb, g_emb, nullptr, g_sh, x_rowmax, g_rowmax, stream);
}
int snet_conv_bwd_edge(const snet_conv_plan *plan, const float *x, const float *sh, const float *w,
                       const int32_t *w_row, const int32_t *row_ptr, const int32_t *src, int64_t n_dst, float scale, const float *g_out,
                       float *g_w, float *g_xe, float *g_sh, void *stream) {
  SNET_REQUIRE(plan != nullptr, "snet_conv_bwd_edge: null plan");
  SNET_REQUIRE(n_dst < (1ll << 31), "snet_conv_bwd_edge: too many nodes");
  if (n_dst <= 0) return 0;
  plan->k->bwd_edge(x, sh, w, w_row, row_ptr, src, n_dst, scale, g_out, g_w, g_xe, g_sh, static_cast<hipStream_t>(stream));
  SNET_CHECK_LAUNCH("snet_conv_bwd_edge");
  return 0;
}
int snet_conv_bwd_edge_vec(const snet_conv_plan *plan, const float *x, const float *sh, const float *dsh,
                           const float *w, const int32_t *w_row, const int32_t *row_ptr, const int32_t *src, int64_t n_dst, float scale,
                           const float *g_out, float *g_w, float *g_xe, float *g_vec, void *stream) {
  SNET_REQUIRE(plan != nullptr, "snet_conv_bwd_edge_vec: null plan");
  SNET_REQUIRE(n_dst < (1ll << 31), "snet_conv_bwd_edge_vec: too many nodes");
  if (n_dst <= 0) return 0;
  plan->k->bwd_edge_vec(x, sh, dsh, w, w_row, row_ptr, src, n_dst, scale, g_out, g_w, g_xe, g_vec,
                        static_cast<hipStream_t>(stream));
  SNET_CHECK_LAUNCH("snet_conv_bwd_edge_vec");
  return 0;
}
int snet_conv_bwd_node(const snet_conv_plan *plan, const float *sh, const float *w, const int32_t *w_row,
                       const int32_t *col_ptr,
                       const int32_t *eperm, const int32_t *dst, int64_t n_src, float scale, const float *g_out,
                       float *g_x, void *stream) {
  SNET_REQUIRE(plan != nullptr, "snet_conv_bwd_node: null plan");
  SNET_REQUIRE(n_src < (1ll << 31), "snet_conv_bwd_node: too many nodes");
  if (n_src <= 0) return 0;
  plan->k->bwd_node(sh, w, w_row, col_ptr, eperm, dst, n_src, scale, g_out, g_x, static_cast<hipStream_t>(stream));
  SNET_CHECK_LAUNCH("snet_conv_bwd_node");
  return 0;
}

}  // extern "C"
