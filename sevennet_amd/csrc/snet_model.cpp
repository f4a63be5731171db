// Native whole-model sequencer: loads a `.snet` model file (sevennet_amd/model_file.py) and runs one
// energy + force evaluation entirely through this library's kernels -- the engine a C++ host (the
// LAMMPS pair styles, sevenn/pair_e3gnn/pair_e3gnn.cpp:74-289 and pair_e3gnn_parallel.cpp:194-528)
// calls instead of `model.forward` + `torch::autograd::grad`.  Same op sequence as the Python host
// (sevennet_amd/engine.py), no torch, no Python.  Ghost-feature exchange is left to the host through
// two callbacks invoked at the reference's exchange points (pair_e3gnn_parallel.cpp:369,435).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <string>
#include <vector>

#include "snet_common.h"

namespace {

struct Reader {
  const unsigned char *p, *end;
  bool ok = true;
  void get(void *dst, size_t n) {
    if (!ok || (size_t)(end - p) < n) {
      ok = false;
      return;
    }
    memcpy(dst, p, n);
    p += n;
  }
  int32_t i32() {
    int32_t v = 0;
    get(&v, 4);
    return v;
  }
  float f32() {
    float v = 0;
    get(&v, 4);
    return v;
  }
  double f64() {
    double v = 0;
    get(&v, 8);
    return v;
  }
  std::vector<double> darr(size_t n) {
    if (!ok || n > (size_t)(end - p) / 8) {
      ok = false;
      return {};
    }
    std::vector<double> v(n);
    if (n) get(v.data(), 8 * n);
    return v;
  }
  std::vector<float> farr(size_t n) {
    // bounds first: a corrupt count must not reach the allocator (bad_alloc / length_error would cross the C ABI)
    if (!ok || n > (size_t)(end - p) / 4) {
      ok = false;
      return {};
    }
    std::vector<float> v(n);
    if (n) get(v.data(), 4 * n);
    return v;
  }
};

struct Block {
  int l, in_off, mul_in, out_off, mul_out, species, accumulate;
  void *W = nullptr, *WT = nullptr;  // device split-packed fragments of [K,N] and [N,K] (snet_gemm_split_pack)
};
struct Linear {
  int dim_in = 0, dim_out = 0, n_species = 0;
  std::vector<Block> blocks;
  std::vector<std::pair<int, int>> zero, zero_in;  // output / input column ranges no block touches
  struct Group {
    int species;
    std::vector<snet_gemm_desc> descs;
  };
  std::vector<Group> fwd, rev;  // launch plans: per-irrep GEMMs with distinct targets share a launch
  float *bias = nullptr;        // device [dim_out]: constant row bias of a multi-modal linear, or null
  float t_norm = 0.f;           // largest row norm of the transposed map: |(L^T g)[k]| <= t_norm ||g||_2
  bool present() const { return dim_out > 0; }
};

// Same grouping rule as the Python host (engine.py::_Linear._plan): a block joins the previous launch
// when it has the same row list and a target no member of that launch writes; a target written by an
// earlier launch with the same row list accumulates.
void plan_groups(Linear &L, bool transpose) {
  auto &groups = transpose ? L.rev : L.fwd;
  std::vector<std::pair<int, int>> written;
  std::vector<std::vector<int>> targets;
  for (const Block &b : L.blocks) {
    const int tgt = transpose ? b.in_off : b.out_off;
    bool acc = false;
    for (auto &w : written) acc |= (w.first == tgt && w.second == b.species);
    snet_gemm_desc d;
    d.B = nullptr;
    d.B_split = transpose ? b.WT : b.W;
    d.a_off = transpose ? b.out_off : b.in_off;
    d.c_off = tgt;
    d.d = 2 * b.l + 1;
    d.K = transpose ? b.mul_out : b.mul_in;
    d.N = transpose ? b.mul_in : b.mul_out;
    d.accumulate = acc;
    bool placed = false;
    if (!groups.empty() && groups.back().species == b.species && groups.back().descs.size() < SNET_MAX_GEMM_GROUP) {
      bool clash = false;
      for (int t : targets.back()) clash |= (t == tgt);
      if (!clash) {
        groups.back().descs.push_back(d);
        targets.back().push_back(tgt);
        placed = true;
      }
    }
    if (!placed) {
      groups.push_back({b.species, {d}});
      targets.push_back({tgt});
    }
    written.push_back({tgt, b.species});
  }
}
struct Layer {
  int dx, dmid, gin, dout, wn;
  char tag[13];
  float conv_scale;
  int mlp[4];
  snet_conv_plan *conv = nullptr;
  snet_mlp_plan *mlp_plan = nullptr;
  snet_fused_plan *fused = nullptr;  // radial-MLP last layer inside the tensor-product kernels (where the shape has them)
  int32_t *gxe_chunks = nullptr;     // device: chunk order of the fused reverse kernel's g_xe rows (dx / 16 entries)
  // scalar-output layer (t > 0): source-row gradient as a forward convolution of the transposed product (snet_hip.h)
  snet_conv_plan *tconv = nullptr;
  snet_mlp_plan *tmlp = nullptr;
  snet_fused_plan *tfused = nullptr;
  int tile_mode = 0;   // work list of the fused reverse kernel: 0 = per-row tiles, 1 = packed (snet_fused_plan_tile_mode)
  std::vector<int32_t> t_dead;       // (offset, length) column ranges of g_h it leaves unwritten
  Linear sc, si1, si2;
  std::vector<snet_gate_seg> segs;
};

bool dev_upload(const std::vector<float> &h, float **d) {
  if (h.empty()) {
    *d = nullptr;
    return true;
  }
  if (hipMalloc((void **)d, h.size() * 4) != hipSuccess) return false;
  return hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
}

bool upload_split(const std::vector<float> &w, int K, int N, void **d) {
  std::vector<unsigned char> packed((size_t)snet_gemm_split_size(K, N));
  if (snet_gemm_split_pack(w.data(), K, N, packed.data())) return false;
  if (hipMalloc(d, packed.size()) != hipSuccess) return false;
  return hipMemcpy(*d, packed.data(), packed.size(), hipMemcpyHostToDevice) == hipSuccess;
}

bool read_linear(Reader &r, Linear &L) {
  L.dim_in = r.i32();
  L.dim_out = r.i32();
  L.n_species = r.i32();
  const int nb = r.i32(), nz = r.i32(), nzi = r.i32();
  if (!r.ok || nb < 0 || nb > 4096 || nz < 0 || nz > 4096 || nzi < 0 || nzi > 4096) return false;
  L.blocks.resize(nb);
  for (auto &b : L.blocks) {
    b.l = r.i32(); b.in_off = r.i32(); b.mul_in = r.i32(); b.out_off = r.i32(); b.mul_out = r.i32();
    b.species = r.i32(); b.accumulate = r.i32();
    if (!r.ok || b.l < 0 || b.l > 8 || b.mul_in <= 0 || b.mul_out <= 0 || b.mul_in > (1 << 16) || b.mul_out > (1 << 16) ||
        b.in_off < 0 || b.out_off < 0 || b.species < -1)
      return false;
  }
  for (int i = 0; i < nz; ++i) {
    const int off = r.i32(), len = r.i32();
    L.zero.push_back({off, len});
  }
  for (int i = 0; i < nzi; ++i) {
    const int off = r.i32(), len = r.i32();
    L.zero_in.push_back({off, len});
  }
  std::map<std::pair<int, int>, std::vector<double>> row_sq;
  for (auto &b : L.blocks) {
    std::vector<float> w = r.farr((size_t)b.mul_in * b.mul_out);
    if (!r.ok) return false;
    std::vector<float> wt(w.size());
    for (int k = 0; k < b.mul_in; ++k)
      for (int n = 0; n < b.mul_out; ++n) wt[(size_t)n * b.mul_in + k] = w[(size_t)k * b.mul_out + n];
    if (!upload_split(w, b.mul_in, b.mul_out, &b.W) || !upload_split(wt, b.mul_out, b.mul_in, &b.WT)) return false;
    // squared norms of the rows of w (one per input channel), summed over the blocks that feed the same input block
    auto &sq = row_sq[{b.in_off, b.species}];
    sq.resize((size_t)b.mul_in, 0.0);
    for (int k = 0; k < b.mul_in; ++k)
      for (int n = 0; n < b.mul_out; ++n) sq[k] += (double)w[(size_t)k * b.mul_out + n] * w[(size_t)k * b.mul_out + n];
  }
  for (auto &kv : row_sq)
    for (double v : kv.second) L.t_norm = std::max(L.t_norm, (float)std::sqrt(v));
  const int has_bias = r.i32();
  if (!r.ok || (has_bias != 0 && has_bias != 1)) return false;
  if (has_bias) {
    std::vector<float> b = r.farr((size_t)L.dim_out);
    if (!r.ok || !dev_upload(b, &L.bias)) return false;
  }
  plan_groups(L, false);
  plan_groups(L, true);
  return r.ok;
}

// bump allocator over one device arena (grown on demand between evaluations)
struct Arena {
  char *base = nullptr;
  size_t cap = 0, off = 0;
  float *f(size_t n) {
    const size_t bytes = ((n * 4 + 255) / 256) * 256;
    float *p = reinterpret_cast<float *>(base + off);
    off += bytes;
    return p;
  }
};

}  // namespace

struct snet_model {
  int n_species, n_layers, lmax, normalize, n_basis, cutoff_kind, poly_p, act_radial, n_scale, d0;
  float cutoff, cutoff_on, act_cst;
  std::vector<float> coeffs;
  float *embed = nullptr, *scale = nullptr, *shift = nullptr;
  double *ro_v = nullptr, ro_c = 0.0;  // folded readout vector (device) and constant
  // `readout_as_fcn` (nn/linear.py:145-180): x -> act(x W0) cst -> ... -> e on exact fp32 GEMMs, like engine.py
  std::vector<int> fcn_dims;            // empty: folded two-linear readout
  std::vector<float *> fcn_w, fcn_wt;   // W_i [d_i, d_{i+1}] and its transpose (device)
  int fcn_act = 0;
  float fcn_cst = 1.f;
  double *one_d = nullptr;              // {1.0}: snet_readout_grad with it writes scale[type] (the seed of the reverse pass)
  float *h0_table = nullptr, *sc0_table = nullptr;  // [n_species, dx0], [n_species, gin0]: layer 0's SI1(x) / sc(x) per species
  float scale0 = 1.f;
  std::vector<Layer> layers;
  Linear ro1, ro2;
  std::string meta;  // key=value lines
  Arena arena;
  snet_halo_fn halo_fwd = nullptr, halo_rev = nullptr;
  void *halo_user = nullptr;
  bool fold_forces = true;
  std::vector<std::vector<int32_t>> species_rows_host;  // scratch
  int32_t *species_rows = nullptr;                       // device, concatenated
  size_t species_rows_cap = 0;
  std::vector<int64_t> species_off, species_cnt;
  // Topology cache (snet_model_set_topology_cache): what depends on the edge LIST only -- the per-species row lists, the
  // 16-edge tile list of the reverse kernels (whose construction reads a count back: a stream sync), the grouping of the
  // edges by source -- is kept across evaluations while the caller's index arrays are the same device buffers and
  // snet_model_topology_changed has not been called.  A host whose list is rebuilt every step (snet_md_compute filters the
  // LAMMPS list by the cutoff per step, like pair_e3gnn.cpp:150-200) leaves it off.
  bool topo_cache = false, topo_valid = false;
  struct TopoKey { int64_t NT, N, E; const void *row_ptr, *src, *eperm, *w_row; } topo_key{0, 0, 0, nullptr, nullptr, nullptr, nullptr};
  int32_t *c_tile_ptr = nullptr, *c_tile_node = nullptr, *c_center_t = nullptr, *c_w_row_t = nullptr;
  size_t c_tile_cap = 0, c_edge_cap = 0, c_node_cap = 0;
  int64_t c_n_tiles = 0;
  // interior / boundary split around the ghost exchange (snet_model_set_interior; library halo only)
  int64_t n_interior = 0;
  hipStream_t halo_stream = nullptr;
  hipEvent_t ev_h0 = nullptr, ev_h1 = nullptr;
  int32_t *c_tile_ptr_b = nullptr;   // tile pointers re-based on the boundary rows' sub-list
  size_t c_tile_ptr_b_cap = 0;
  int64_t c_tile_k = 0;              // first tile of the first boundary row
  bool c_split_valid = false;        // c_tile_ptr_b / c_tile_k hold the cached graph's boundary sub-list (built only by an
                                     // evaluation that ran split: a cache hit of an un-split evaluation must not trust them)
  // packed tiles (snet_edge_tiles_packed): one list, the interior rows' tiles [0, k) in front of the boundary rows'
  int32_t *c_ptile_e0 = nullptr, *c_ptile_nodes = nullptr;
  size_t c_ptile_cap = 0;
  int64_t c_pn_tiles = 0, c_ptile_k = 0;
  std::vector<int32_t> types_prev;  // species of the local atoms the cached row lists were built from
  int64_t eval_syncs = 0;           // stream synchronisations issued inside snet_model_eval so far (tests / tools)
  // second stream for the radial MLPs (graphs of at most OVERLAP_MAX_EDGES edges, see engine.py)
  hipStream_t side = nullptr;
  hipEvent_t ev_main = nullptr, ev_bwd[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> ev_w;
  bool overlap = getenv("SNET_NO_OVERLAP") == nullptr;
  // precision mode of the fused in-kernel products (snet_fused_plan_create): 4 = f16x3, the fp32-class default.  SNET_FUSED_TERMS=2
  // (bf16x3: BASELINE config 5's "bf16 compute", outside the 1e-4 eV/A bar at MD-scale forces) / 3 / 1 select the others for
  // measurements (tools/md_loop.py); anything else keeps the default
  int fused_terms = [] {
    const char *e = getenv("SNET_FUSED_TERMS");
    const int v = e ? atoi(e) : SNET_FUSED_TERMS_DEFAULT;
    return v >= 1 && v <= 4 ? v : SNET_FUSED_TERMS_DEFAULT;
  }();
};

constexpr int64_t OVERLAP_MAX_EDGES = 1000000;  // same policy as engine.py

namespace {

int run_linear(snet_model *m, const Linear &L, const float *x, float *y, int64_t n, bool transpose, bool accumulate_all,
               hipStream_t st) {
  if (n <= 0) return 0;
  const int64_t a_stride = transpose ? L.dim_out : L.dim_in, c_stride = transpose ? L.dim_in : L.dim_out;
  if (!accumulate_all)  // column ranges no launch writes
    for (auto &z : transpose ? L.zero_in : L.zero)
      if (hipMemset2DAsync(y + z.first, (size_t)c_stride * 4, 0, (size_t)z.second * 4, (size_t)n, st) != hipSuccess) {
        snet::set_error("snet_model_eval: memset failed");
        return 1;
      }
  for (const Linear::Group &g : transpose ? L.rev : L.fwd) {
    const int32_t *rows = nullptr;
    int64_t cnt = n;
    if (g.species >= 0) {
      cnt = m->species_cnt[g.species];
      rows = m->species_rows + m->species_off[g.species];
      if (cnt == 0) continue;
    }
    int rc;
    if (accumulate_all) {
      snet_gemm_desc tmp[SNET_MAX_GEMM_GROUP];
      for (size_t i = 0; i < g.descs.size(); ++i) {
        tmp[i] = g.descs[i];
        tmp[i].accumulate = 1;
      }
      rc = snet_gemm_grouped(tmp, (int)g.descs.size(), x, y, cnt, a_stride, c_stride, rows, st);
    } else {
      rc = snet_gemm_grouped(g.descs.data(), (int)g.descs.size(), x, y, cnt, a_stride, c_stride, rows, st);
    }
    if (rc) return rc;
  }
  if (!transpose && L.bias) return snet_add_row_bias(y, L.bias, n, L.dim_out, st);  // multi-modal linear
  return 0;
}

}  // namespace

extern "C" int snet_model_load_memory(const void *blob, int64_t n_bytes, snet_model **out) {
  SNET_REQUIRE(blob != nullptr && out != nullptr && n_bytes > 8, "snet_model_load: null / empty model");
  Reader r{static_cast<const unsigned char *>(blob), static_cast<const unsigned char *>(blob) + n_bytes};
  char magic[8];
  r.get(magic, 8);
  SNET_REQUIRE(r.ok && memcmp(magic, "SNETMDL4", 8) == 0, "snet_model_load: not a .snet model file of this version (SNETMDL4)");
  auto *m = new snet_model;
  bool good = false;
  try {
  m->n_species = r.i32(); m->n_layers = r.i32(); m->lmax = r.i32(); m->normalize = r.i32(); m->n_basis = r.i32();
  m->cutoff_kind = r.i32(); m->poly_p = r.i32(); m->act_radial = r.i32(); m->n_scale = r.i32(); m->d0 = r.i32();
  m->cutoff = r.f32(); m->cutoff_on = r.f32(); m->act_cst = r.f32();
  good = r.ok && m->n_layers > 0 && m->n_layers < 64 && m->n_basis > 0 && m->n_basis <= 16 && m->n_species > 0 &&
         m->n_species <= 4096 && m->lmax >= 0 && m->lmax <= 3 && (m->normalize == 0 || m->normalize == 1) &&
         (m->act_radial >= 0 && m->act_radial < 7) && (m->cutoff_kind == 0 || m->cutoff_kind == 1) &&
         (m->n_scale == 1 || m->n_scale == m->n_species) && m->d0 > 0 && m->d0 <= (1 << 16) && m->cutoff > 0.f;
  if (good) {
    m->coeffs = r.farr(m->n_basis);
    std::vector<float> emb = r.farr((size_t)m->n_species * m->d0), sc = r.farr(m->n_scale), sh = r.farr(m->n_scale);
    good = r.ok && dev_upload(emb, &m->embed) && dev_upload(sc, &m->scale) && dev_upload(sh, &m->shift);
    if (good) m->scale0 = sc[0];
  }
  for (int t = 0; good && t < m->n_layers; ++t) {
    Layer L;
    L.dx = r.i32(); L.dmid = r.i32(); L.gin = r.i32(); L.dout = r.i32(); L.wn = r.i32();
    r.get(L.tag, 12);
    L.tag[12] = 0;
    L.conv_scale = r.f32();
    for (int i = 0; i < 4; ++i) L.mlp[i] = r.i32();
    if (!r.ok) { good = false; break; }
    bool dims_ok = L.dx > 0 && L.dmid > 0 && L.gin > 0 && L.dout > 0 && L.wn > 0;
    for (int i = 0; i < 4; ++i) dims_ok = dims_ok && L.mlp[i] > 0 && L.mlp[i] <= (1 << 16);
    for (int v : {L.dx, L.dmid, L.gin, L.dout, L.wn}) dims_ok = dims_ok && v <= (1 << 20);
    if (!dims_ok) { good = false; break; }
    std::vector<float> w0 = r.farr((size_t)L.mlp[0] * L.mlp[1]), w1 = r.farr((size_t)L.mlp[1] * L.mlp[2]),
                       w2 = r.farr((size_t)L.mlp[2] * L.mlp[3]);
    good = r.ok && L.mlp[3] == L.wn;
    if (good && snet_radial_mlp_plan_create(L.mlp[0], L.mlp[1], L.mlp[2], L.mlp[3], w0.data(), w1.data(), w2.data(),
                                            m->act_radial, m->act_cst, 1, &L.mlp_plan)) good = false;
    if (good && snet_conv_plan_create(L.tag, &L.conv)) good = false;
    if (good && getenv("SNET_NO_FUSED") == nullptr && snet_conv_fused_available(L.conv) &&
        snet_fused_plan_create(L.conv, L.mlp_plan, m->fused_terms, &L.fused))
      good = false;
    if (good && L.fused) {
      L.tile_mode = snet_fused_plan_tile_mode(L.fused);
      std::vector<int32_t> cp((size_t)L.dx / 16);
      good = snet_fused_plan_gxe_chunks(L.fused, cp.data(), (int32_t)cp.size()) == 0 &&
             hipMalloc((void **)&L.gxe_chunks, cp.size() * 4) == hipSuccess &&
             hipMemcpy(L.gxe_chunks, cp.data(), cp.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (good && L.fused && t > 0 && L.dmid < L.dx && getenv("SNET_NO_TRANSPOSED") == nullptr) {
      char ttag[13];
      std::vector<float> cs((size_t)L.wn);
      int32_t dead[32], nd = 0;
      if (snet_conv_plan_transposed(L.conv, ttag, cs.data(), dead, 32, &nd) == 0 && ttag[0] != 0 &&
          snet_conv_plan_create(ttag, &L.tconv) == 0 && snet_conv_fused_available(L.tconv)) {
        std::vector<float> w2t(w2);
        for (int k = 0; k < L.mlp[2]; ++k)
          for (int c = 0; c < L.wn; ++c) w2t[(size_t)k * L.wn + c] *= cs[c];
        good = snet_radial_mlp_plan_create(L.mlp[0], L.mlp[1], L.mlp[2], L.mlp[3], w0.data(), w1.data(), w2t.data(),
                                           m->act_radial, m->act_cst, 1, &L.tmlp) == 0 &&
               snet_fused_plan_create(L.tconv, L.tmlp, m->fused_terms, &L.tfused) == 0;
        L.t_dead.assign(dead, dead + 2 * nd);
      } else if (L.tconv) {  // the transposed shape is not in this build: per-edge rows + segment sum
        snet_conv_plan_destroy(L.tconv);
        L.tconv = nullptr;
      }
    }
    good = good && read_linear(r, L.sc) && read_linear(r, L.si1) && read_linear(r, L.si2);
    if (good) {
      const int ns = r.i32();
      good = r.ok && ns >= 1 && ns <= SNET_MAX_GATE_SEGS;
      for (int i = 0; good && i < ns; ++i) {
        snet_gate_seg s;
        s.kind = r.i32(); s.in_off = r.i32(); s.out_off = r.i32(); s.mul = r.i32(); s.l = r.i32();
        s.gate_off = r.i32(); s.act = r.i32(); s.cst = r.f32();
        L.segs.push_back(s);
      }
      good = good && r.ok;
    }
    m->layers.push_back(L);
  }
  const int ro_kind = good ? r.i32() : 0;
  good = good && r.ok && (ro_kind == 0 || ro_kind == 1);
  if (good && ro_kind == 0) good = read_linear(r, m->ro1) && read_linear(r, m->ro2);
  if (good) {  // layer 0's species-only rows SI1(x), sc(x): fp64-evaluated tables (model_spec.species_only_tables)
    const int dx0 = r.i32(), gin0 = r.i32();
    good = r.ok && dx0 == m->layers[0].dx && (gin0 == 0 || gin0 == m->layers[0].gin) && (gin0 != 0) == m->layers[0].sc.present();
    if (good) {
      std::vector<float> h0 = r.farr((size_t)m->n_species * dx0), sc0 = r.farr((size_t)m->n_species * gin0);
      good = r.ok && dev_upload(h0, &m->h0_table) && dev_upload(sc0, &m->sc0_table);
    }
  }
  if (good && ro_kind == 1) {
    const int nl = r.i32();
    good = r.ok && nl >= 1 && nl <= 16;
    for (int i = 0; good && i <= nl; ++i) {
      m->fcn_dims.push_back(r.i32());
      good = r.ok && m->fcn_dims.back() >= 1 && m->fcn_dims.back() <= (1 << 16);
    }
    m->fcn_act = good ? r.i32() : 0;
    m->fcn_cst = good ? r.f32() : 1.f;
    good = good && r.ok && m->fcn_act >= 0 && m->fcn_act < snet::N_ACT && m->fcn_dims.front() == m->layers.back().dout && m->fcn_dims.back() == 1;
    for (int i = 0; good && i < nl; ++i) {
      const int di = m->fcn_dims[i], dn = m->fcn_dims[i + 1];
      std::vector<float> w = r.farr((size_t)di * dn), wt((size_t)di * dn);
      for (int a = 0; a < di && r.ok; ++a)
        for (int b = 0; b < dn; ++b) wt[(size_t)b * di + a] = w[(size_t)a * dn + b];
      float *dw = nullptr, *dwt = nullptr;
      good = r.ok && dev_upload(w, &dw) && dev_upload(wt, &dwt);
      m->fcn_w.push_back(dw);
      m->fcn_wt.push_back(dwt);
    }
    const double one = 1.0;
    good = good && hipMalloc((void **)&m->one_d, 8) == hipSuccess && hipMemcpy(m->one_d, &one, 8, hipMemcpyHostToDevice) == hipSuccess;
    if (good) m->ro1.dim_in = m->fcn_dims.front();
  }
  if (good && ro_kind == 0) {  // folded readout: e_i = x_i . v + c (model_spec.folded_readout)
    const int d_ro = r.i32();
    m->ro_c = r.f64();
    good = r.ok && d_ro == m->ro1.dim_in && d_ro == m->layers.back().dout;
    if (good) {
      std::vector<double> v = r.darr((size_t)d_ro);
      good = r.ok && hipMalloc((void **)&m->ro_v, v.size() * 8) == hipSuccess &&
             hipMemcpy(m->ro_v, v.data(), v.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
    }
  }
  if (good) {
    const int nm = r.i32();
    good = r.ok && nm >= 0 && nm <= (1 << 20) && (int64_t)(r.end - r.p) == nm;
    if (good) m->meta.assign(reinterpret_cast<const char *>(r.p), (size_t)nm);
  }
  } catch (const std::exception &e) {  // nothing may propagate through an extern "C" entry point
    snet::set_error(std::string("exception: ") + e.what());
    good = false;
  }
  if (!good) {
    const std::string prev = snet_last_error();
    snet::set_error("snet_model_load: malformed model file or device upload failed" +
                    (prev.empty() ? std::string() : " (" + prev + ")"));
    snet_model_destroy(m);  // releases the device allocations and plans created so far
    return 1;
  }
  *out = m;
  return 0;
}

extern "C" int snet_model_load(const char *path, snet_model **out) {
  SNET_REQUIRE(path != nullptr, "snet_model_load: null path");
  FILE *f = fopen(path, "rb");
  if (!f) {
    snet::set_error(std::string("snet_model_load: cannot open ") + path);
    return 1;
  }
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<unsigned char> buf(n > 0 ? n : 0);
  const size_t got = n > 0 ? fread(buf.data(), 1, n, f) : 0;
  fclose(f);
  SNET_REQUIRE((long)got == n && n > 0, "snet_model_load: short read");
  return snet_model_load_memory(buf.data(), n, out);
}

extern "C" void snet_model_destroy(snet_model *m) {
  if (!m) return;
  auto free_lin = [](Linear &L) {
    for (auto &b : L.blocks) {
      if (b.W) (void)hipFree(b.W);
      if (b.WT) (void)hipFree(b.WT);
    }
    if (L.bias) (void)hipFree(L.bias);
  };
  for (float *w : m->fcn_w) if (w) (void)hipFree(w);
  for (float *w : m->fcn_wt) if (w) (void)hipFree(w);
  if (m->one_d) (void)hipFree(m->one_d);
  for (auto &L : m->layers) {
    snet_fused_plan_destroy(L.fused);
    snet_fused_plan_destroy(L.tfused);
    snet_radial_mlp_plan_destroy(L.tmlp);
    snet_conv_plan_destroy(L.tconv);
    if (L.gxe_chunks) (void)hipFree(L.gxe_chunks);
    snet_conv_plan_destroy(L.conv);
    snet_radial_mlp_plan_destroy(L.mlp_plan);
    free_lin(L.sc); free_lin(L.si1); free_lin(L.si2);
  }
  free_lin(m->ro1); free_lin(m->ro2);
  for (hipEvent_t e : m->ev_w) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : {m->ev_main, m->ev_bwd[0], m->ev_bwd[1]}) if (e) (void)hipEventDestroy(e);
  if (m->side) (void)hipStreamDestroy(m->side);
  if (m->halo_stream) (void)hipStreamDestroy(m->halo_stream);
  for (hipEvent_t e : {m->ev_h0, m->ev_h1}) if (e) (void)hipEventDestroy(e);
  for (void *d : {(void *)m->embed, (void *)m->scale, (void *)m->shift, (void *)m->h0_table, (void *)m->sc0_table, (void *)m->ro_v, (void *)m->c_tile_ptr, (void *)m->c_tile_node,
                  (void *)m->c_center_t, (void *)m->c_w_row_t, (void *)m->c_tile_ptr_b, (void *)m->c_ptile_e0, (void *)m->c_ptile_nodes, (void *)m->arena.base,
                  (void *)m->species_rows})
    if (d) (void)hipFree(d);
  delete m;
}

extern "C" int snet_model_info(const snet_model *m, float *cutoff, int32_t *n_species, int32_t *n_layers,
                               int32_t *comm_dims, int32_t max_layers) {
  SNET_REQUIRE(m != nullptr, "snet_model_info: null model");
  if (cutoff) *cutoff = m->cutoff;
  if (n_species) *n_species = m->n_species;
  if (n_layers) *n_layers = m->n_layers;
  if (comm_dims)
    for (int t = 0; t < m->n_layers && t < max_layers; ++t) comm_dims[t] = m->layers[t].dx;
  return 0;
}

extern "C" int snet_model_meta(const snet_model *m, const char *key, char *value, int32_t capacity) {
  SNET_REQUIRE(m != nullptr && key != nullptr && value != nullptr && capacity > 0, "snet_model_meta: bad argument");
  const std::string k = std::string(key) + "=";
  size_t pos = 0;
  while (pos < m->meta.size()) {
    size_t eol = m->meta.find('\n', pos);
    if (eol == std::string::npos) eol = m->meta.size();
    if (m->meta.compare(pos, k.size(), k) == 0) {
      const std::string v = m->meta.substr(pos + k.size(), eol - pos - k.size());
      SNET_REQUIRE((int64_t)v.size() < capacity, "snet_model_meta: value does not fit the buffer");
      memcpy(value, v.c_str(), v.size() + 1);
      return 0;
    }
    pos = eol + 1;
  }
  snet::set_error(std::string("snet_model_meta: no such key: ") + key);
  return 3;
}

extern "C" int snet_model_set_halo(snet_model *m, snet_halo_fn forward, snet_halo_fn reverse, void *user,
                                   int32_t fold_forces) {
  SNET_REQUIRE(m != nullptr, "snet_model_set_halo: null model");
  SNET_REQUIRE((forward == nullptr) == (reverse == nullptr), "snet_model_set_halo: set both hooks or neither");
  m->halo_fwd = forward;
  m->halo_rev = reverse;
  m->halo_user = user;
  m->fold_forces = fold_forces != 0;
  m->topo_valid = false;   // the split lists of a cached graph depend on whether (and which) halo is installed
  m->c_split_valid = false;
  return 0;
}

extern "C" int snet_model_set_interior(snet_model *m, int64_t n_interior) {
  SNET_REQUIRE(m != nullptr && n_interior >= 0, "snet_model_set_interior: bad argument");
  if (m->n_interior != n_interior) m->topo_valid = false;   // the tile sub-lists depend on it
  m->n_interior = n_interior;
  return 0;
}

extern "C" int snet_model_set_topology_cache(snet_model *m, int32_t enable) {
  SNET_REQUIRE(m != nullptr, "snet_model_set_topology_cache: null model");
  m->topo_cache = enable != 0;
  m->topo_valid = false;
  return 0;
}

extern "C" int snet_model_topology_changed(snet_model *m) {
  SNET_REQUIRE(m != nullptr, "snet_model_topology_changed: null model");
  m->topo_valid = false;
  m->types_prev.clear();
  return 0;
}

extern "C" int64_t snet_model_eval_syncs(const snet_model *m) { return m ? m->eval_syncs : -1; }

extern "C" int snet_model_eval(snet_model *m, int64_t NT, int64_t N, int64_t E, const int32_t *types,
                               const int32_t *types_host, const int32_t *row_ptr, const int32_t *src,
                               const int32_t *col_ptr, const int32_t *eperm, const float *edge_vec,
                               const int32_t *w_row, const int32_t *pair_edge, int64_t n_pairs, double *energy,
                               float *e_atom, float *dE_dr, float *forces, double *virial, float *virial_atom,
                               void *stream) {
  SNET_REQUIRE(m != nullptr, "snet_model_eval: null model");
  SNET_REQUIRE(NT >= N && N > 0 && E >= 0, "snet_model_eval: need n_total >= n_local > 0 and n_edges >= 0");
  SNET_REQUIRE(NT == N || (m->halo_fwd && m->halo_rev), "snet_model_eval: ghost atoms need halo callbacks");
  // The exchange is a collective of send/recv pairs: a rank without ghost rows of its own may still own rows its
  // peers wait for (slab / cluster systems whose only ghosts are periodic self-images), so an installed halo is
  // called on every layer whatever NT - N is (snet_halo_* handles n_ghost == 0)
  const bool has_halo = m->halo_fwd != nullptr && m->halo_rev != nullptr;
  const bool pairs = w_row != nullptr && E > 0;
  SNET_REQUIRE(!pairs || (pair_edge != nullptr && n_pairs > 0 && n_pairs <= E),
               "snet_model_eval: w_row needs pair_edge and 0 < n_pairs <= n_edges");
  const int64_t WR = pairs ? n_pairs : E;  // rows of each layer's radial-weight matrix
  bool any_fused = false, any_transposed = false;
  bool need_tiles[2] = {false, false};   // per work-list format of the fused reverse kernels
  for (auto &L : m->layers) {
    any_fused |= L.fused != nullptr;
    any_transposed |= L.tfused != nullptr;
    if (L.fused) need_tiles[L.tile_mode != 0] = true;
  }
  // the second stream only carries the separate radial-MLP kernels (same policy as engine.py)
  bool ov = m->overlap && E > 0 && E <= OVERLAP_MAX_EDGES && !any_fused;
  if (ov && m->side == nullptr) {  // created on first use; any failure just keeps everything on one stream
    bool ok = hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&m->ev_main, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&m->ev_bwd[0], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&m->ev_bwd[1], hipEventDisableTiming) == hipSuccess;
    m->ev_w.resize(m->n_layers, nullptr);
    for (int t = 0; ok && t < m->n_layers; ++t) ok = hipEventCreateWithFlags(&m->ev_w[t], hipEventDisableTiming) == hipSuccess;
    if (!ok) m->overlap = ov = false;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nb = m->n_basis, nsh = (m->lmax + 1) * (m->lmax + 1);
  const int Lc = m->n_layers;

  // ---- per-species row lists (FCTP self-connection only)
  bool need_rows = false;
  for (auto &L : m->layers) need_rows |= L.sc.n_species > 0;
  if (need_rows) {
    SNET_REQUIRE(types_host != nullptr, "snet_model_eval: this model needs types_host (per-species self-connection)");
    const bool same_types = (int64_t)m->types_prev.size() == N && m->species_rows != nullptr &&
                            memcmp(m->types_prev.data(), types_host, (size_t)N * 4) == 0;
    if (!same_types) {
    m->species_rows_host.assign(m->n_species, {});
    for (int64_t i = 0; i < N; ++i) {
      SNET_REQUIRE(types_host[i] >= 0 && types_host[i] < m->n_species, "snet_model_eval: species index out of range");
      m->species_rows_host[types_host[i]].push_back((int32_t)i);
    }
    if (m->species_rows_cap < (size_t)N) {
      if (m->species_rows) (void)hipFree(m->species_rows);
      SNET_REQUIRE(hipMalloc((void **)&m->species_rows, (size_t)(N + 1) * 4) == hipSuccess, "snet_model_eval: alloc");
      m->species_rows_cap = N;
    }
    m->species_off.assign(m->n_species, 0);
    m->species_cnt.assign(m->n_species, 0);
    int64_t o = 0;
    for (int s = 0; s < m->n_species; ++s) {
      auto &v = m->species_rows_host[s];
      m->species_off[s] = o;
      m->species_cnt[s] = (int64_t)v.size();
      if (!v.empty())
        SNET_REQUIRE(hipMemcpyAsync(m->species_rows + o, v.data(), v.size() * 4, hipMemcpyHostToDevice, st) == hipSuccess,
                     "snet_model_eval: upload of species rows failed");
      o += (int64_t)v.size();
    }
    SNET_REQUIRE(hipStreamSynchronize(st) == hipSuccess, "snet_model_eval: sync");  // host vectors are reused
    ++m->eval_syncs;
    m->types_prev.assign(types_host, types_host + N);
    }
  }

  // ---- arena sizing
  size_t need = 0;
  auto add = [&](size_t n) { need += ((n * 4 + 255) / 256) * 256; };
  size_t dmax = (size_t)m->d0, trans = 0;
  add((size_t)E * nb); add((size_t)E * nsh); add((size_t)E * nsh * 3); add((size_t)E * 3); add((size_t)E * nb);
  add((size_t)WR * nb);
  for (auto &L : m->layers) {
    dmax = dmax > (size_t)L.dout ? dmax : (size_t)L.dout;
    add((size_t)NT * L.dx); add(L.fused ? (size_t)WR * 64 : (size_t)WR * L.wn); add((size_t)N * L.gin);  // saved h, w | h2, y
    add((size_t)NT + 64);  // x_max (kept through the reverse pass)
    const size_t t = ((size_t)N * L.gin + 64) * 2 + (size_t)N * L.dmid * 2 + (size_t)E * (L.fused ? 64 : L.wn) + (size_t)E * L.dx +
                     (size_t)NT * L.dx * 2 + (size_t)N * L.dout + (size_t)NT + (size_t)N + 4096;  // + x_max, g_max
    trans = trans > t ? trans : t;
  }
  size_t wn_max = 0;
  for (auto &L : m->layers) wn_max = wn_max > (size_t)L.wn ? wn_max : (size_t)L.wn;
  if (ov) { add((size_t)E * wn_max); add((size_t)E * wn_max); }  // g_w double buffer (its reader runs on the side stream)
  if (need_tiles[0]) { add((size_t)N + 64); add((size_t)N + (size_t)E / 16 + 64); }  // tile_ptr, tile_node
  if (need_tiles[1]) { add((size_t)N + (size_t)E / 16 + 64); add(2 * ((size_t)N + (size_t)E / 16 + 64)); }  // packed: tile_e0, tile_nodes
  if (any_transposed) { add((size_t)E + 64); add((size_t)E + 64); add((size_t)E * nsh + 64); }  // center_t, w_row_t, sh_t
  add((size_t)NT * dmax * 2 + 256); add((size_t)N * (m->ro1.dim_out + 8) * 2); add(trans + 64 * 1024);
  for (size_t i = 0; i + 1 < m->fcn_dims.size(); ++i) {   // readout_as_fcn: z_i, a_i forward, the gradient rows in reverse
    add((size_t)N * m->fcn_dims[i + 1] + 64); add((size_t)N * m->fcn_dims[i + 1] + 64); add((size_t)N * m->fcn_dims[i] + 64);
  }
  need += 1 << 20;
  if (m->arena.cap < need) {
    if (m->arena.base) (void)hipFree(m->arena.base);
    m->arena.base = nullptr;
    m->arena.cap = 0;
    SNET_REQUIRE(hipMalloc((void **)&m->arena.base, need) == hipSuccess, "snet_model_eval: arena allocation failed");
    m->arena.cap = need;
  }
  Arena &A = m->arena;
  A.off = 0;

  snet_edge_params ep{m->cutoff, m->n_basis, m->cutoff_kind, m->poly_p, m->cutoff_on, m->lmax, m->normalize};
  float *emb = A.f((size_t)E * nb), *sh = A.f((size_t)E * nsh), *dsh = A.f((size_t)E * nsh * 3);
  float *g_vec = dE_dr ? dE_dr : A.f((size_t)E * 3), *g_emb = A.f((size_t)E * nb);
  int rc;
  if ((rc = snet_edge_embed_fwd(&ep, m->coeffs.data(), edge_vec, E, emb, sh, dsh, st))) return rc;
  const float *emb_w = emb;  // rows the radial MLP runs on: one per undirected pair when a pair map is given
  if (pairs) {
    float *ep = A.f((size_t)WR * nb);
    if ((rc = snet_gather_rows(emb, pair_edge, ep, WR, nb, st))) return rc;
    emb_w = ep;
  }
  float *x = A.f((size_t)NT * dmax), *x2 = A.f((size_t)NT * dmax);  // layer 0 reads the species tables, not x

  struct Saved { float *h, *w, *y; };  // w: radial weights [WR, wn], or (fused layers) hidden activations h2 [WR, 64]
  std::vector<Saved> saved(Lc);
  for (int t = 0; t < Lc; ++t) {
    saved[t].h = A.f((size_t)NT * m->layers[t].dx);
    saved[t].w = A.f(m->layers[t].fused ? (size_t)WR * 64 : (size_t)WR * m->layers[t].wn);
    saved[t].y = A.f((size_t)N * m->layers[t].gin);
  }
  // topology-only work: rebuilt unless the cache holds it for exactly these index arrays
  const snet_model::TopoKey key{NT, N, E, row_ptr, src, eperm, pairs ? w_row : nullptr};
  const bool topo_hit = m->topo_cache && m->topo_valid && memcmp(&key, &m->topo_key, sizeof key) == 0;
  // interior / boundary split: the library's own exchange (stream-safe) on a second stream, rows [0, n_int) have no ghost source
  const bool lib_halo = has_halo && m->halo_fwd == &snet_halo_forward && m->halo_rev == &snet_halo_reverse;
  const int64_t n_int = m->n_interior;
  bool hsplit = lib_halo && n_int > 0 && n_int < N && E > 0 && any_fused && getenv("SNET_NO_HALO_SPLIT") == nullptr;
  if (hsplit && m->halo_stream == nullptr) {
    hsplit = hipStreamCreateWithFlags(&m->halo_stream, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&m->ev_h0, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&m->ev_h1, hipEventDisableTiming) == hipSuccess;
  }
  int32_t *tile_ptr = nullptr, *tile_node = nullptr;
  int64_t n_tiles = 0;
  if (need_tiles[0] && E > 0) {  // 16-edge tiles of the CSR segments: work list of the fused reverse kernels
    if (m->topo_cache) {
      if (m->c_node_cap < (size_t)N + 64 || m->c_tile_cap < (size_t)N + (size_t)E / 16 + 64) {
        if (m->c_tile_ptr) (void)hipFree(m->c_tile_ptr);
        if (m->c_tile_node) (void)hipFree(m->c_tile_node);
        m->c_tile_ptr = m->c_tile_node = nullptr;
        m->c_node_cap = (size_t)N + 64;
        m->c_tile_cap = (size_t)N + (size_t)E / 16 + 64;
        SNET_REQUIRE(hipMalloc((void **)&m->c_tile_ptr, m->c_node_cap * 4) == hipSuccess &&
                         hipMalloc((void **)&m->c_tile_node, m->c_tile_cap * 4) == hipSuccess, "snet_model_eval: alloc");
      }
      tile_ptr = m->c_tile_ptr;
      tile_node = m->c_tile_node;
    } else {
      tile_ptr = reinterpret_cast<int32_t *>(A.f((size_t)N + 64));
      tile_node = reinterpret_cast<int32_t *>(A.f((size_t)N + (size_t)E / 16 + 64));
    }
    if (topo_hit) {
      n_tiles = m->c_n_tiles;
    } else {
      if ((rc = snet_edge_tiles(row_ptr, N, tile_ptr, tile_node, N + E / 16 + 1, &n_tiles, st))) return rc;
      ++m->eval_syncs;   // (snet_edge_tiles reads the tile count back)
      m->c_n_tiles = n_tiles;
    }
  }
  // packed tiles: built per row range when the step is split (no tile straddles the cut; the interior list's end sentinel is the
  // first boundary tile's first edge, so the two lists are one)
  int32_t *ptile_e0 = nullptr, *ptile_nodes = nullptr;
  int64_t pn_tiles = 0, ptile_k = 0;
  if (need_tiles[1] && E > 0) {
    const size_t cap = (size_t)N + (size_t)E / 16 + 2;
    if (m->topo_cache) {
      if (m->c_ptile_cap < cap) {
        if (m->c_ptile_e0) (void)hipFree(m->c_ptile_e0);
        if (m->c_ptile_nodes) (void)hipFree(m->c_ptile_nodes);
        m->c_ptile_e0 = m->c_ptile_nodes = nullptr;
        m->c_ptile_cap = cap + cap / 8 + 64;
        SNET_REQUIRE(hipMalloc((void **)&m->c_ptile_e0, (m->c_ptile_cap + 1) * 4) == hipSuccess &&
                         hipMalloc((void **)&m->c_ptile_nodes, m->c_ptile_cap * 8) == hipSuccess, "snet_model_eval: alloc");
      }
      ptile_e0 = m->c_ptile_e0;
      ptile_nodes = m->c_ptile_nodes;
    } else {
      ptile_e0 = reinterpret_cast<int32_t *>(A.f((size_t)N + (size_t)E / 16 + 64));
      ptile_nodes = reinterpret_cast<int32_t *>(A.f(2 * ((size_t)N + (size_t)E / 16 + 64)));
    }
    if (topo_hit) {
      pn_tiles = m->c_pn_tiles;
      ptile_k = m->c_ptile_k;
    } else {
      const int64_t cut = n_int > 0 && n_int < N ? n_int : 0;   // (whether or not this evaluation is split: one tiling per graph, like engine.py)
      int64_t nb = 0;
      if (cut > 0) {
        if ((rc = snet_edge_tiles_packed(row_ptr, 0, cut, ptile_e0, ptile_nodes, (int64_t)cap, &ptile_k, st))) return rc;
        ++m->eval_syncs;   // (the builder reads the tile count back)
      }
      if ((rc = snet_edge_tiles_packed(row_ptr, cut, N, ptile_e0 + ptile_k, ptile_nodes + 2 * ptile_k, (int64_t)cap - ptile_k, &nb, st)))
        return rc;
      ++m->eval_syncs;
      pn_tiles = ptile_k + nb;
      m->c_pn_tiles = pn_tiles;
      m->c_ptile_k = ptile_k;
    }
  }
  int32_t *center_t = nullptr, *w_row_t = nullptr;  // edges grouped by source (transposed scalar convolution)
  float *sh_t = nullptr;
  if (any_transposed && E > 0) {
    if (m->topo_cache) {
      if (m->c_edge_cap < (size_t)E + 64) {
        if (m->c_center_t) (void)hipFree(m->c_center_t);
        if (m->c_w_row_t) (void)hipFree(m->c_w_row_t);
        m->c_center_t = m->c_w_row_t = nullptr;
        m->c_edge_cap = (size_t)E + 64;
        SNET_REQUIRE(hipMalloc((void **)&m->c_center_t, m->c_edge_cap * 4) == hipSuccess &&
                         hipMalloc((void **)&m->c_w_row_t, m->c_edge_cap * 4) == hipSuccess, "snet_model_eval: alloc");
      }
      center_t = m->c_center_t;
      w_row_t = m->c_w_row_t;
    } else {
      center_t = reinterpret_cast<int32_t *>(A.f((size_t)E + 64));
      w_row_t = reinterpret_cast<int32_t *>(A.f((size_t)E + 64));
    }
    sh_t = A.f((size_t)E * nsh + 64);
    if (!topo_hit && (rc = snet_edges_by_source(row_ptr, N, eperm, pairs ? w_row : nullptr, E, center_t, w_row_t, st))) return rc;
    if ((rc = snet_gather_rows(sh, eperm, sh_t, E, nsh, st))) return rc;   // (the harmonics change with the positions: every step)
  }
  int32_t *tile_ptr_b = nullptr;   // boundary rows' tiles: tile_node + k, tile pointers re-based by -k
  int64_t tile_k = 0;
  if (hsplit && need_tiles[0]) {
    if (m->c_tile_ptr_b_cap < (size_t)N + 64) {
      if (m->c_tile_ptr_b) (void)hipFree(m->c_tile_ptr_b);
      m->c_tile_ptr_b = nullptr;
      m->c_tile_ptr_b_cap = (size_t)N + 64;
      m->c_split_valid = false;
      SNET_REQUIRE(hipMalloc((void **)&m->c_tile_ptr_b, m->c_tile_ptr_b_cap * 4) == hipSuccess, "snet_model_eval: alloc");
    }
    tile_ptr_b = m->c_tile_ptr_b;
    if (topo_hit && m->c_split_valid) {
      tile_k = m->c_tile_k;
    } else {
      int32_t k32 = 0;
      SNET_REQUIRE(hipMemcpyAsync(&k32, tile_ptr + n_int, 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
                       hipStreamSynchronize(st) == hipSuccess, "snet_model_eval: tile readback failed");
      ++m->eval_syncs;
      tile_k = m->c_tile_k = k32;
      if ((rc = snet_i32_shift(tile_ptr, -k32, tile_ptr_b, N + 1, st))) return rc;
      m->c_split_valid = true;
    }
  }
  if (!topo_hit && !(hsplit && need_tiles[0])) m->c_split_valid = false;   // a new graph was tiled without its boundary sub-list
  if (m->topo_cache) {
    m->topo_key = key;
    m->topo_valid = true;
  }
  float *gw_buf[2] = {nullptr, nullptr};
  bool gw_busy[2] = {false, false};
  if (ov) {
    gw_buf[0] = A.f((size_t)E * wn_max);
    gw_buf[1] = A.f((size_t)E * wn_max);
    // every layer's radial weights depend on the edge embedding only: enqueue them all on the side stream now
    SNET_REQUIRE(hipEventRecord(m->ev_main, st) == hipSuccess && hipStreamWaitEvent(m->side, m->ev_main, 0) == hipSuccess,
                 "snet_model_eval: stream ordering failed");
    for (int t = 0; t < Lc; ++t) {
      if ((rc = snet_radial_mlp_fwd(m->layers[t].mlp_plan, emb_w, WR, saved[t].w, m->side))) return rc;
      SNET_REQUIRE(hipEventRecord(m->ev_w[t], m->side) == hipSuccess, "snet_model_eval: stream ordering failed");
    }
  }
  const size_t mark = A.off;  // transient region starts here

  {  // hidden activations of every fused layer's radial MLP in one launch (the layers share the edge embedding; up to 8 per call)
    const snet_mlp_plan *hp[8];
    float *ho[8];
    int nh = 0;
    for (int t = 0; t <= Lc; ++t) {
      if (t < Lc && m->layers[t].fused) { hp[nh] = m->layers[t].mlp_plan; ho[nh] = saved[t].w; ++nh; }
      if (nh == 8 || (t == Lc && nh > 0)) {
        if ((rc = snet_radial_mlp_hidden_fwd_layers(hp, nh, emb_w, WR, ho, st))) return rc;
        nh = 0;
      }
    }
  }

  // ---------------- forward
  for (int t = 0; t < Lc; ++t) {
    Layer &L = m->layers[t];
    A.off = mark;
    float *sc = nullptr;
    float *h = saved[t].h;
    if (t == 0) {  // species-only inputs: table lookups (ghost rows included)
      if (L.sc.present()) {
        sc = saved[t].y;
        if ((rc = snet_embed_rows(m->sc0_table, types, sc, N, L.gin, st))) return rc;
      }
      if ((rc = snet_embed_rows(m->h0_table, types, h, NT, L.dx, st))) return rc;
    } else {
      if (L.sc.present()) {
        sc = saved[t].y;
        if ((rc = run_linear(m, L.sc, x, sc, N, false, false, st))) return rc;
      }
      if ((rc = run_linear(m, L.si1, x, h, N, false, false, st))) return rc;
    }
    const bool fsplit = hsplit && t > 0 && L.fused != nullptr;
    if (t > 0 && has_halo) {
      if (fsplit)  // the ghost rows travel on the halo stream while the hidden radial layers and the interior rows run
        SNET_REQUIRE(hipEventRecord(m->ev_h0, st) == hipSuccess && hipStreamWaitEvent(m->halo_stream, m->ev_h0, 0) == hipSuccess,
                     "snet_model_eval: stream ordering failed");
      if ((rc = m->halo_fwd(m->halo_user, h, NT, N, L.dx, fsplit ? (void *)m->halo_stream : stream))) {
        snet::set_error("snet_model_eval: forward halo callback failed");
        return rc;
      }
      if (fsplit) SNET_REQUIRE(hipEventRecord(m->ev_h1, m->halo_stream) == hipSuccess, "snet_model_eval: stream ordering failed");
    }
    float *mid = A.f((size_t)N * L.dmid);
    if (E == 0) SNET_REQUIRE(hipMemsetAsync(mid, 0, (size_t)N * L.dmid * 4, st) == hipSuccess, "snet_model_eval: memset");
    else  // columns of pruned (unread) paths are never written by the tensor-product kernel: defined zeros
      for (auto &z : L.si2.zero_in)
        SNET_REQUIRE(hipMemset2DAsync(mid + z.first, (size_t)L.dmid * 4, 0, (size_t)z.second * 4, (size_t)N, st) == hipSuccess,
                     "snet_model_eval: memset");
    if (L.fused) {  // w = h2 @ W2 is formed inside the tensor-product kernel
      const int64_t na = fsplit ? n_int : 0;   // rows [0, na) before the exchange has landed, [na, N) after
      if (na > 0 && (rc = snet_conv_fwd_fused(L.fused, h, sh, saved[t].w, pairs ? w_row : nullptr, row_ptr, src, na, L.conv_scale, mid, st)))
        return rc;
      if (fsplit) SNET_REQUIRE(hipStreamWaitEvent(st, m->ev_h1, 0) == hipSuccess, "snet_model_eval: stream ordering failed");
      if ((rc = snet_conv_fwd_fused(L.fused, h, sh, saved[t].w, pairs ? w_row : nullptr, row_ptr + na, src, N - na, L.conv_scale,
                                    mid + (size_t)na * L.dmid, st)))
        return rc;
    } else {
      if (ov)
        SNET_REQUIRE(hipStreamWaitEvent(st, m->ev_w[t], 0) == hipSuccess, "snet_model_eval: stream ordering failed");
      else if ((rc = snet_radial_mlp_fwd(L.mlp_plan, emb_w, WR, saved[t].w, st)))
        return rc;
      if ((rc = snet_conv_fwd(L.conv, h, sh, saved[t].w, pairs ? w_row : nullptr, row_ptr, src, N, L.conv_scale, mid, st)))
        return rc;
    }
    float *y = saved[t].y;   // (sc == y when the layer has a self-connection: SI2 accumulates into its rows, as engine.py does)
    if ((rc = run_linear(m, L.si2, mid, y, N, false, sc != nullptr, st))) return rc;
    if ((rc = snet_gate_fwd(y, nullptr, x2, N, L.gin, L.dout, L.segs.data(), (int)L.segs.size(), st))) return rc;
    std::swap(x, x2);
  }
  A.off = mark;
  float *ea = e_atom ? e_atom : A.f((size_t)N + 64);
  float *g_x = x2, *gx_next = x;  // the forward features are dead after the readout: x / x2 ping-pong as gradient rows
  if (!m->fcn_dims.empty()) {
    // `readout_as_fcn`: the same launches, in the same order, as engine.py (exact fp32 GEMMs, activation kernels): bit-identical hosts
    const std::vector<int> &d = m->fcn_dims;
    const int nl = (int)d.size() - 1;
    std::vector<float *> zs;
    const float *a = x;
    float *z = nullptr;
    for (int i = 0; i < nl; ++i) {
      z = A.f((size_t)N * d[i + 1] + 64);
      if ((rc = snet_gemm(a, m->fcn_w[i], z, N, 1, d[i], d[i + 1], d[i], 0, d[i + 1], 0, nullptr, 0, st))) return rc;
      if (i + 1 < nl) {
        float *an = A.f((size_t)N * d[i + 1] + 64);
        if ((rc = snet_act_fwd(z, an, N * d[i + 1], m->fcn_act, m->fcn_cst, st))) return rc;
        zs.push_back(z);
        a = an;
      }
    }
    if ((rc = snet_rescale_reduce(z, types, m->scale, m->shift, m->n_scale, N, ea, energy, st))) return rc;
    // reverse: dE/dz_last = scale[type]; then W^T and the activation's derivative per layer; the last product lands in g_x
    float *gz = A.f((size_t)N + 64);
    if ((rc = snet_readout_grad(m->one_d, 1, types, m->scale, m->n_scale, N, gz, st))) return rc;
    for (int i = nl - 1; i >= 0; --i) {
      float *ga = i == 0 ? g_x : A.f((size_t)N * d[i] + 64);
      if ((rc = snet_gemm(gz, m->fcn_wt[i], ga, N, 1, d[i + 1], d[i], d[i + 1], 0, d[i], 0, nullptr, 0, st))) return rc;
      if (i > 0 && (rc = snet_act_bwd(zs[i - 1], ga, ga, N * d[i], m->fcn_act, m->fcn_cst, st))) return rc;
      gz = ga;
    }
  } else {
    if ((rc = snet_readout_energy(x, N, m->ro1.dim_in, m->ro_v, m->ro_c, types, m->scale, m->shift, m->n_scale, ea, energy, st)))
      return rc;
    // ---------------- reverse: dE/dx of the folded readout = scale[type] * v
    if ((rc = snet_readout_grad(m->ro_v, m->ro1.dim_in, types, m->scale, m->n_scale, N, g_x, st))) return rc;
  }
  SNET_REQUIRE(hipMemsetAsync(g_vec, 0, (size_t)E * 3 * 4, st) == hipSuccess &&
                   hipMemsetAsync(g_emb, 0, (size_t)E * nb * 4, st) == hipSuccess,
               "snet_model_eval: memset failed");
  // fp16 operands of the fused reverse kernels: the source-row bounds of all layers from one launch (up to 8 matrices per call)
  std::vector<float *> x_max_of((size_t)Lc, nullptr);
  if (E > 0 && m->fused_terms == 4) {
    const float *xs[8];
    float *outs[8];
    int64_t rows[8];
    int32_t dims[8];
    int nj = 0;
    for (int t = 0; t <= Lc; ++t) {
      if (t < Lc && m->layers[t].fused) {
        x_max_of[t] = A.f((size_t)NT + 64);
        xs[nj] = saved[t].h; outs[nj] = x_max_of[t]; rows[nj] = NT; dims[nj] = m->layers[t].dx; ++nj;
      }
      if (nj == 8 || (t == Lc && nj > 0)) {
        if ((rc = snet_row_absmax_multi(xs, rows, dims, outs, nj, st))) return rc;
        nj = 0;
      }
    }
  }
  const size_t mark2 = A.off;
  for (int t = Lc - 1; t >= 0; --t) {
    Layer &L = m->layers[t];
    A.off = mark2;
    float *g_y = A.f((size_t)N * L.gin);
    // fp16 operands of the fused reverse kernel: bound of every row of g_m = SI2^T g_y (Cauchy-Schwarz), taken in the same pass
    float *g_max = (L.fused && E > 0 && m->fused_terms == 4) ? A.f((size_t)N) : nullptr;
    if ((rc = snet_gate_bwd_norm(saved[t].y, g_x, g_y, N, L.gin, L.dout, L.segs.data(), (int)L.segs.size(),
                                 g_max ? L.si2.t_norm : 0.f, g_max, st)))
      return rc;
    float *g_m = A.f((size_t)N * L.dmid);
    if ((rc = run_linear(m, L.si2, g_y, g_m, N, true, false, st))) return rc;
    const bool transposed = t > 0 && L.tfused != nullptr && E > 0;
    float *g_xe = t > 0 && !transposed ? A.f((size_t)E * L.dx) : nullptr;
    if (L.fused) {  // g_w is contracted with W2^T inside the kernel; with the hidden-layer tail not even g_h2 leaves it
      const bool tail = snet_fused_plan_has_mlp_tail(L.fused) != 0;
      float *g_h2 = tail ? nullptr : A.f((size_t)E * 64);
      float *x_max = x_max_of[t];   // (bounds of every edge's g_w, see snet_row_absmax; computed before the layer loop)
      auto bwd_tiles = [&](const int32_t *tp, const int32_t *tn, int64_t nt) -> int {
        if (E <= 0 || nt <= 0) return 0;
        return snet_conv_bwd_fused(L.fused, saved[t].h, sh, dsh, saved[t].w, pairs ? w_row : nullptr, row_ptr, src, tp, tn, nt,
                                   L.conv_scale, g_m, g_xe, g_h2, tail ? emb : nullptr, tail ? g_emb : nullptr, g_vec, x_max, g_max, st);
      };
      const bool rsplit = hsplit && t > 0;
      const bool packed = L.tile_mode != 0;
      auto bwd_all = [&]() -> int { return packed ? bwd_tiles(ptile_e0, ptile_nodes, pn_tiles) : bwd_tiles(tile_ptr, tile_node, n_tiles); };
      if (!rsplit && (rc = bwd_all())) return rc;
      if (t > 0) {
        float *g_h = A.f((size_t)NT * L.dx);
        // source rows [a, b) of g_h: the transposed scalar convolution over the edges grouped by source, or the segment sum
        auto gh_rows = [&](int64_t a, int64_t b) -> int {
          if (b <= a) return 0;
          if (transposed)
            return snet_conv_fwd_fused(L.tfused, g_m, sh_t, saved[t].w, w_row_t, col_ptr + a, center_t, b - a, L.conv_scale,
                                       g_h + (size_t)a * L.dx, st);
          return snet_segment_sum_rows_chunked(g_xe, col_ptr + a, eperm, b - a, L.dx, L.gxe_chunks, g_h + (size_t)a * L.dx, st);
        };
        if (transposed)
          for (size_t i = 0; i + 1 < L.t_dead.size(); i += 2)
            SNET_REQUIRE(hipMemset2DAsync(g_h + L.t_dead[i], (size_t)L.dx * 4, 0, (size_t)L.t_dead[i + 1] * 4, (size_t)NT, st) ==
                             hipSuccess, "snet_model_eval: memset failed");
        if (rsplit) {
          // boundary tiles hold every edge with a ghost source: they go first, the ghost rows of g_h start travelling at once and
          // the interior tiles / the local rows run under the exchange (the transposed convolution needs no tiles at all)
          if (!transposed && (rc = packed ? bwd_tiles(ptile_e0 + ptile_k, ptile_nodes + 2 * ptile_k, pn_tiles - ptile_k)
                                          : bwd_tiles(tile_ptr_b, tile_node + tile_k, n_tiles - tile_k))) return rc;
          if ((rc = gh_rows(N, NT))) return rc;
          SNET_REQUIRE(hipEventRecord(m->ev_h0, st) == hipSuccess && hipStreamWaitEvent(m->halo_stream, m->ev_h0, 0) == hipSuccess,
                       "snet_model_eval: stream ordering failed");
          if ((rc = snet_halo_reverse_exchange(m->halo_user, g_h, NT, N, L.dx, m->halo_stream))) return rc;
          SNET_REQUIRE(hipEventRecord(m->ev_h1, m->halo_stream) == hipSuccess, "snet_model_eval: stream ordering failed");
          if ((rc = transposed ? bwd_all() : packed ? bwd_tiles(ptile_e0, ptile_nodes, ptile_k) : bwd_tiles(tile_ptr, tile_node, tile_k))) return rc;
          if ((rc = gh_rows(0, N))) return rc;
          if (!tail && (rc = snet_radial_mlp_hidden_bwd(L.mlp_plan, emb, g_h2, E, g_emb, st))) return rc;
          if (L.sc.present())   // sc^T g_y does not read the ghost gradients: it runs under the exchange too
            if ((rc = run_linear(m, L.sc, g_y, gx_next, N, true, false, st))) return rc;
          SNET_REQUIRE(hipStreamWaitEvent(st, m->ev_h1, 0) == hipSuccess, "snet_model_eval: stream ordering failed");
          if ((rc = snet_halo_reverse_accumulate(m->halo_user, g_h, L.dx, st))) return rc;
          if ((rc = run_linear(m, L.si1, g_h, gx_next, N, true, L.sc.present(), st))) return rc;
          std::swap(g_x, gx_next);
          continue;
        }
        if ((rc = gh_rows(0, NT))) return rc;
        if (has_halo)
          if ((rc = m->halo_rev(m->halo_user, g_h, NT, N, L.dx, stream))) {
            snet::set_error("snet_model_eval: reverse halo callback failed");
            return rc;
          }
        if (!tail && (rc = snet_radial_mlp_hidden_bwd(L.mlp_plan, emb, g_h2, E, g_emb, st))) return rc;
        // g_x = sc^T g_y, then SI1^T g_h accumulated onto it -- in THIS order in every branch and in engine.py: an accumulating
        // launch starts its accumulators from the rows already there, so the order is part of the result's last bits
        if (L.sc.present())
          if ((rc = run_linear(m, L.sc, g_y, gx_next, N, true, false, st))) return rc;
        if ((rc = run_linear(m, L.si1, g_h, gx_next, N, true, L.sc.present(), st))) return rc;
        std::swap(g_x, gx_next);
        continue;
      }
      if (!tail && (rc = snet_radial_mlp_hidden_bwd(L.mlp_plan, emb, g_h2, E, g_emb, st))) return rc;
      break;  // layer-0 inputs depend on species only
    }
    float *g_w = ov ? gw_buf[t & 1] : A.f((size_t)E * L.wn);
    if (ov && gw_busy[t & 1])  // the MLP reverse of layer t+2 read this buffer on the side stream
      SNET_REQUIRE(hipStreamWaitEvent(st, m->ev_bwd[t & 1], 0) == hipSuccess, "snet_model_eval: stream ordering failed");
    if ((rc = snet_conv_bwd_edge_vec(L.conv, saved[t].h, sh, dsh, saved[t].w, pairs ? w_row : nullptr, row_ptr, src, N,
                                     L.conv_scale, g_m, g_w,
                                     g_xe, g_vec, st)))
      return rc;
    if (ov) {  // only the final radial gradient needs g_emb: runs beside the rest of the reverse pass
      SNET_REQUIRE(hipEventRecord(m->ev_main, st) == hipSuccess && hipStreamWaitEvent(m->side, m->ev_main, 0) == hipSuccess,
                   "snet_model_eval: stream ordering failed");
      if ((rc = snet_radial_mlp_bwd(L.mlp_plan, emb, g_w, E, g_emb, m->side))) return rc;
      SNET_REQUIRE(hipEventRecord(m->ev_bwd[t & 1], m->side) == hipSuccess, "snet_model_eval: stream ordering failed");
      gw_busy[t & 1] = true;
    } else if ((rc = snet_radial_mlp_bwd(L.mlp_plan, emb, g_w, E, g_emb, st))) {
      return rc;
    }
    if (t == 0) break;  // layer-0 inputs depend on species only
    float *g_h = A.f((size_t)NT * L.dx);
    if ((rc = snet_segment_sum_rows(g_xe, col_ptr, eperm, NT, L.dx, g_h, st))) return rc;
    if (has_halo)
      if ((rc = m->halo_rev(m->halo_user, g_h, NT, N, L.dx, stream))) {
        snet::set_error("snet_model_eval: reverse halo callback failed");
        return rc;
      }
    // g_x (for the previous layer's gate output) = sc^T g_y + SI1^T g_h (same order as above)
    if (L.sc.present())
      if ((rc = run_linear(m, L.sc, g_y, gx_next, N, true, false, st))) return rc;
    if ((rc = run_linear(m, L.si1, g_h, gx_next, N, true, L.sc.present(), st))) return rc;
    std::swap(g_x, gx_next);
  }
  for (int b = 0; b < 2; ++b)
    if (gw_busy[b])
      SNET_REQUIRE(hipStreamWaitEvent(st, m->ev_bwd[b], 0) == hipSuccess, "snet_model_eval: stream ordering failed");
  if ((rc = snet_edge_embed_bwd(&ep, m->coeffs.data(), edge_vec, E, g_emb, nullptr, g_vec, 1, st))) return rc;
  A.off = mark2;
  float *F = forces ? forces : A.f((size_t)NT * 3);
  if ((rc = snet_edge_force(g_vec, edge_vec, row_ptr, col_ptr, eperm, NT, E, F, virial_atom, virial, st))) return rc;
  if (has_halo && m->fold_forces) {
    if ((rc = m->halo_rev(m->halo_user, F, NT, N, 3, stream))) return rc;
    if (virial_atom && (rc = m->halo_rev(m->halo_user, virial_atom, NT, N, 6, stream))) return rc;
  }
  return 0;
}
