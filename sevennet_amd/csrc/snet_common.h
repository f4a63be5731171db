// Internal helpers shared by the gfx950 kernels of libsnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "snet_hip.h"

namespace snet {

void set_error(const std::string &msg);

#define SNET_CHECK_LAUNCH(what)                                                    \
  do {                                                                             \
    hipError_t _e = hipGetLastError();                                             \
    if (_e != hipSuccess) {                                                        \
      snet::set_error(std::string(what) + ": " + hipGetErrorString(_e));           \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

#define SNET_REQUIRE(cond, msg)                                                    \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      snet::set_error(std::string(msg));                                           \
      return 2;                                                                    \
    }                                                                              \
  } while (0)

// ---- compiled tensor-product specialisations (filled by generated TUs) ------
struct ConvKernels {
  const char *tag;
  int dx, dout, nsh, wn, threads;
  // w_row (nullable): row of w read by edge e (edges of one undirected pair may share a row)
  void (*fwd)(const float *x, const float *sh, const float *w, const int32_t *w_row, const int32_t *row_ptr,
              const int32_t *src, int64_t n_dst, float scale, float *out, hipStream_t st);
  void (*bwd_edge)(const float *x, const float *sh, const float *w, const int32_t *w_row, const int32_t *row_ptr,
                   const int32_t *src, int64_t n_dst, float scale, const float *g_out, float *g_w, float *g_xe,
                   float *g_sh, hipStream_t st);
  void (*bwd_edge_vec)(const float *x, const float *sh, const float *dsh, const float *w, const int32_t *w_row,
                       const int32_t *row_ptr, const int32_t *src, int64_t n_dst, float scale, const float *g_out,
                       float *g_w, float *g_xe, float *g_vec, hipStream_t st);
  void (*bwd_node)(const float *sh, const float *w, const int32_t *w_row, const int32_t *col_ptr, const int32_t *eperm,
                   const int32_t *dst, int64_t n_src, float scale, const float *g_out, float *g_x,
                   hipStream_t st);
  // scalar-output shapes (every path (l, l -> 0)): the source-row gradient is a forward convolution of the TRANSPOSED
  // product (shape t_tag: x = the g_out row, out = g_x) with W2's column c scaled by t_col_scale[c]; t_dead = (offset,
  // length) ranges of g_x no path writes.  t_tag == nullptr: the shape has a non-scalar output.
  const char *t_tag;
  const float *t_col_scale;
  int t_ndead;
  const int *t_dead;
};
void register_conv(const ConvKernels *k);

// radial-MLP plan internals needed by the fused tensor-product kernels (snet_mlp.hip)
int mlp_plan_wn(const snet_mlp_plan *plan);


// ---- fused radial-weight + tensor-product kernels (generated convf_<tag>.hip) ------------------------
// The radial MLP's last layer w = h2 @ W2 is evaluated inside the tensor-product kernels on
// v_mfma_f32_16x16x32_bf16 (lane & 15 = edge, registers = channels), so neither w[E,wn] nor its
// gradient g_w[E,wn] exists in memory.  A kernel walks the weight columns in `n_sub` sub-steps of two
// 16-column tiles; `sub_cols[2 s + tp]` is the first weight column of tile tp of sub-step s (-1: padding).
// W2 reaches the kernels as a stream of pre-split MFMA A fragments in that order (snet_fused_plan).
// arguments of the reverse kernels' hidden-layer tail (g_h2 -> g_emb in the same kernel); g_emb == nullptr: the
// kernel stores g_h2 instead
struct FusedTail {
  const float *emb;  // [E, nb] radial basis values per edge
  float *g_emb;      // [E, nb] +=
  int nb, act;
  float cst;
  // precision mode 4 (fp16 terms): powers of two the host scaled W2 / W1 / W0 by before splitting them
  int w2_exp, w1_exp, w0_exp;
  // precision mode 4: max |x[row]| per source row and max |g_out[node]| per destination node (snet_row_absmax):
  // the bound the per-edge power-of-two scale of the fp16 operand g_w is derived from
  const float *x_max, *g_max;
  // optional output (b1 plug-in: the reference's autograd needs the gradient with respect to edge_attr itself): g_sh[E, nsh]
  // OVERWRITTEN with dE/dY of every edge; NULL in the MD hosts, which take dE/d(edge_vec) through dsh / g_vec instead
  float *g_sh;
};
struct FusedKernels {
  const char *tag;
  int dx, dout, nsh, wn;
  int n_sub;
  const int32_t *sub_cols;   // forward kernel's sub-step order
  int n_sub_b;
  const int32_t *sub_cols_b;  // reverse kernel's sub-step order (pairs a partnerless path's tiles across two channel tiles)
  // g_xe rows are written in the kernel's own chunk order: gxe_chunk[standard 16-channel chunk] = its position in the row
  // (dx / 16 entries); snet_segment_sum_rows_chunked undoes it while it sums
  const int32_t *gxe_chunk;
  // reverse pass of one tile list (snet_edge_tiles): g_xe[E,dx] (nullable), g_vec[E,3] +=, and either g_h2[E,64]
  // or (tail.g_emb set) the radial MLP's hidden layers reversed in the same kernel: g_emb[E,nb] +=
  // nt = precision mode of the in-kernel products: 1 / 2 / 3 bf16 terms per operand, 4 = two fp16 terms ("f16x3")
  void (*bwd)(int nt, const float *x, const float *sh, const float *dsh, const float *h2, const int32_t *w_row,
              const int32_t *row_ptr, const int32_t *src, const int32_t *tile_ptr, const int32_t *tile_node,
              int64_t n_tiles, const void *slabs, float scale, const float *g_out, float *g_xe, float *g_h2,
              float *g_vec, FusedTail tail, hipStream_t st);
  // forward: out[n_dst, dout]
  void (*fwd)(int nt, const float *x, const float *sh, const float *h2, const int32_t *w_row, const int32_t *row_ptr,
              const int32_t *src, int64_t n_dst, const void *slabs, float scale, float *out, int w2_exp, hipStream_t st);
  int tile_mode;  // work list of bwd: 0 = per-node tiles (snet_edge_tiles), 1 = packed tiles (snet_edge_tiles_packed)
};
void register_fused(const FusedKernels *k);
const FusedKernels *find_fused(const char *tag);
struct FusedRegistrar {
  explicit FusedRegistrar(const FusedKernels *k) { register_fused(k); }
};
// host copy of the radial MLP's pre-normalised last-layer weights W2'[64, wn] (row-major)
const float *mlp_plan_w2_host(const snet_mlp_plan *plan);
// W2'[64, wn] -> device stream of 1-KB fragment lines, per sub-step: [tile(2)][kstep(2)][term(nt)] (the
// 16-column tile as A/B operand of w = h2 @ W2) then [mtile(4)][term(nt)] (W2 rows as A operand of
// g_h2 = g_w @ W2^T over the sub-step's 32 columns).  Returns 0 on success.
// hidden layers of a split-precision radial-MLP plan (host copies, pre-normalised): W0'[nb, 64], W1'[64, 64]
struct MlpHidden {
  const float *w0, *w1;
  int nb, act;
  float cst;
};
MlpHidden mlp_plan_hidden(const snet_mlp_plan *plan);  // w0 == nullptr: the plan has no split-precision hidden layers
constexpr int FUSED_TAIL_FRAGS = 22;  // z1 (4) + z2 (8) + g_a1 (8) + g_emb (2) operand fragments, nt terms each
// `tail` (nullable): append the hidden-layer fragments the reverse kernels' g_h2 -> g_emb tail multiplies with.
// mode: 1 / 2 / 3 bf16 terms per value, or 4 = two fp16 terms of the value times 2^exps[i] (i = 0 W2, 1 W1, 2 W0)
int pack_fused_slabs(const float *w2, int wn, int n_sub, const int32_t *sub_cols, int mode, const MlpHidden *tail,
                     void **dev_out, int32_t (&exps)[3]);

// host-side fp16 helpers of the two-term ("f16x3") operand formats (snet_mlp.hip): round-to-nearest-even conversion with
// subnormals, its inverse, and the exponent k that puts max |v| 2^k into [2^13, 2^14)
uint16_t f16_rne(float f);
float f16_f(uint16_t h);
int f16_scale_exp(const float *v, size_t n);

struct ConvRegistrar {
  explicit ConvRegistrar(const ConvKernels *k) { register_conv(k); }
};

// ---- device helpers -----------------------------------------------------------
// Workgroups are dispatched round-robin over the 8 XCDs (block b runs on XCD b % 8), each with its own
// L2.  One-workgroup-per-node kernels map block b to node xcd_node(b, n) so that every XCD walks a
// CONTIGUOUS range of nodes: atoms that are close in index are close in space (MD codes keep them
// sorted), so the source rows and radial-weight rows their edges share are re-read from the same L2
// instead of being fetched into eight of them.  Bijection of [0, n).
__device__ __forceinline__ int xcd_node(unsigned b, unsigned n) {
  const unsigned k = b & 7u, j = b >> 3, q = n >> 3, r = n & 7u;
  return (int)(k * q + (k < r ? k : r) + j);
}

// 64-lane wavefront sum; every lane returns the total.
__device__ __forceinline__ float wave_sum(float v) {
  v += __shfl_xor(v, 32, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- cross-lane exchange at VALU speed (no LDS): gfx950 permlane swaps + DPP -------------------
// a + b where the "low" lanes (bit 5 / bit 4 of the lane id clear) end up with a_self + a_partner
// and the "high" lanes with b_self + b_partner, partner = lane ^ 32 (resp. ^ 16).
__device__ __forceinline__ float swap_add32(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// value of lane ^ MASK for MASK in {8, 4, 2, 1} (inside a 16-lane DPP row)
template <int MASK>
__device__ __forceinline__ float lane_xor(float v) {
  const int x = __float_as_int(v);
  if constexpr (MASK == 8) return __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false));  // row_ror:8
  if constexpr (MASK == 4) {
    int t = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0x5, false);  // banks 0,2 read lane+4 (row_shl:4)
    t = __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);      // banks 1,3 read lane-4 (row_shr:4)
    return __int_as_float(t);
  }
  if constexpr (MASK == 2) return __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false));                            // quad_perm [1,0,3,2]
}

// broadcast of lane K's value as a wave-uniform scalar (v_readlane_b32 -> SGPR operand).  Per-edge
// constants (spherical harmonics, their Jacobian) are fetched by ONE coalesced vector load -- lane k
// holds element k -- and read back through this, instead of one broadcast load per element.
template <int K>
__device__ __forceinline__ float lane_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), K));
}

// Activation ids (sevenn/_const.py:33-47): 0 silu, 1 tanh, 2 relu, 3 abs, 4 ssp (softplus - ln 2, sevenn/nn/activation.py:7),
// 5 sigmoid, 6 elu.  The e3nn `normalize2mom` constant of each travels beside the id (model_spec.ACT_CST).
constexpr int N_ACT = 7;
__device__ __forceinline__ float act_fwd(float z, int act) {
  switch (act) {
    case 0: return z / (1.0f + expf(-z));                                     // silu
    case 1: return tanhf(z);
    case 2: return fmaxf(z, 0.0f);                                            // relu
    case 3: return fabsf(z);
    case 4: return fmaxf(z, 0.0f) + log1pf(expf(-fabsf(z))) - 0.69314718055994531f;   // softplus(z) - ln 2, overflow-free form
    case 5: return 1.0f / (1.0f + expf(-z));                                  // sigmoid
    default: return z > 0.0f ? z : expm1f(z);                                 // elu (alpha = 1)
  }
}
// derivative of the activation wrt its pre-activation (relu / abs at 0: torch's convention, 0)
__device__ __forceinline__ float act_grad(float z, int act) {
  switch (act) {
    case 0: {
      const float s = 1.0f / (1.0f + expf(-z));
      return s * (1.0f + z * (1.0f - s));
    }
    case 1: {
      const float t = tanhf(z);
      return 1.0f - t * t;
    }
    case 2: return z > 0.0f ? 1.0f : 0.0f;
    case 3: return z > 0.0f ? 1.0f : (z < 0.0f ? -1.0f : 0.0f);
    case 4: return 1.0f / (1.0f + expf(-z));                                  // d softplus = sigmoid
    case 5: {
      const float s = 1.0f / (1.0f + expf(-z));
      return s * (1.0f - s);
    }
    default: return z > 0.0f ? 1.0f : expf(z);
  }
}
// activation value and derivative in one go, hardware exp2 / rcp (~1 ulp each) instead of the IEEE division and
// libm exp of act_fwd / act_grad (~25 instructions per call): for code that evaluates it per element of a tile.
// silu and tanh only (the fused reverse kernels' hidden-layer tail: snet_fused_plan_create leaves the tail off for the others)
__device__ __forceinline__ void act_both_fast(float z, int act, float &f, float &g) {
  if (act == 0) {  // silu: s = sigmoid(z); f = z s; g = s (1 + z (1 - s))
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
    f = z * s;
    g = s * fmaf(z, 1.0f - s, 1.0f);
  } else {
    f = tanhf(z);
    g = 1.0f - f * f;
  }
}
// activation value alone, same hardware exp2 / rcp form for silu (the hidden radial layers evaluate 128 of these per row: with
// libm exp and the IEEE division their kernel was bound by exactly that, 0.19 ms per launch); every other id: act_fwd
__device__ __forceinline__ float sigmoid_fast(float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z)); }
__device__ __forceinline__ float act_fwd_fast(float z, int act) {
  if (act == 0) return z * sigmoid_fast(z);
  return act_fwd(z, act);
}
// The gate kernels' forms (round 6): silu (0) and sigmoid (5) -- the scalar and gate activations of every SevenNet preset
// (sevenn/_const.py:33-47 defaults) -- on hardware exp2 / rcp, every other id through libm.  The id differs per LANE there (a lane
// owns four channels of one segment), so the libm branch is entered only by waves that hold such a segment.  With libm exp and the
// IEEE division for all ids the reverse gate kernel was bound by vector issue: 3 408 static instructions, 0.145 ms against 0.08 ms of
// memory traffic.
__device__ __forceinline__ float act_fwd_gate(float z, int act) {
  if (act == 0) return z * sigmoid_fast(z);
  if (act == 5) return sigmoid_fast(z);
  return act_fwd(z, act);
}
__device__ __forceinline__ float act_grad_gate(float z, int act) {
  if (act == 0 || act == 5) {
    const float s = sigmoid_fast(z);
    return act == 0 ? s * fmaf(z, 1.0f - s, 1.0f) : s * (1.0f - s);
  }
  return act_grad(z, act);
}
// value f and derivative g of the same pre-activation (one exponential for silu / sigmoid)
__device__ __forceinline__ void act_pair_gate(float z, int act, float &f, float &g) {
  if (act == 0 || act == 5) {
    const float s = sigmoid_fast(z);
    f = act == 0 ? z * s : s;
    g = act == 0 ? s * fmaf(z, 1.0f - s, 1.0f) : s * (1.0f - s);
  } else {
    f = act_fwd(z, act);
    g = act_grad(z, act);
  }
}

}  // namespace snet
