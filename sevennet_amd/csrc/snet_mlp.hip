// Fused radial MLP: e3nn FullyConnectedNet([nb, 64, 64, wn], act) as ONE
// kernel per direction on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
//   fwd:  emb[E,nb] -> w[E,wn]                 (z1, a1, z2, a2 never leave registers)
//   bwd:  g_w[E,wn], emb -> g_emb[E,nb] +=     (hidden activations recomputed, not stored)
//
// One wavefront owns 32 edges.  The hidden layers are computed TRANSPOSED (Z^T = W^T A^T) so the
// accumulator layout -- lane&31 = edge, registers = hidden units -- is directly the B operand of
// the next transposed product and the A operand of the final (non-transposed) product, provided
// the reduction index is visited in the order the accumulator registers hold it:
//     step s = 16*t + r  ->  k(s, half) = 32*t + (r&3) + 8*(r>>2) + 4*half     (half = lane>>5)
// (a contraction is invariant under a permutation of its summation index).  No LDS round trip, no
// cross-lane traffic between the layers.  W2 (64 x wn, up to 450 KB) streams from L2 as B fragments
// (one coalesced 128-B row segment per half-wave per step); the output tile is stored as 128-B row
// segments of w.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <initializer_list>
#include <vector>

#include "snet_common.h"
#include "snet_split.h"

namespace {

using namespace snet;  // f32x16, f32x4, u32x4, Split3, split8, mfma6, zero16 (snet_split.h)
constexpr int H = 64;  // hidden width of both hidden layers

__device__ __forceinline__ int krow(int s, int half) { return 32 * (s >> 4) + (s & 3) + 8 * ((s & 15) >> 2) + 4 * half; }

// z1^T, z2^T for the wave's 32 edges.  W1s = W1' in LDS, row-major [h][h'].
__device__ __forceinline__ void hidden_forward(const float *__restrict__ emb, int64_t e_lane, bool e_ok, int nb,
                                               const float *__restrict__ W0, const float *W1s, int act, float cst,
                                               int lane, f32x16 (&z1)[2], f32x16 (&a1)[2], f32x16 (&z2)[2]) {
  const int half = lane >> 5, li = lane & 31;
  z1[0] = zero16();
  z1[1] = zero16();
  // L1: Z1^T[h, e] = sum_k W0'[k][h] emb[e][k]
  for (int s = 0; 2 * s < nb; ++s) {
    const int k = 2 * s + half;
    const float b = (e_ok && k < nb) ? emb[e_lane * nb + k] : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float a = (k < nb) ? W0[k * H + 32 * t + li] : 0.f;
      z1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, z1[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) a1[t][r] = snet::act_fwd(z1[t][r], act) * cst;
  // L2: Z2^T[h', e] = sum_h W1'[h][h'] a1[e][h]
  z2[0] = zero16();
  z2[1] = zero16();
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = krow(16 * t + r, half);
      const float b = a1[t][r];
#pragma unroll
      for (int to = 0; to < 2; ++to) {
        const float a = W1s[k * H + 32 * to + li];
        z2[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, z2[to], 0, 0, 0);
      }
    }
}

__global__ __launch_bounds__(256) void radial_mlp_fwd_kernel(const float *__restrict__ emb, int64_t E, int nb, int wn,
                                                             const float *__restrict__ W0,
                                                             const float *__restrict__ W1,
                                                             const float *__restrict__ W2, int act, float cst,
                                                             float *__restrict__ w_out) {
  __shared__ float W1s[H * H];
  for (int i = threadIdx.x; i < H * H; i += 256) W1s[i] = W1[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, li = lane & 31;
  const int64_t e0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  if (e0 >= E) return;
  const int64_t e_lane = e0 + li;
  const bool e_ok = e_lane < E;

  f32x16 z1[2], a1[2], z2[2];
  hidden_forward(emb, e_lane, e_ok, nb, W0, W1s, act, cst, lane, z1, a1, z2);
  f32x16 a2[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) a2[t][r] = snet::act_fwd(z2[t][r], act) * cst;

  // L3: W[e, ch] = sum_h a2[e][h] W2'[h][ch], one 32-column tile at a time, B fragments prefetched
  const int n_tiles = (wn + 31) >> 5;
  float bcur[32], bnxt[32];
  {
    const int ch = li;
#pragma unroll
    for (int s = 0; s < 32; ++s) bcur[s] = (ch < wn) ? W2[(int64_t)krow(s, half) * wn + ch] : 0.f;
  }
  for (int c = 0; c < n_tiles; ++c) {
    const int chn = 32 * (c + 1) + li;
    if (c + 1 < n_tiles) {
#pragma unroll
      for (int s = 0; s < 32; ++s) bnxt[s] = (chn < wn) ? W2[(int64_t)krow(s, half) * wn + chn] : 0.f;
    }
    f32x16 acc = zero16();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[t][r], bcur[16 * t + r], acc, 0, 0, 0);
    const int ch = 32 * c + li;
    if (ch < wn) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t e = e0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (e < E) w_out[e * wn + ch] = acc[r];
      }
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) bcur[s] = bnxt[s];
  }
}

// ---- backward -------------------------------------------------------------------------------
// LDS plan (50 KB -> 3 workgroups per CU):
//   W1p   [64][65]      W1' padded: conflict-free both as A[i=h'][k=h] (forward recompute) and as
//                       A[i=h][k=h'] (G_a1), so no transposed copy is needed
//   gws   4 x [32][33]  per-wave transpose tile of g_w (32 edges x 32 channels)
//   w2t   2 x [32][64]  double-buffered 32-channel slab of W2^T shared by the 4 waves
constexpr int W1P = 65;
constexpr int GW_STRIDE = 33;
constexpr int GW_TILE = 32 * GW_STRIDE;
constexpr int CH = 32;  // channels per chunk

__global__ __launch_bounds__(256) void radial_mlp_bwd_kernel(const float *__restrict__ emb,
                                                             const float *__restrict__ g_w, int64_t E, int nb, int wn,
                                                             const float *__restrict__ W0,
                                                             const float *__restrict__ W1,
                                                             const float *__restrict__ W2T, int act, float cst,
                                                             float *__restrict__ g_emb) {
  __shared__ float W1p[H * W1P];
  __shared__ float gws[4 * GW_TILE];
  __shared__ float w2t[2][CH * H];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, li = lane & 31;
  const int64_t e0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  const bool wave_ok = e0 < E;  // every wave stays in the block-wide barriers
  const int64_t e_lane = e0 + li;
  const bool e_ok = wave_ok && e_lane < E;
  float *tile = gws + wave * GW_TILE;
  const bool vec_ok = (wn & 3) == 0;
  const int srow = lane >> 3, scol = 4 * (lane & 7);

  f32x4 st_g[4], st_w[2];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // g_w[e0 + 8i + srow][c0 + scol .. +4)
      const int64_t e = e0 + 8 * i + srow;
      f32x4 q = {0.f, 0.f, 0.f, 0.f};
      if (wave_ok && e < E && c0 + scol < wn) {
        const float *p = g_w + e * wn + c0 + scol;
        if (vec_ok && c0 + scol + 3 < wn) {
          q = *reinterpret_cast<const f32x4 *>(p);
        } else {
          q.x = p[0];
          if (c0 + scol + 1 < wn) q.y = p[1];
          if (c0 + scol + 2 < wn) q.z = p[2];
          if (c0 + scol + 3 < wn) q.w = p[3];
        }
      }
      st_g[i] = q;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // W2T[c0 + row][0..64): 512 float4 per slab, 2 per thread
      const int f = tid + 256 * i;
      const int row = f >> 4, col = 4 * (f & 15);
      st_w[i] = (c0 + row < wn) ? *reinterpret_cast<const f32x4 *>(W2T + (int64_t)(c0 + row) * H + col)
                                : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float *q = tile + (8 * i + srow) * GW_STRIDE + scol;
      q[0] = st_g[i].x; q[1] = st_g[i].y; q[2] = st_g[i].z; q[3] = st_g[i].w;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i;
      *reinterpret_cast<f32x4 *>(&w2t[buf][4 * f]) = st_w[i];
    }
  };

  load_chunk(0);
  for (int i = tid; i < H * H; i += 256) W1p[(i >> 6) * W1P + (i & 63)] = W1[i];
  store_chunk(0);
  __syncthreads();

  // recompute z1, z2 (transposed layout: lane&31 = edge, registers = hidden units)
  f32x16 z1[2], z2[2];
  {
    f32x16 a1[2];
    z1[0] = zero16();
    z1[1] = zero16();
    for (int s = 0; 2 * s < nb; ++s) {
      const int k = 2 * s + half;
      const float b = (e_ok && k < nb) ? emb[e_lane * nb + k] : 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float a = (k < nb) ? W0[k * H + 32 * t + li] : 0.f;
        z1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, z1[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[t][r] = snet::act_fwd(z1[t][r], act) * cst;
    z2[0] = zero16();
    z2[1] = zero16();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = krow(16 * t + r, half);
#pragma unroll
        for (int to = 0; to < 2; ++to)
          z2[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(W1p[k * W1P + 32 * to + li], a1[t][r], z2[to], 0, 0, 0);
      }
  }

  // G_a2^T[k, e] = sum_ch W2'[k][ch] g_w[e][ch]: A = w2t slab [ch][k], B = transposed g_w tile
  f32x16 ga2[2];
  ga2[0] = zero16();
  ga2[1] = zero16();
  int buf = 0;
  for (int c0 = 0; c0 < wn; c0 += CH) {
    const bool more = c0 + CH < wn;
    if (more) load_chunk(c0 + CH);  // global -> registers, in flight during the MFMAs below
    const float *wt = w2t[buf];
#pragma unroll
    for (int s = 0; s < CH / 2; ++s) {
      const float b = tile[li * GW_STRIDE + 2 * s + half];
#pragma unroll
      for (int t = 0; t < 2; ++t)
        ga2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[(2 * s + half) * H + 32 * t + li], b, ga2[t], 0, 0, 0);
    }
    if (more) store_chunk(buf ^ 1);  // tile: wave-private (in-order LDS); w2t[buf^1]: last read before
                                     // the previous barrier
    __syncthreads();
    buf ^= 1;
  }
  // g_z2 = g_a2 * cst * act'(z2)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) ga2[t][r] *= cst * snet::act_grad(z2[t][r], act);
  // G_a1^T[h, e] = sum_h' W1'[h][h'] g_z2[e][h']     (A[i=h][k=h'] = W1p[h][h'])
  f32x16 ga1[2];
  ga1[0] = zero16();
  ga1[1] = zero16();
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = krow(16 * t + r, half);
#pragma unroll
      for (int to = 0; to < 2; ++to)
        ga1[to] = __builtin_amdgcn_mfma_f32_32x32x2f32(W1p[(32 * to + li) * W1P + k], ga2[t][r], ga1[to], 0, 0, 0);
    }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) ga1[t][r] *= cst * snet::act_grad(z1[t][r], act);
  // G_emb^T[k0, e] = sum_h W0'[k0][h] g_z1[e][h]     (rows k0 < nb <= 32)
  f32x16 ge = zero16();
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = krow(16 * t + r, half);
      const float a = (li < nb) ? W0[li * H + k] : 0.f;
      ge = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ga1[t][r], ge, 0, 0, 0);
    }
  if (e_ok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k0 = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (k0 < nb) g_emb[e_lane * nb + k0] += ge[r];
    }
  }
}

}  // namespace


// =================================================================================================
// Split-precision variant: every fp32 operand x is written x = x1 + x2 + x3 with bf16 x1, x2, x3
// (24 mantissa bits), and a*b is evaluated as the six bf16 products of order <= 2
// (a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1), each exact in the fp32 accumulator of
// v_mfma_f32_32x32x16_bf16.  Dropped terms are O(2^-24) relative, i.e. fp32 rounding class, while
// the matrix pipe runs 16x faster per flop than with fp32 inputs: 6/16 of the fp32-MFMA time.
// Weights are split and packed into per-lane MFMA fragments once at plan creation.
// Fragment k-slots follow the same accumulator-order permutation as the fp32 kernels:
//     MFMA q (16 reduction slots), half = lane>>5, slot i8:  r = 8*(q&1) + i8
//     hidden index kmap(q, half, i8) = 32*(q>>1) + (r&3) + 8*(r>>2) + 4*half
// =================================================================================================
namespace {

// fragment index helper: packed arrays are [..][term(3)][lane(64)] of uint4
__device__ __forceinline__ void load_frag3(const u32x4 *__restrict__ base, int frag, int lane, bf16x8 (&out)[3]) {
#pragma unroll
  for (int t = 0; t < 3; ++t) out[t] = as_bf16x8(base[(frag * 3 + t) * 64 + lane]);
}

// the activation id as a compile-time constant (ACT >= 0: no per-element scalar branch, and the compiler may move the matrix
// instructions of one k step under the vector work of the next) or read at run time (ACT < 0)
template <int ACT>
__device__ __forceinline__ float act_sel(float z, int act) {
  if constexpr (ACT >= 0) return snet::act_fwd_fast(z, ACT);
  else return snet::act_fwd_fast(z, act);
}

// z1 (fp32 MFMA, K = nb) and z2 (split MFMA) in the transposed layout.  NB > 0: the number of basis functions at compile time.
template <int ACT = -1, int NB = 0>
__device__ __forceinline__ void hidden_forward_split(const float *__restrict__ emb, int64_t e_lane, bool e_ok, int nb,
                                                     const float *__restrict__ W0, const u32x4 *__restrict__ W1A,
                                                     int act, float cst, int lane, f32x16 (&z1)[2],
                                                     f32x16 (&z2)[2]) {
  const int half = lane >> 5, li = lane & 31;
  z1[0] = zero16();
  z1[1] = zero16();
  auto l1_step = [&](int s, int n_basis) {
    const int k = 2 * s + half;
    const float b = (e_ok && k < n_basis) ? emb[e_lane * n_basis + k] : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float a = (k < n_basis) ? W0[k * H + 32 * t + li] : 0.f;
      z1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, z1[t], 0, 0, 0);
    }
  };
  if constexpr (NB > 0) {
#pragma unroll
    for (int s = 0; 2 * s < NB; ++s) l1_step(s, NB);
  } else {
    for (int s = 0; 2 * s < nb; ++s) l1_step(s, nb);
  }
  z2[0] = zero16();
  z2[1] = zero16();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = act_sel<ACT>(z1[q >> 1][8 * (q & 1) + i], act) * cst;
    const Split3 b = split8(v);
#pragma unroll
    for (int to = 0; to < 2; ++to) {
      bf16x8 a[3];
      load_frag3(W1A, to * 4 + q, lane, a);
      z2[to] = mfma6(a, b, z2[to]);
    }
  }
}

constexpr int W2B_TILE_U4 = 4 * 3 * 64;  // uint4 per 32-column tile of W2B: [q(4)][term(3)][lane]

__global__ __launch_bounds__(256, 2) void radial_mlp_fwd_split_kernel(const float *__restrict__ emb, int64_t E, int nb,
                                                                   int wn, const float *__restrict__ W0,
                                                                   const u32x4 *__restrict__ W1A,
                                                                   const u32x4 *__restrict__ W2B, int act, float cst,
                                                                   float *__restrict__ w_out) {
  // W2 fragments are shared by the 4 waves: one 12-KB slab per 32-column tile, double buffered
  // (cuts the L2 -> CU fragment traffic 4x relative to per-wave loads)
  __shared__ u32x4 slab[2][W2B_TILE_U4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, li = lane & 31;
  const int64_t e0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  const bool wave_ok = e0 < E;  // all waves stay for the block barriers
  const int64_t e_lane = e0 + li;
  const bool e_ok = wave_ok && e_lane < E;
  const int n_tiles = (wn + 31) >> 5;
  u32x4 st[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) slab[0][tid + 256 * i] = W2B[tid + 256 * i];
  __syncthreads();
  f32x16 z1[2], z2[2];
  hidden_forward_split(emb, e_ok ? e_lane : 0, e_ok, nb, W0, W1A, act, cst, lane, z1, z2);
  Split3 a2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = snet::act_fwd_fast(z2[q >> 1][8 * (q & 1) + i], act) * cst;
    a2[q] = split8(v);
  }
  int buf = 0;
  for (int c = 0; c < n_tiles; ++c) {
    const bool more = c + 1 < n_tiles;
    if (more) {
#pragma unroll
      for (int i = 0; i < 3; ++i) st[i] = W2B[(int64_t)(c + 1) * W2B_TILE_U4 + tid + 256 * i];
    }
    f32x16 acc = zero16();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bf16x8 b[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) b[t] = as_bf16x8(slab[buf][(q * 3 + t) * 64 + lane]);
      acc = mfma6(a2[q], b, acc);
    }
    const int ch = 32 * c + li;
    if (wave_ok && ch < wn) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t e = e0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (e < E) w_out[e * wn + ch] = acc[r];
      }
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < 3; ++i) slab[buf ^ 1][tid + 256 * i] = st[i];
    }
    __syncthreads();
    buf ^= 1;
  }
}

// hidden activations only: h2[E,64] = act(act(emb W0) cst W1) cst -- the operand of the radial MLP's last
// layer when that layer is fused into the tensor-product kernels (generated conv_ffwd_*).
// Round 6: ALL interaction layers' hidden activations in one launch (they share the edge embedding and differ in weights only).
// At an eighth of the benchmark cell (340 k edges) five launches of 2 660 workgroups cost 55 us each against 24 us of work;
// one launch with five times the work per wave removes four dependent dispatches from the step.  The activation id and the
// basis count are template constants for the common case (silu, 8 Bessel functions): with the id read at run time every one of
// the 128 activations per edge sat behind its own scalar compare-and-branch, and the kernel was bound by vector issue, not by
// its 50 matrix instructions (19 241 static instructions, `tools/isa_census.py snet_mlp radial_mlp_hidden`).
constexpr int MAX_HIDDEN_LAYERS = 8;
struct HiddenLayers {
  const float *W0[MAX_HIDDEN_LAYERS];
  const u32x4 *W1A[MAX_HIDDEN_LAYERS];
  float *h2[MAX_HIDDEN_LAYERS];
  float cst[MAX_HIDDEN_LAYERS];
  int act[MAX_HIDDEN_LAYERS];
  int n;
};

template <int ACT, int NB>
__global__ __launch_bounds__(256, 2) void radial_mlp_hidden_layers_kernel(const float *__restrict__ emb, int64_t E, int nb,
                                                                       HiddenLayers P) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, li = lane & 31;
  const int64_t e_lane = ((int64_t)blockIdx.x * 4 + wave) * 32 + li;
  const bool e_ok = e_lane < E;
  for (int l = 0; l < P.n; ++l) {
    const int act = P.act[l];
    const float cst = P.cst[l];
    f32x16 z1[2], z2[2];
    hidden_forward_split<ACT, NB>(emb, e_ok ? e_lane : 0, e_ok, nb, P.W0[l], P.W1A[l], act, cst, lane, z1, z2);
    if (!e_ok) continue;
    float *__restrict__ h2 = P.h2[l];
    // lane (li, half) holds hidden units 32t + (r&3) + 8(r>>2) + 4 half of edge li: 16-byte groups
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = act_sel<ACT>(z2[t][4 * g + i], act) * cst;
        *reinterpret_cast<f32x4 *>(h2 + e_lane * H + 32 * t + 8 * g + 4 * half) = v;
      }
  }
}

#ifndef SNET_MLP_BWD_OCC
#define SNET_MLP_BWD_OCC 3
#endif
#ifndef SNET_MLP_BWD_WAVES
#define SNET_MLP_BWD_WAVES 4   // wavefronts (32 edges each) sharing one W2 slab per workgroup
#endif
constexpr int GS_STRIDE = 36;               // g_w tile row stride (floats): conflict-free 16-B column reads
constexpr int GS_TILE = 32 * GS_STRIDE;
constexpr int SLAB_U4 = 2 * 2 * 3 * 64;     // uint4 per 32-channel slab of W2A: [step(2)][tile(2)][term(3)][lane]

__global__ __launch_bounds__(64 * SNET_MLP_BWD_WAVES, SNET_MLP_BWD_OCC) void radial_mlp_bwd_split_kernel(
    const float *__restrict__ emb, const float *__restrict__ g_w, int64_t E, int nb, int wn,
    const float *__restrict__ W0, const u32x4 *__restrict__ W1A, const u32x4 *__restrict__ W2A,
    const u32x4 *__restrict__ W1A2, const u32x4 *__restrict__ W0A, int act, float cst, float *__restrict__ g_emb) {
  constexpr int NWV = SNET_MLP_BWD_WAVES, NTH = 64 * NWV, NSL = (768 + NTH - 1) / NTH;  // slab = 768 uint4
  __shared__ float gws[NWV * GS_TILE];
  __shared__ u32x4 slab[2][SLAB_U4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, li = lane & 31;
  const int64_t e0 = ((int64_t)blockIdx.x * NWV + wave) * 32;
  const bool wave_ok = e0 < E;
  const int64_t e_lane = e0 + li;
  const bool e_ok = wave_ok && e_lane < E;
  float *tile = gws + wave * GS_TILE;
  const bool vec_ok = (wn & 3) == 0;
  const int srow = lane >> 3, scol = 4 * (lane & 7);
  const int n_chunks = (wn + CH - 1) / CH;

  // two chunks in flight in registers (HBM latency exceeds one 24-MFMA chunk): stage[ck & 1]
  f32x4 st_g[2][4];
  u32x4 st_w[2][NSL];
  auto load_chunk = [&](int ck, int sl) {
    const int c0 = ck * CH;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t e = e0 + 8 * i + srow;
      f32x4 q = {0.f, 0.f, 0.f, 0.f};
      if (wave_ok && e < E && c0 + scol < wn) {
        const float *p = g_w + e * wn + c0 + scol;
        if (vec_ok && c0 + scol + 3 < wn) {
          q = *reinterpret_cast<const f32x4 *>(p);
        } else {
          q.x = p[0];
          if (c0 + scol + 1 < wn) q.y = p[1];
          if (c0 + scol + 2 < wn) q.z = p[2];
          if (c0 + scol + 3 < wn) q.w = p[3];
        }
      }
      st_g[sl][i] = q;
    }
#pragma unroll
    for (int i = 0; i < NSL; ++i)
      if (tid + NTH * i < 768) st_w[sl][i] = W2A[(int64_t)ck * SLAB_U4 + tid + NTH * i];
  };
  auto store_chunk = [&](int buf, int sl) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4 *>(tile + (8 * i + srow) * GS_STRIDE + scol) = st_g[sl][i];
#pragma unroll
    for (int i = 0; i < NSL; ++i)
      if (tid + NTH * i < 768) slab[buf][tid + NTH * i] = st_w[sl][i];
  };

  load_chunk(0, 0);
  if (n_chunks > 1) load_chunk(1, 1);
  store_chunk(0, 0);
  __syncthreads();

  // G_a2^T[k, e] = sum_ch W2'[k][ch] g_w[e][ch]   (the hidden activations are recomputed AFTER this
  // loop: 64 fewer live VGPRs while g_w streams through, i.e. one more resident workgroup per CU)
  f32x16 ga2[2];
  ga2[0] = zero16();
  ga2[1] = zero16();
  // iteration ck: (start) refill the register stage that held chunk ck with chunk ck+2;
  // (end) move chunk ck+1 from its stage into the LDS tile / the other slab buffer.
#define SNET_CHUNK_BODY(ck, PAR) /* PAR = ck & 1 as a literal: register stages stay statically indexed */   \
  {                                                                                                          \
    if ((ck) + 2 < n_chunks) load_chunk((ck) + 2, PAR);                                                      \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                          \
      float v[8];                                                                                            \
      const f32x4 lo4 = *reinterpret_cast<const f32x4 *>(tile + li * GS_STRIDE + 16 * s + 8 * half);       \
      const f32x4 hi4 = *reinterpret_cast<const f32x4 *>(tile + li * GS_STRIDE + 16 * s + 8 * half + 4);   \
      v[0] = lo4.x; v[1] = lo4.y; v[2] = lo4.z; v[3] = lo4.w;                                                \
      v[4] = hi4.x; v[5] = hi4.y; v[6] = hi4.z; v[7] = hi4.w;                                                \
      const Split3 b = split8(v);                                                                            \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                        \
        bf16x8 a[3];                                                                                         \
        _Pragma("unroll") for (int tm = 0; tm < 3; ++tm)                                                     \
            a[tm] = as_bf16x8(slab[PAR][((s * 2 + t) * 3 + tm) * 64 + lane]);                                \
        ga2[t] = mfma6(a, b, ga2[t]);                                                                        \
      }                                                                                                      \
    }                                                                                                        \
    if ((ck) + 1 < n_chunks) store_chunk((PAR) ^ 1, (PAR) ^ 1);                                              \
    __syncthreads();                                                                                         \
  }
  for (int ck = 0; ck < n_chunks; ck += 2) {
    SNET_CHUNK_BODY(ck, 0)
    if (ck + 1 < n_chunks) SNET_CHUNK_BODY(ck + 1, 1)
  }
#undef SNET_CHUNK_BODY
  f32x16 z1[2], z2[2];
  hidden_forward_split(emb, e_ok ? e_lane : 0, e_ok, nb, W0, W1A, act, cst, lane, z1, z2);
  // g_z2, then G_a1^T[h, e] = sum_h' W1'[h][h'] g_z2[e][h']
  f32x16 ga1[2];
  ga1[0] = zero16();
  ga1[1] = zero16();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 8 * (q & 1) + i;
      v[i] = ga2[q >> 1][r] * cst * snet::act_grad(z2[q >> 1][r], act);
    }
    const Split3 b = split8(v);
#pragma unroll
    for (int to = 0; to < 2; ++to) {
      bf16x8 a[3];
      load_frag3(W1A2, to * 4 + q, lane, a);
      ga1[to] = mfma6(a, b, ga1[to]);
    }
  }
  // g_z1, then G_emb^T[k0, e] = sum_h W0'[k0][h] g_z1[e][h]
  f32x16 ge = zero16();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 8 * (q & 1) + i;
      v[i] = ga1[q >> 1][r] * cst * snet::act_grad(z1[q >> 1][r], act);
    }
    const Split3 b = split8(v);
    bf16x8 a[3];
    load_frag3(W0A, q, lane, a);
    ge = mfma6(a, b, ge);
  }
  if (e_ok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k0 = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (k0 < nb) g_emb[e_lane * nb + k0] += ge[r];
    }
  }
}


// reverse of the two hidden layers only: g_h2[E,64] = dE/d(h2) (produced by the fused tensor-product reverse
// kernels, which contract g_w with W2^T on the fly) -> g_emb[E,nb] +=
__global__ __launch_bounds__(256, 2) void radial_mlp_hidden_bwd_split_kernel(
    const float *__restrict__ emb, const float *__restrict__ g_h2, int64_t E, int nb, const float *__restrict__ W0,
    const u32x4 *__restrict__ W1A, const u32x4 *__restrict__ W1A2, const u32x4 *__restrict__ W0A, int act, float cst,
    float *__restrict__ g_emb) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, li = lane & 31;
  const int64_t e_lane = ((int64_t)blockIdx.x * 4 + wave) * 32 + li;
  const bool e_ok = e_lane < E;
  // transposed layout: lane & 31 = edge, registers = hidden units 32 t + (r & 3) + 8 (r >> 2) + 4 half
  f32x16 ga2[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (e_ok) v = *reinterpret_cast<const f32x4 *>(g_h2 + e_lane * H + 32 * t + 8 * g + 4 * half);
#pragma unroll
      for (int i = 0; i < 4; ++i) ga2[t][4 * g + i] = v[i];
    }
  f32x16 z1[2], z2[2];
  hidden_forward_split(emb, e_ok ? e_lane : 0, e_ok, nb, W0, W1A, act, cst, lane, z1, z2);
  f32x16 ga1[2];
  ga1[0] = zero16();
  ga1[1] = zero16();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 8 * (q & 1) + i;
      v[i] = ga2[q >> 1][r] * cst * snet::act_grad(z2[q >> 1][r], act);
    }
    const Split3 b = split8(v);
#pragma unroll
    for (int to = 0; to < 2; ++to) {
      bf16x8 a[3];
      load_frag3(W1A2, to * 4 + q, lane, a);
      ga1[to] = mfma6(a, b, ga1[to]);
    }
  }
  f32x16 ge = zero16();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 8 * (q & 1) + i;
      v[i] = ga1[q >> 1][r] * cst * snet::act_grad(z1[q >> 1][r], act);
    }
    const Split3 b = split8(v);
    bf16x8 a[3];
    load_frag3(W0A, q, lane, a);
    ge = mfma6(a, b, ge);
  }
  if (e_ok) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k0 = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (k0 < nb) g_emb[e_lane * nb + k0] += ge[r];
    }
  }
}

}  // namespace

// ---- plan: device copies of the (pre-normalised) weights, fp32 and split/packed -----------------
struct snet_mlp_plan {
  int nb, wn, act, mode;
  float cst;
  float *W0 = nullptr, *W1 = nullptr, *W2 = nullptr, *W2T = nullptr;               // fp32 mode
  u32x4 *W1A = nullptr, *W2B = nullptr, *W2A = nullptr, *W1A2 = nullptr, *W0A = nullptr;  // split mode
  std::vector<float> w2_host;  // W2'[64, wn]: source of the fused tensor-product kernels' fragment stream
  std::vector<float> w0_host, w1_host;  // W0'[nb, 64], W1'[64, 64]: the fused reverse kernels' hidden-layer tail
};

namespace {

inline uint16_t bf16_rne(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float bf16_f(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline void split3(float x, uint16_t (&o)[3]) {
  o[0] = bf16_rne(x);
  const float r1 = x - bf16_f(o[0]);
  o[1] = bf16_rne(r1);
  o[2] = bf16_rne(r1 - bf16_f(o[1]));
}
inline int kmap(int q, int half, int i8) {
  const int r = 8 * (q & 1) + i8;
  return 32 * (q >> 1) + (r & 3) + 8 * (r >> 2) + 4 * half;
}
// frags[(frag*3 + term)*64 + lane][8] <- value(frag, lane, i8)
template <class F>
std::vector<uint16_t> pack_frags(int n_frag, F value) {
  std::vector<uint16_t> out((size_t)n_frag * 3 * 64 * 8);
  for (int f = 0; f < n_frag; ++f)
    for (int lane = 0; lane < 64; ++lane)
      for (int i = 0; i < 8; ++i) {
        uint16_t s[3];
        split3(value(f, lane, i), s);
        for (int t = 0; t < 3; ++t) out[(((size_t)f * 3 + t) * 64 + lane) * 8 + i] = s[t];
      }
  return out;
}
template <class T>
int upload(const std::vector<T> &h, void **dev) {
  if (hipMalloc(dev, h.size() * sizeof(T)) != hipSuccess) return 1;
  return hipMemcpy(*dev, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess;
}

}  // namespace

extern "C" int snet_radial_mlp_plan_create(int32_t nb, int32_t h1, int32_t h2, int32_t wn, const float *W0_host,
                                           const float *W1_host, const float *W2_host, int32_t act, float cst,
                                           int32_t mode, snet_mlp_plan **plan) {
  SNET_REQUIRE(plan != nullptr && W0_host && W1_host && W2_host, "snet_radial_mlp_plan_create: null argument");
  SNET_REQUIRE(h1 == H && h2 == H, "snet_radial_mlp_plan_create: fused kernels need hidden widths [64, 64]");
  SNET_REQUIRE(nb >= 1 && nb <= 32 && wn >= 1, "snet_radial_mlp_plan_create: need 1 <= n_basis <= 32, wn >= 1");
  SNET_REQUIRE(act >= 0 && act < snet::N_ACT, "snet_radial_mlp_plan_create: unknown activation");
  SNET_REQUIRE(mode == 0 || mode == 1, "snet_radial_mlp_plan_create: mode 0 (fp32 MFMA) or 1 (bf16 x6 split)");
  auto *p = new snet_mlp_plan;
  p->nb = nb; p->wn = wn; p->act = act; p->mode = mode; p->cst = cst;
  const std::vector<float> w0(W0_host, W0_host + (size_t)nb * H), w1(W1_host, W1_host + (size_t)H * H),
      w2(W2_host, W2_host + (size_t)H * wn);
  int bad = upload(w0, (void **)&p->W0);
  if (mode == 0) {
    std::vector<float> w2t((size_t)wn * H);
    for (int k = 0; k < H; ++k)
      for (int c = 0; c < wn; ++c) w2t[(size_t)c * H + k] = w2[(size_t)k * wn + c];
    bad |= upload(w1, (void **)&p->W1) | upload(w2, (void **)&p->W2) | upload(w2t, (void **)&p->W2T);
  } else {
    const int n_tiles = (wn + 31) / 32, n_chunks = (wn + CH - 1) / CH;
    // forward L2:  A[i = h'][k = h] = W1'[h][h'], frag = to*4 + q
    bad |= upload(pack_frags(8, [&](int f, int lane, int i) {
                    const int to = f >> 2, q = f & 3;
                    return w1[(size_t)kmap(q, lane >> 5, i) * H + 32 * to + (lane & 31)];
                  }), (void **)&p->W1A);
    // forward L3:  B[k = h][j = ch] = W2'[h][ch], frag = c*4 + q
    bad |= upload(pack_frags(n_tiles * 4, [&](int f, int lane, int i) {
                    const int c = f >> 2, q = f & 3, ch = 32 * c + (lane & 31);
                    return ch < wn ? w2[(size_t)kmap(q, lane >> 5, i) * wn + ch] : 0.f;
                  }), (void **)&p->W2B);
    // backward G_a2:  A[i = k][kk = ch] = W2'[k][ch], frag = (chunk*2 + step)*2 + tile, ch natural order
    bad |= upload(pack_frags(n_chunks * 4, [&](int f, int lane, int i) {
                    const int t = f & 1, s = (f >> 1) & 1, ck = f >> 2;
                    const int ch = ck * CH + 16 * s + 8 * (lane >> 5) + i;
                    return ch < wn ? w2[(size_t)(32 * t + (lane & 31)) * wn + ch] : 0.f;
                  }), (void **)&p->W2A);
    // backward G_a1:  A[i = h][k = h'] = W1'[h][h'], frag = to*4 + q
    bad |= upload(pack_frags(8, [&](int f, int lane, int i) {
                    const int to = f >> 2, q = f & 3;
                    return w1[(size_t)(32 * to + (lane & 31)) * H + kmap(q, lane >> 5, i)];
                  }), (void **)&p->W1A2);
    // backward G_emb:  A[i = k0][k = h] = W0'[k0][h], frag = q
    bad |= upload(pack_frags(4, [&](int f, int lane, int i) {
                    const int k0 = lane & 31;
                    return k0 < nb ? w0[(size_t)k0 * H + kmap(f, lane >> 5, i)] : 0.f;
                  }), (void **)&p->W0A);
  }
  p->w2_host = w2;
  p->w0_host = w0;
  p->w1_host = w1;
  if (bad) {
    snet::set_error("snet_radial_mlp_plan_create: device allocation / upload failed");
    delete p;
    return 1;
  }
  *plan = p;
  return 0;
}

extern "C" void snet_radial_mlp_plan_destroy(snet_mlp_plan *p) {
  if (!p) return;
  for (void *d : {(void *)p->W0, (void *)p->W1, (void *)p->W2, (void *)p->W2T, (void *)p->W1A, (void *)p->W2B,
                  (void *)p->W2A, (void *)p->W1A2, (void *)p->W0A})
    if (d) (void)hipFree(d);
  delete p;
}

extern "C" int snet_radial_mlp_fwd(const snet_mlp_plan *p, const float *emb, int64_t E, float *w_out, void *stream) {
  SNET_REQUIRE(p != nullptr, "snet_radial_mlp_fwd: null plan");
  if (E <= 0) return 0;
  const int64_t grid = (E + 127) / 128;
  SNET_REQUIRE(grid < (1ll << 31), "snet_radial_mlp_fwd: too many edges");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (p->mode == 0)
    radial_mlp_fwd_kernel<<<(unsigned)grid, 256, 0, st>>>(emb, E, p->nb, p->wn, p->W0, p->W1, p->W2, p->act, p->cst,
                                                          w_out);
  else
    radial_mlp_fwd_split_kernel<<<(unsigned)grid, 256, 0, st>>>(emb, E, p->nb, p->wn, p->W0, p->W1A, p->W2B, p->act,
                                                                p->cst, w_out);
  SNET_CHECK_LAUNCH("snet_radial_mlp_fwd");
  return 0;
}

extern "C" int snet_radial_mlp_hidden_fwd_layers(const snet_mlp_plan *const *plans, int32_t n_layers, const float *emb, int64_t E,
                                                 float *const *h2, void *stream) {
  SNET_REQUIRE(plans != nullptr && h2 != nullptr, "snet_radial_mlp_hidden_fwd_layers: null argument");
  SNET_REQUIRE(n_layers >= 1 && n_layers <= MAX_HIDDEN_LAYERS, "snet_radial_mlp_hidden_fwd_layers: 1 .. 8 layers per call");
  HiddenLayers P{};
  P.n = n_layers;
  bool silu = true;
  for (int l = 0; l < n_layers; ++l) {
    const snet_mlp_plan *p = plans[l];
    SNET_REQUIRE(p != nullptr && h2[l] != nullptr, "snet_radial_mlp_hidden_fwd_layers: null plan / output");
    SNET_REQUIRE(p->mode == 1, "snet_radial_mlp_hidden_fwd_layers: split-precision plans only (mode 1)");
    SNET_REQUIRE(p->nb == plans[0]->nb, "snet_radial_mlp_hidden_fwd_layers: the layers must share the edge embedding (same n_basis)");
    P.W0[l] = p->W0; P.W1A[l] = p->W1A; P.h2[l] = h2[l]; P.cst[l] = p->cst; P.act[l] = p->act;
    silu = silu && p->act == 0;
  }
  if (E <= 0) return 0;
  const int64_t grid = (E + 127) / 128;
  SNET_REQUIRE(grid < (1ll << 31), "snet_radial_mlp_hidden_fwd_layers: too many edges");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nb = plans[0]->nb;
  if (silu && nb == 8) radial_mlp_hidden_layers_kernel<0, 8><<<(unsigned)grid, 256, 0, st>>>(emb, E, nb, P);
  else if (silu) radial_mlp_hidden_layers_kernel<0, 0><<<(unsigned)grid, 256, 0, st>>>(emb, E, nb, P);
  else radial_mlp_hidden_layers_kernel<-1, 0><<<(unsigned)grid, 256, 0, st>>>(emb, E, nb, P);
  SNET_CHECK_LAUNCH("snet_radial_mlp_hidden_fwd_layers");
  return 0;
}

extern "C" int snet_radial_mlp_hidden_fwd(const snet_mlp_plan *p, const float *emb, int64_t E, float *h2, void *stream) {
  SNET_REQUIRE(p != nullptr, "snet_radial_mlp_hidden_fwd: null plan");
  return snet_radial_mlp_hidden_fwd_layers(&p, 1, emb, E, &h2, stream);   // (the same kernel: one layer)
}

namespace snet {
int mlp_plan_wn(const snet_mlp_plan *plan) { return plan ? plan->wn : 0; }
const float *mlp_plan_w2_host(const snet_mlp_plan *plan) {
  return (plan && !plan->w2_host.empty()) ? plan->w2_host.data() : nullptr;
}

MlpHidden mlp_plan_hidden(const snet_mlp_plan *plan) {
  MlpHidden h{nullptr, nullptr, 0, 0, 0.f};
  if (plan && plan->mode == 1 && !plan->w0_host.empty())
    h = MlpHidden{plan->w0_host.data(), plan->w1_host.data(), plan->nb, plan->act, plan->cst};
  return h;
}
// fp32 -> fp16 bits, round-to-nearest-even, subnormals kept (what v_cvt_pk_f16_f32 does on the device)
uint16_t f16_rne(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  const uint32_t a = u & 0x7fffffffu;
  if (a >= 0x7f800000u) return sign | (a > 0x7f800000u ? 0x7e00u : 0x7c00u);
  if (a >= 0x477ff000u) return sign | 0x7c00u;                 // rounds to >= 2^16: infinity
  if (a < 0x33000001u) return sign;                             // <= 2^-25: zero
  if (a < 0x38800000u) {                                        // subnormal result: quantum 2^-24
    const int shift = 126 - (int)(a >> 23);                     // 14 .. 24
    const uint32_t m = (a & 0x7fffffu) | 0x800000u;
    const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    return sign | (uint16_t)(q + ((rem > half || (rem == half && (q & 1))) ? 1 : 0));
  }
  const uint32_t r = a + 0xfffu + ((a >> 13) & 1);              // normal: round the 13 dropped bits
  return sign | (uint16_t)((r - 0x38000000u) >> 13);
}
float f16_f(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31, m = h & 0x3ffu;
  float f;
  if (e == 0) {
    f = std::ldexp((float)m, -24);
  } else {
    const uint32_t u = ((e + 112) << 23) | (m << 13);
    std::memcpy(&f, &u, 4);
  }
  return sign ? -f : f;
}
// k with max |v| * 2^k in [2^13, 2^14) (what the kernels' bound_exp / F16_TOP produce for dynamic operands)
int f16_scale_exp(const float *v, size_t n) {
  float m = 0.f;
  for (size_t i = 0; i < n; ++i) m = std::max(m, std::fabs(v[i]));
  if (!(m > 0.f) || !std::isfinite(m)) return 0;
  int e;
  (void)std::frexp(m, &e);  // m = f 2^e, f in [0.5, 1)
  return std::max(-100, std::min(100, 14 - e));
}

int pack_fused_slabs(const float *w2, int wn, int n_sub, const int32_t *sub_cols, int mode, const MlpHidden *tail,
                     void **dev_out, int32_t (&exps)[3]) {
  // mode 1 / 2 / 3: that many bf16 terms per value; mode 4 ("f16x3"): two fp16 terms of the value scaled by a power
  // of two per matrix (exps[0..2] = the exponents applied to W2, W1, W0; the kernels divide them out again)
  const bool f16 = mode == 4;
  const int nt = f16 ? 2 : mode;
  exps[0] = exps[1] = exps[2] = 0;
  if (f16) {
    exps[0] = f16_scale_exp(w2, (size_t)H * wn);
    if (tail) {
      exps[1] = f16_scale_exp(tail->w1, (size_t)H * H);
      exps[2] = f16_scale_exp(tail->w0, (size_t)tail->nb * H);
    }
  }
  const int lps = 8 * nt;  // 1-KB lines per sub-step
  const size_t tail_lines = tail ? (size_t)FUSED_TAIL_FRAGS * nt : 0;
  std::vector<uint16_t> out(((size_t)n_sub * lps + tail_lines) * 64 * 8, 0);
  int cur_exp = exps[0];
  auto put = [&](size_t line, int lane, int slot, float v, int term) {
    uint16_t sp[3];
    if (f16) {
      const float x = std::ldexp(v, cur_exp);
      sp[0] = f16_rne(x);
      sp[1] = f16_rne(x - f16_f(sp[0]));
    } else {
      split3(v, sp);
    }
    out[((line + term) * 64 + lane) * 8 + slot] = sp[term];
  };
  for (int s = 0; s < n_sub; ++s) {
    const size_t base = (size_t)s * lps;
    for (int lane = 0; lane < 64; ++lane) {
      const int i = lane & 15, gg = lane >> 4;
      // w part: tile tp, k-step q: operand row/column i = weight column c0 + i, k slot (gg, t) = hidden 32 q + 8 gg + t
      for (int tp = 0; tp < 2; ++tp) {
        const int c0 = sub_cols[2 * s + tp];
        if (c0 < 0) continue;
        for (int q = 0; q < 2; ++q)
          for (int t = 0; t < 8; ++t)
            for (int term = 0; term < nt; ++term)
              put(base + (size_t)(tp * 2 + q) * nt, lane, t, w2[(size_t)(32 * q + 8 * gg + t) * wn + c0 + i], term);
      }
      // g part: A[i = hidden 16 m + i][k slot (gg, t)] = W2[16 m + i][column of slot], slot t -> tile t >> 2, channel 4 gg + (t & 3)
      for (int m = 0; m < 4; ++m)
        for (int t = 0; t < 8; ++t) {
          const int c0 = sub_cols[2 * s + (t >> 2)];
          if (c0 < 0) continue;
          for (int term = 0; term < nt; ++term)
            put(base + (size_t)4 * nt + (size_t)m * nt, lane, t, w2[(size_t)(16 * m + i) * wn + c0 + 4 * gg + (t & 3)], term);
        }
    }
  }
  if (tail) {
    // Hidden-layer tail of the reverse kernels (nt terms per fragment like the W2 stream).  All four products run in the
    // transposed layout of the g_h2 accumulators -- [unit 16 m + 4 g + r][edge] -- so the k slot (gg, t) of k-step s
    // names hidden unit u(s, gg, t) = 16 (2 s + (t >> 2)) + 4 gg + (t & 3): each lane's own accumulator registers.
    const float *w0 = tail->w0, *w1 = tail->w1;
    const int nb = tail->nb;
    const size_t base = (size_t)n_sub * lps;
    auto unit = [](int s, int gg, int t) { return 16 * (2 * s + (t >> 2)) + 4 * gg + (t & 3); };
    for (int lane = 0; lane < 64; ++lane) {
      const int i = lane & 15, gg = lane >> 4;
      for (int t = 0; t < 8; ++t)
        for (int term = 0; term < nt; ++term) {
          for (int m = 0; m < 4; ++m) {
            // z1^T[16 m + i][edge] = sum_k W0'[k][16 m + i] emb[edge][k], k = 8 gg + t
            const int k = 8 * gg + t;
            cur_exp = exps[2];
            put(base + (size_t)(0 + m) * nt, lane, t, k < nb ? w0[(size_t)k * H + 16 * m + i] : 0.f, term);
            cur_exp = exps[1];
            for (int s = 0; s < 2; ++s) {
              const int u = unit(s, gg, t);
              // z2^T[16 m + i][edge] = sum_u W1'[u][16 m + i] a1[edge][u]
              put(base + (size_t)(4 + 2 * m + s) * nt, lane, t, w1[(size_t)u * H + 16 * m + i], term);
              // g_a1^T[16 m + i][edge] = sum_u W1'[16 m + i][u] g_z2[edge][u]
              put(base + (size_t)(12 + 2 * m + s) * nt, lane, t, w1[(size_t)(16 * m + i) * H + u], term);
            }
          }
          cur_exp = exps[2];
          for (int s = 0; s < 2; ++s)  // g_emb^T[k0 = i][edge] = sum_u W0'[i][u] g_z1[edge][u]
            put(base + (size_t)(20 + s) * nt, lane, t, i < nb ? w0[(size_t)i * H + unit(s, gg, t)] : 0.f, term);
        }
    }
  }
  return upload(out, dev_out);
}
}  // namespace snet

extern "C" int snet_radial_mlp_bwd(const snet_mlp_plan *p, const float *emb, const float *g_w, int64_t E,
                                   float *g_emb, void *stream) {
  SNET_REQUIRE(p != nullptr, "snet_radial_mlp_bwd: null plan");
  if (E <= 0) return 0;
  const int64_t grid = p->mode == 0 ? (E + 127) / 128 : (E + 32 * SNET_MLP_BWD_WAVES - 1) / (32 * SNET_MLP_BWD_WAVES);
  SNET_REQUIRE(grid < (1ll << 31), "snet_radial_mlp_bwd: too many edges");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (p->mode == 0)
    radial_mlp_bwd_kernel<<<(unsigned)grid, 256, 0, st>>>(emb, g_w, E, p->nb, p->wn, p->W0, p->W1, p->W2T, p->act,
                                                          p->cst, g_emb);
  else
    radial_mlp_bwd_split_kernel<<<(unsigned)grid, 64 * SNET_MLP_BWD_WAVES, 0, st>>>(emb, g_w, E, p->nb, p->wn, p->W0, p->W1A, p->W2A,
                                                                p->W1A2, p->W0A, p->act, p->cst, g_emb);
  SNET_CHECK_LAUNCH("snet_radial_mlp_bwd");
  return 0;
}

extern "C" int snet_radial_mlp_hidden_bwd(const snet_mlp_plan *p, const float *emb, const float *g_h2, int64_t E,
                                          float *g_emb, void *stream) {
  SNET_REQUIRE(p != nullptr, "snet_radial_mlp_hidden_bwd: null plan");
  SNET_REQUIRE(p->mode == 1, "snet_radial_mlp_hidden_bwd: split-precision plans only (mode 1)");
  if (E <= 0) return 0;
  const int64_t grid = (E + 127) / 128;
  SNET_REQUIRE(grid < (1ll << 31), "snet_radial_mlp_hidden_bwd: too many edges");
  radial_mlp_hidden_bwd_split_kernel<<<(unsigned)grid, 256, 0, static_cast<hipStream_t>(stream)>>>(
      emb, g_h2, E, p->nb, p->W0, p->W1A, p->W1A2, p->W0A, p->act, p->cst, g_emb);
  SNET_CHECK_LAUNCH("snet_radial_mlp_hidden_bwd");
  return 0;
}
