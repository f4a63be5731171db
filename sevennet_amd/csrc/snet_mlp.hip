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
#include "snet_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int H = 64;  // hidden width of both hidden layers

__device__ __forceinline__ int krow(int s, int half) { return 32 * (s >> 4) + (s & 3) + 8 * ((s & 15) >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, li = lane & 31;
  const int64_t e0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  const bool wave_ok = e0 < E;  // every wave stays in the block-wide barriers
  const int64_t e_lane = e0 + li;
  const bool e_ok = wave_ok && e_lane < E;
  float *tile = gws + wave * GW_TILE;
  const bool vec_ok = (wn & 3) == 0;
  const int srow = lane >> 3, scol = 4 * (lane & 7);

  float4 st_g[4], st_w[2];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // g_w[e0 + 8i + srow][c0 + scol .. +4)
      const int64_t e = e0 + 8 * i + srow;
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (wave_ok && e < E && c0 + scol < wn) {
        const float *p = g_w + e * wn + c0 + scol;
        if (vec_ok && c0 + scol + 3 < wn) {
          q = *reinterpret_cast<const float4 *>(p);
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
      st_w[i] = (c0 + row < wn) ? *reinterpret_cast<const float4 *>(W2T + (int64_t)(c0 + row) * H + col)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
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
      *reinterpret_cast<float4 *>(&w2t[buf][4 * f]) = st_w[i];
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

extern "C" int snet_radial_mlp_fwd(const float *emb, int64_t E, int32_t nb, int32_t h1, int32_t h2, int32_t wn,
                                   const float *W0, const float *W1, const float *W2, int32_t act, float cst,
                                   float *w_out, void *stream) {
  SNET_REQUIRE(h1 == H && h2 == H, "snet_radial_mlp_fwd: fused kernel needs hidden widths [64, 64]");
  SNET_REQUIRE(nb >= 1 && nb <= 32 && wn >= 1, "snet_radial_mlp_fwd: need 1 <= n_basis <= 32, wn >= 1");
  SNET_REQUIRE(act == 0 || act == 1, "snet_radial_mlp_fwd: unknown activation");
  if (E <= 0) return 0;
  const int64_t grid = (E + 127) / 128;
  SNET_REQUIRE(grid < (1ll << 31), "snet_radial_mlp_fwd: too many edges");
  radial_mlp_fwd_kernel<<<(unsigned)grid, 256, 0, static_cast<hipStream_t>(stream)>>>(emb, E, nb, wn, W0, W1, W2, act,
                                                                                       cst, w_out);
  SNET_CHECK_LAUNCH("snet_radial_mlp_fwd");
  return 0;
}

extern "C" int snet_radial_mlp_bwd(const float *emb, const float *g_w, int64_t E, int32_t nb, int32_t h1, int32_t h2,
                                   int32_t wn, const float *W0, const float *W1, const float *W2T, int32_t act,
                                   float cst, float *g_emb, void *stream) {
  SNET_REQUIRE(h1 == H && h2 == H, "snet_radial_mlp_bwd: fused kernel needs hidden widths [64, 64]");
  SNET_REQUIRE(nb >= 1 && nb <= 32 && wn >= 1, "snet_radial_mlp_bwd: need 1 <= n_basis <= 32, wn >= 1");
  SNET_REQUIRE(act == 0 || act == 1, "snet_radial_mlp_bwd: unknown activation");
  if (E <= 0) return 0;
  const int64_t grid = (E + 127) / 128;
  SNET_REQUIRE(grid < (1ll << 31), "snet_radial_mlp_bwd: too many edges");
  radial_mlp_bwd_kernel<<<(unsigned)grid, 256, 0, static_cast<hipStream_t>(stream)>>>(emb, g_w, E, nb, wn, W0, W1, W2T,
                                                                                       act, cst, g_emb);
  SNET_CHECK_LAUNCH("snet_radial_mlp_bwd");
  return 0;
}
