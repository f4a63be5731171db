// Dense channel mixing on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Serves every dense contraction of the SevenNet step:
// per-irrep blocks of o3.Linear, per-species FCTP slices and the radial MLP layers.
// Rows are addressed through (node, m) so that ir_mul feature slabs are read in place.
//
// Tiling: 256 threads = 4 wavefronts; block tile BM=128 rows x BN columns (BN = 32/64/128
// chosen by N), K staged through LDS in slabs of BK=32.  A is stored k-major in LDS
// (As[k][row], row stride 129 -> conflict-free ds_write_b32 and ds_read_b32), so the
// MFMA A fragment (lane l: row l&31, k = 2*kk + (l>>5)) is one ds_read_b32 per step.
// Weights packed by snet_gemm_split_pack take the split-precision kernel further down instead.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <type_traits>

#include "snet_common.h"
#include "snet_split.h"

#ifndef SNET_GEMM_OCC
#define SNET_GEMM_OCC 4
#endif
namespace {

using snet::f32x16;

// Row r of a GEMM is component m of node n (r = n d + m).  A 64-bit division per row costs ~100 instructions, and the
// epilogue needs 16 of them per lane: the tile's first row is divided once (wave-uniform), every other row of the tile
// is an offset < 256 from it, divided by a 16-bit reciprocal (exact for (d + 256) d < 65536, i.e. d <= 150).
struct RowMap {
  int64_t n0;
  uint32_t m0, d, inv;
  __device__ __forceinline__ RowMap(int64_t row0, int d_) : d((uint32_t)d_), inv(65536u / (uint32_t)d_ + 1u) {
    n0 = row0 / d_;
    m0 = (uint32_t)(row0 - n0 * d_);
  }
  __device__ __forceinline__ void at(int off, int64_t &n, int &m) const {  // row0 + off, 0 <= off < 256
    const uint32_t x = m0 + (uint32_t)off, q = (x * inv) >> 16;
    n = n0 + q;
    m = (int)(x - q * d);
  }
};

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int AS_STRIDE = BM + 1;

template <int NT>  // number of 32-wide column tiles per block
__device__ __forceinline__ void gemm_body(float *As, float *Bs, int bx, int by,
    const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int64_t n_rows, int d, int K,
    int N, int64_t a_node_stride, int64_t a_off, int64_t c_node_stride, int64_t c_off,
    const int32_t *__restrict__ row_idx, int accumulate) {
  constexpr int BN = 32 * NT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)bx * BM;
  const int col0 = by * BN;
  const RowMap rows(row0, d);

  // each thread stages 4 A rows: r = tid/8 + 32*i, k-quad = tid%8
  const int lr = tid >> 3;
  const int kq = tid & 7;
  const float *a_ptr[4];
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = row0 + lr + 32 * i;
    a_ok[i] = r < n_rows;
    int64_t n = 0;
    int m = 0;
    if (a_ok[i]) rows.at(lr + 32 * i, n, m);
    const int64_t node = row_idx ? (int64_t)row_idx[n] : n;
    a_ptr[i] = A + node * a_node_stride + a_off + (int64_t)m * K;
  }
  const bool a_vec = ((K & 3) == 0) && ((a_off & 3) == 0) && ((a_node_stride & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(A) & 15) == 0);

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.0f;

  for (int k0 = 0; k0 < K; k0 += BK) {
    // ---- stage A (BM x BK) k-major
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + 4 * kq;
      float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
      if (a_ok[i]) {
        if (a_vec && k + 3 < K) {
          const float4 t = *reinterpret_cast<const float4 *>(a_ptr[i] + k);
          v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w;
        } else {
          if (k + 0 < K) v0 = a_ptr[i][k + 0];
          if (k + 1 < K) v1 = a_ptr[i][k + 1];
          if (k + 2 < K) v2 = a_ptr[i][k + 2];
          if (k + 3 < K) v3 = a_ptr[i][k + 3];
        }
      }
      const int r = lr + 32 * i;
      As[(4 * kq + 0) * AS_STRIDE + r] = v0;
      As[(4 * kq + 1) * AS_STRIDE + r] = v1;
      As[(4 * kq + 2) * AS_STRIDE + r] = v2;
      As[(4 * kq + 3) * AS_STRIDE + r] = v3;
    }
    // ---- stage B (BK x BN)
    for (int idx = tid; idx < BK * BN; idx += 256) {
      const int kk = idx / BN;
      const int nn = idx - kk * BN;
      const int k = k0 + kk;
      const int n = col0 + nn;
      Bs[idx] = (k < K && n < N) ? B[(int64_t)k * N + n] : 0.0f;
    }
    __syncthreads();
    // ---- MFMA: wave owns rows [32*wave, 32*wave+32)
    const float *as = As + 32 * wave + (lane & 31);
    const float *bs = Bs + (lane & 31);
    const int kh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float a = as[(2 * kk + kh) * AS_STRIDE];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float b = bs[(2 * kk + kh) * BN + 32 * t];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: D[row = (j&3) + 8*(j>>2) + 4*(lane>>5)][col = lane&31]
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int rl = 32 * wave + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
    const int64_t r = row0 + rl;
    if (r >= n_rows) continue;
    int64_t n;
    int m;
    rows.at(rl, n, m);
    const int64_t node = row_idx ? (int64_t)row_idx[n] : n;
    float *crow = C + node * c_node_stride + c_off + (int64_t)m * N;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = col0 + 32 * t + (lane & 31);
      if (col < N) {
        const float v = acc[t][j];
        crow[col] = accumulate ? crow[col] + v : v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Split-precision variant (bf16 x 6 on v_mfma_f32_32x32x16_bf16, snet_split.h): same contraction,
// weights pre-packed on the host by snet_gemm_split_pack into B fragments
//     packed[((tile*nq + q)*3 + term)*64 + lane] = 8 bf16 : B[k = 16q + 8(lane>>5) + i][32 tile + (lane&31)]
template <int NT, int MT>
__device__ __forceinline__ void gemm_split_store(const snet::f32x16 (&acc)[MT][NT], const RowMap &rows, int64_t row0, int half, int li,
    int tile0, float *__restrict__ C, int64_t n_rows, int N, int64_t c_node_stride, int64_t c_off,
    const int32_t *__restrict__ row_idx, int accumulate) {
  // Fast path (round 6; wave-uniform condition): scalar blocks (d = 1: row = node), no row list, the wave's 32 MT rows and NT column
  // tiles all in range -- the l = 0 blocks of every linear away from the last tile, i.e. most of the node-GEMM work.  A row's address is
  // then a per-lane base plus a wave-uniform multiple of the node stride: no row map, no gather, no predicate per store (the general
  // path below spends ~12 vector instructions and a branch per row on them; the epilogue is a third of the stream of a K = 128 tile).
  if (rows.d == 1u && row_idx == nullptr && row0 + 32 * MT <= n_rows && 32 * (tile0 + NT) <= N) {
    float *base = C + (row0 + 4 * half) * c_node_stride + c_off + 32 * tile0 + li;
    if (accumulate) {
      // read-modify-write in chunks of CH rows: ALL loads of a chunk are issued before its first store.  Written row by row
      // (`crow[..] += acc`), every row's store had to wait for its own load and the next row's load came after that store --
      // C may alias C -- i.e. 16 memory latencies in series per 32-row tile.
      constexpr int CH = NT >= 4 ? 8 : 16;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j0 = 0; j0 < 16; j0 += CH) {
          float old[CH][NT];
#pragma unroll
          for (int jj = 0; jj < CH; ++jj) {
            const int j = j0 + jj;
            const float *crow = base + (int64_t)(32 * mt + (j & 3) + 8 * (j >> 2)) * c_node_stride;
#pragma unroll
            for (int t = 0; t < NT; ++t) old[jj][t] = crow[32 * t];
          }
#pragma unroll
          for (int jj = 0; jj < CH; ++jj) {
            const int j = j0 + jj;
            float *crow = base + (int64_t)(32 * mt + (j & 3) + 8 * (j >> 2)) * c_node_stride;
#pragma unroll
            for (int t = 0; t < NT; ++t) crow[32 * t] = old[jj][t] + acc[mt][t][j];
          }
        }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          float *crow = base + (int64_t)(32 * mt + (j & 3) + 8 * (j >> 2)) * c_node_stride;
#pragma unroll
          for (int t = 0; t < NT; ++t) crow[32 * t] = acc[mt][t][j];
        }
    }
    return;
  }
  auto row_of = [&](int mt, int j) -> float * {   // the lane's C row of accumulator entry (mt, j), nullptr past the end
    const int off = 32 * mt + (j & 3) + 8 * (j >> 2) + 4 * half;
    if (row0 + off >= n_rows) return nullptr;
    int64_t n;
    int m;
    rows.at(off, n, m);
    const int64_t node = row_idx ? (int64_t)row_idx[n] : n;
    return C + node * c_node_stride + c_off + (int64_t)m * N;
  };
  if (accumulate) {   // chunks of 4 rows, loads before stores (see the fast path)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j0 = 0; j0 < 16; j0 += 4) {
        float *crow[4];
        float old[4][NT];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          crow[jj] = row_of(mt, j0 + jj);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int col = 32 * (tile0 + t) + li;
            old[jj][t] = (crow[jj] != nullptr && col < N) ? crow[jj][col] : 0.f;
          }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int col = 32 * (tile0 + t) + li;
            if (crow[jj] != nullptr && col < N) crow[jj][col] = old[jj][t] + acc[mt][t][j0 + jj];
          }
      }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float *crow = row_of(mt, j);
      if (crow == nullptr) continue;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int col = 32 * (tile0 + t) + li;
        if (col < N) crow[col] = acc[mt][t][j];
      }
    }
}

// A wave owns MT*32 rows and all NT column tiles of its block: the A fragment (8 consecutive k of
// one row = 32 B per lane) comes straight from global memory into registers and is split there --
// A is not shared between waves, so it never visits LDS.  B fragments are shared by the 4 waves:
// one NT*3-KB slab per 16-wide k step, double buffered in LDS (each ds_read_b128 feeds MT MFMAs).
template <int NT, int MT>
__device__ __forceinline__ void gemm_split_body(snet::u32x4 *Bs, int bx, int by, const float *__restrict__ A,
    const snet::u32x4 *__restrict__ Bp, float *__restrict__ C, int64_t n_rows, int d, int K, int N,
    int64_t a_node_stride, int64_t a_off, int64_t c_node_stride, int64_t c_off, const int32_t *__restrict__ row_idx,
    int accumulate) {
  using namespace snet;
  constexpr int SLAB = NT * 192;  // u32x4 per k step
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, li = lane & 31;
  const int64_t row0 = ((int64_t)bx * 4 + wave) * (32 * MT);
  const int nq = (K + 15) >> 4, n_tiles = (N + 31) >> 5, tile0 = by * NT;
  const RowMap rows(row0, d);

  const float *a_ptr[MT];
  bool a_ok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int64_t r = row0 + 32 * mt + li;
    a_ok[mt] = r < n_rows;
    int64_t n = 0;
    int m = 0;
    if (a_ok[mt]) rows.at(32 * mt + li, n, m);
    const int64_t node = row_idx ? (int64_t)row_idx[n] : n;
    a_ptr[mt] = A + node * a_node_stride + a_off + (int64_t)m * K + 8 * half;   // (rows past the end: node 0, m 0 -- valid memory, results never stored)
  }
  const bool a_vec = ((K & 3) == 0) && ((a_off & 3) == 0) && ((a_node_stride & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  auto load_a = [&](int q, float (&v)[MT][8]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int k = 16 * q + 8 * half;
      if (a_ok[mt] && a_vec && k + 7 < K) {
        // (streaming loads here were measured: node linears 2.18 -> 3.20 ms per step -- a 128-byte line of a row serves two k steps)
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(a_ptr[mt] + 16 * q);
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(a_ptr[mt] + 16 * q + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[mt][i] = lo[i]; v[mt][4 + i] = hi[i]; }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[mt][i] = (a_ok[mt] && k + i < K) ? a_ptr[mt][16 * q + i] : 0.f;
      }
    }
  };
  constexpr int NST = (SLAB + 255) / 256;
  auto load_b = [&](int q, u32x4 (&st)[NST]) {
    // Fast path (round 6), wave-uniform condition: all NT column tiles exist -- no per-thread tile test.  Same box, three interleaved
    // runs: node_linear_bwd 2.53 -> 2.39 ms, node_linear_fwd 2.19 -> 2.20.  The matching fast path for the A operand (two
    // unconditional 16-byte loads when the k step lies inside K) was measured too and is NOT taken: fwd 2.19 -> 2.25, and with both
    // the backward gain shrinks to 2.48 (profiles/r06_ab_node_kernels.txt).
    if (tile0 + NT <= n_tiles) {
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        const int idx = tid + 256 * i;
        const int t = idx / 192, rem = idx - 192 * t;
        if (SLAB % 256 == 0 || idx < SLAB) st[i] = Bp[((int64_t)(tile0 + t) * nq + q) * 192 + rem];
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int idx = tid + 256 * i;
      const int t = idx / 192, rem = idx - 192 * t;
      u32x4 z = {0u, 0u, 0u, 0u};
      if (idx < SLAB && tile0 + t < n_tiles) z = Bp[((int64_t)(tile0 + t) * nq + q) * 192 + rem];
      st[i] = z;
    }
  };
  auto store_b = [&](int buf, const u32x4 (&st)[NST]) {
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int idx = tid + 256 * i;
      if (idx < SLAB) Bs[buf * SLAB + idx] = st[i];
    }
  };

  // (accumulating launches add C in the epilogue, ONE rounding per entry.  Starting the accumulators from C instead hides the read
  // -- node_linear_fwd 2.51 -> 2.33 ms -- but every matrix instruction then rounds at the magnitude of the row already there:
  // measured, the energy error of the 10 648-atom cell went from 0.6e-4 to 1.1e-4 eV/atom, outside the fp32 class; not taken)
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[mt][t] = zero16();

  float av[MT][8], an[MT][8];
  u32x4 st[NST];
  load_a(0, av);
  load_b(0, st);
  store_b(0, st);
  __syncthreads();
  int buf = 0;
  for (int q = 0; q < nq; ++q) {
    const bool more = q + 1 < nq;
    if (more) {
      load_a(q + 1, an);
      load_b(q + 1, st);
    }
    Split3 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = split8(av[mt]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      bf16x8 b[3];
#pragma unroll
      for (int term = 0; term < 3; ++term) b[term] = as_bf16x8(Bs[buf * SLAB + (t * 3 + term) * 64 + lane]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][t] = mfma6(a[mt], b, acc[mt][t]);
    }
    if (more) {
      store_b(buf ^ 1, st);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 8; ++i) av[mt][i] = an[mt][i];
    }
    __syncthreads();
    buf ^= 1;
  }

  gemm_split_store<NT, MT>(acc, rows, row0, half, li, tile0, C, n_rows, N, c_node_stride, c_off, row_idx, accumulate);
}


template <int NT>
__global__ __launch_bounds__(256) void gemm_kernel(
    const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int64_t n_rows, int d, int K,
    int N, int64_t a_node_stride, int64_t a_off, int64_t c_node_stride, int64_t c_off,
    const int32_t *__restrict__ row_idx, int accumulate) {
  __shared__ float As[BK * AS_STRIDE];
  __shared__ float Bs[BK * 32 * NT];
  gemm_body<NT>(As, Bs, blockIdx.x, blockIdx.y, A, B, C, n_rows, d, K, N, a_node_stride, a_off, c_node_stride, c_off,
                row_idx, accumulate);
}

// Grouped launch: all per-irrep GEMMs of one equivariant linear in ONE grid (they share A, C, the node
// strides and the row list).  At MD sizes of ~10^4 atoms per GPU the separate launches (3 per linear,
// ~90 per step, each only a few hundred workgroups) were launch/tail bound: 24 % of the step.
struct GroupArgs {
  int n;
  int first_block[SNET_MAX_GEMM_GROUP + 1];  // prefix of workgroups per problem
  int nbx[SNET_MAX_GEMM_GROUP];
  snet_gemm_desc p[SNET_MAX_GEMM_GROUP];
};

__global__ __launch_bounds__(256) void gemm_grouped_kernel(GroupArgs G, const float *__restrict__ A,
                                                           float *__restrict__ C, int64_t n_nodes,
                                                           int64_t a_node_stride, int64_t c_node_stride,
                                                           const int32_t *__restrict__ row_idx) {
  __shared__ float As[BK * AS_STRIDE];
  __shared__ float Bs[BK * 128];
  int q = 0;
  while (q + 1 < G.n && (int)blockIdx.x >= G.first_block[q + 1]) ++q;
  const snet_gemm_desc P = G.p[q];
  const int b = blockIdx.x - G.first_block[q];
  const int bx = b % G.nbx[q], by = b / G.nbx[q];
  const int64_t n_rows = n_nodes * P.d;
  if (P.N > 64)
    gemm_body<4>(As, Bs, bx, by, A, P.B, C, n_rows, P.d, P.K, P.N, a_node_stride, P.a_off, c_node_stride, P.c_off,
                 row_idx, P.accumulate);
  else if (P.N > 32)
    gemm_body<2>(As, Bs, bx, by, A, P.B, C, n_rows, P.d, P.K, P.N, a_node_stride, P.a_off, c_node_stride, P.c_off,
                 row_idx, P.accumulate);
  else
    gemm_body<1>(As, Bs, bx, by, A, P.B, C, n_rows, P.d, P.K, P.N, a_node_stride, P.a_off, c_node_stride, P.c_off,
                 row_idx, P.accumulate);
}

template <int MT>
__global__ __launch_bounds__(256, MT == 1 ? SNET_GEMM_OCC : 2) void gemm_split_grouped_kernel(GroupArgs G, const float *__restrict__ A,
                                                                    float *__restrict__ C, int64_t n_nodes,
                                                                    int64_t a_node_stride, int64_t c_node_stride,
                                                                    const int32_t *__restrict__ row_idx) {
  __shared__ snet::u32x4 Bs[2 * 4 * 192];
  int q = 0;
  while (q + 1 < G.n && (int)blockIdx.x >= G.first_block[q + 1]) ++q;
  const snet_gemm_desc P = G.p[q];
  const int b = blockIdx.x - G.first_block[q];
  const int bx = b % G.nbx[q], by = b / G.nbx[q];
  const int64_t n_rows = n_nodes * P.d;
  const snet::u32x4 *Bp = static_cast<const snet::u32x4 *>(P.B_split);
  if (P.N > 64)
    gemm_split_body<4, MT>(Bs, bx, by, A, Bp, C, n_rows, P.d, P.K, P.N, a_node_stride, P.a_off, c_node_stride, P.c_off,
                           row_idx, P.accumulate);
  else if (P.N > 32)
    gemm_split_body<2, MT>(Bs, bx, by, A, Bp, C, n_rows, P.d, P.K, P.N, a_node_stride, P.a_off, c_node_stride, P.c_off,
                           row_idx, P.accumulate);
  else
    gemm_split_body<1, MT>(Bs, bx, by, A, Bp, C, n_rows, P.d, P.K, P.N, a_node_stride, P.a_off, c_node_stride, P.c_off,
                           row_idx, P.accumulate);
}

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

}  // namespace

extern "C" int64_t snet_gemm_split_size(int32_t K, int32_t N) {
  if (K < 1 || N < 1) return 0;
  return (int64_t)((N + 31) / 32) * ((K + 15) / 16) * 3 * 64 * 16;
}

extern "C" int snet_gemm_split_pack(const float *B_host, int32_t K, int32_t N, void *packed_host) {
  SNET_REQUIRE(B_host != nullptr && packed_host != nullptr && K >= 1 && N >= 1, "snet_gemm_split_pack: bad argument");
  uint16_t *out = static_cast<uint16_t *>(packed_host);
  const int nq = (K + 15) / 16, n_tiles = (N + 31) / 32;
  for (int t = 0; t < n_tiles; ++t)
    for (int q = 0; q < nq; ++q)
      for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 8; ++i) {
          const int k = 16 * q + 8 * (lane >> 5) + i, n = 32 * t + (lane & 31);
          const float x = (k < K && n < N) ? B_host[(size_t)k * N + n] : 0.f;
          const uint16_t h = bf16_rne(x);
          const float r1 = x - bf16_f(h);
          const uint16_t m = bf16_rne(r1);
          const uint16_t l = bf16_rne(r1 - bf16_f(m));
          const size_t base = ((size_t)(t * nq + q) * 3) * 64;
          out[((base + 0 * 64 + lane) * 8) + i] = h;
          out[((base + 1 * 64 + lane) * 8) + i] = m;
          out[((base + 2 * 64 + lane) * 8) + i] = l;
        }
  return 0;
}

extern "C" int snet_gemm_grouped(const snet_gemm_desc *descs_host, int32_t n_desc, const float *A, float *C,
                                 int64_t n_nodes, int64_t a_node_stride, int64_t c_node_stride,
                                 const int32_t *row_idx, void *stream) {
  SNET_REQUIRE(descs_host != nullptr && n_desc >= 1 && n_desc <= SNET_MAX_GEMM_GROUP,
               "snet_gemm_grouped: 1..8 problems required");
  if (n_nodes <= 0) return 0;
  GroupArgs G;
  G.n = n_desc;
  int n_split = 0;
  int64_t rows_max = 0;  (void)rows_max;
  for (int i = 0; i < n_desc; ++i) {
    n_split += descs_host[i].B_split != nullptr;
    rows_max = rows_max > n_nodes * descs_host[i].d ? rows_max : n_nodes * descs_host[i].d;
  }
  SNET_REQUIRE(n_split == 0 || n_split == n_desc, "snet_gemm_grouped: mix of split-packed and fp32 weights in one group");
  const bool split = n_split > 0;
  static const int mt_env = getenv("SNET_GEMM_MT") ? atoi(getenv("SNET_GEMM_MT")) : 0;  // tuning knob
  // 32 rows per wave (MT = 1, 3 waves/SIMD) beats 64 (MT = 2, 2 waves/SIMD) at every size measured: the
  // node linears are HBM-bound and want bytes in flight more than B-fragment reuse
  const int mt = mt_env ? mt_env : 1;
  const int bm = split ? 128 * mt : BM;
  int64_t total = 0;
  for (int i = 0; i < n_desc; ++i) {
    const snet_gemm_desc &p = descs_host[i];
    SNET_REQUIRE(p.d >= 1 && p.d <= 150 && p.K >= 1 && p.N >= 1 && (p.B != nullptr || p.B_split != nullptr),
                 "snet_gemm_grouped: bad problem");
    const int bn = p.N > 64 ? 128 : (p.N > 32 ? 64 : 32);
    const int64_t nbx = (n_nodes * p.d + bm - 1) / bm;
    const int64_t nby = (p.N + bn - 1) / bn;
    SNET_REQUIRE(nbx < (1ll << 30), "snet_gemm_grouped: too many rows");
    G.p[i] = p;
    G.nbx[i] = (int)nbx;
    G.first_block[i] = (int)total;
    total += nbx * nby;
  }
  G.first_block[n_desc] = (int)total;
  SNET_REQUIRE(total < (1ll << 31), "snet_gemm_grouped: grid too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!split)
    gemm_grouped_kernel<<<(unsigned)total, 256, 0, st>>>(G, A, C, n_nodes, a_node_stride, c_node_stride, row_idx);
  else if (mt == 2)
    gemm_split_grouped_kernel<2><<<(unsigned)total, 256, 0, st>>>(G, A, C, n_nodes, a_node_stride, c_node_stride, row_idx);
  else
    gemm_split_grouped_kernel<1><<<(unsigned)total, 256, 0, st>>>(G, A, C, n_nodes, a_node_stride, c_node_stride, row_idx);
  SNET_CHECK_LAUNCH("snet_gemm_grouped");
  return 0;
}

extern "C" int snet_gemm(const float *A, const float *B, float *C, int64_t n_nodes, int32_t d, int32_t K,
                         int32_t N, int64_t a_node_stride, int64_t a_off, int64_t c_node_stride, int64_t c_off,
                         const int32_t *row_idx, int32_t accumulate, void *stream) {
  SNET_REQUIRE(d >= 1 && d <= 150 && K >= 1 && N >= 1, "snet_gemm: bad shape (1 <= d <= 150)");
  if (n_nodes <= 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t n_rows = n_nodes * d;
  const int64_t gx = (n_rows + BM - 1) / BM;
  SNET_REQUIRE(gx < (1ll << 31), "snet_gemm: too many rows");
  if (N > 64) {
    dim3 grid((unsigned)gx, (unsigned)((N + 127) / 128));
    gemm_kernel<4><<<grid, 256, 0, st>>>(A, B, C, n_rows, d, K, N, a_node_stride, a_off, c_node_stride, c_off,
                                         row_idx, accumulate);
  } else if (N > 32) {
    dim3 grid((unsigned)gx, 1);
    gemm_kernel<2><<<grid, 256, 0, st>>>(A, B, C, n_rows, d, K, N, a_node_stride, a_off, c_node_stride, c_off,
                                         row_idx, accumulate);
  } else {
    dim3 grid((unsigned)gx, 1);
    gemm_kernel<1><<<grid, 256, 0, st>>>(A, B, C, n_rows, d, K, N, a_node_stride, a_off, c_node_stride, c_off,
                                         row_idx, accumulate);
  }
  SNET_CHECK_LAUNCH("snet_gemm");
  return 0;
}
