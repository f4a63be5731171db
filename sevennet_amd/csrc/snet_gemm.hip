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
#include "snet_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

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
  const int wave = tid >> 6;
  const int64_t row0 = (int64_t)bx * BM;
  const int col0 = by * BN;

  // each thread stages 4 A rows: r = tid/8 + 32*i, k-quad = tid%8
  const int lr = tid >> 3;
  const int kq = tid & 7;
  const float *a_ptr[4];
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = row0 + lr + 32 * i;
    a_ok[i] = r < n_rows;
    const int64_t rr = a_ok[i] ? r : 0;
    const int64_t n = rr / d;
    const int m = (int)(rr - n * d);
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
    const int64_t n = r / d;
    const int m = (int)(r - n * d);
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

}  // namespace

extern "C" int snet_gemm_grouped(const snet_gemm_desc *descs_host, int32_t n_desc, const float *A, float *C,
                                 int64_t n_nodes, int64_t a_node_stride, int64_t c_node_stride,
                                 const int32_t *row_idx, void *stream) {
  SNET_REQUIRE(descs_host != nullptr && n_desc >= 1 && n_desc <= SNET_MAX_GEMM_GROUP,
               "snet_gemm_grouped: 1..8 problems required");
  if (n_nodes <= 0) return 0;
  GroupArgs G;
  G.n = n_desc;
  int64_t total = 0;
  for (int i = 0; i < n_desc; ++i) {
    const snet_gemm_desc &p = descs_host[i];
    SNET_REQUIRE(p.d >= 1 && p.K >= 1 && p.N >= 1 && p.B != nullptr, "snet_gemm_grouped: bad problem");
    const int bn = p.N > 64 ? 128 : (p.N > 32 ? 64 : 32);
    const int64_t nbx = (n_nodes * p.d + BM - 1) / BM;
    const int64_t nby = (p.N + bn - 1) / bn;
    SNET_REQUIRE(nbx < (1ll << 30), "snet_gemm_grouped: too many rows");
    G.p[i] = p;
    G.nbx[i] = (int)nbx;
    G.first_block[i] = (int)total;
    total += nbx * nby;
  }
  G.first_block[n_desc] = (int)total;
  SNET_REQUIRE(total < (1ll << 31), "snet_gemm_grouped: grid too large");
  gemm_grouped_kernel<<<(unsigned)total, 256, 0, static_cast<hipStream_t>(stream)>>>(G, A, C, n_nodes, a_node_stride,
                                                                                     c_node_stride, row_idx);
  SNET_CHECK_LAUNCH("snet_gemm_grouped");
  return 0;
}

extern "C" int snet_gemm(const float *A, const float *B, float *C, int64_t n_nodes, int32_t d, int32_t K,
                         int32_t N, int64_t a_node_stride, int64_t a_off, int64_t c_node_stride, int64_t c_off,
                         const int32_t *row_idx, int32_t accumulate, void *stream) {
  SNET_REQUIRE(d >= 1 && K >= 1 && N >= 1, "snet_gemm: bad shape");
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
