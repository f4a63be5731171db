// Rate of scattering the per-edge source-row gradients straight into g_h[src] with fp32 atomics, against writing g_xe[E,480]
// rows (what conv_bwdf does today, followed by a 5-GB segment sum).  Synthetic graph of the benchmark's size: 97 336 nodes on a
// 46^3 raster, 28 neighbours each inside a +-2 cell box (index distance up to ~4 400 rows), XCD-contiguous node ranges.
//   hipcc --offload-arch=gfx950 -O3 tools/gpu/atomic_probe/atomic_probe.hip -o tools/gpu/atomic_probe/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int DX = 480, DEG = 28, NWV = 4;
using f4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ int xcd_node(unsigned b, unsigned n) {
  const unsigned k = b & 7u, j = b >> 3, q = n >> 3, r = n & 7u;
  return (int)(k * q + (k < r ? k : r) + j);
}

// MODE 0: 16-byte stores of the edge's row to g_xe[e]; 1: fp32 atomics into g_h[src[e]]; 2: the same with unsafe (no-return,
// hardware float add) atomics; 3: plain (racy) read-modify-write, the bandwidth of the same access pattern without atomicity
template <int MODE>
__global__ __launch_bounds__(64 * NWV) void scatter_kernel(const int *__restrict__ src, int n_nodes, float *__restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int node = xcd_node(blockIdx.x, gridDim.x) * NWV + wave;
  if (node >= n_nodes) return;
  for (int k = 0; k < DEG; ++k) {
    const int e = node * DEG + k;
    const int s = src[e];
    if (MODE == 0) {
      float *row = out + (size_t)e * DX;
      for (int c = lane; c < DX / 4; c += 64) {
        const f4 v = {(float)c, 1.f, 2.f, (float)k};
        __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(row) + c);
      }
    } else {
      float *row = out + (size_t)s * DX;
      for (int c = lane; c < DX; c += 64) {
        const float v = 1e-3f * (float)(c + k);
        if (MODE == 1) atomicAdd(row + c, v);
        else if (MODE == 2) unsafeAtomicAdd(row + c, v);
        else row[c] += v;
      }
    }
  }
}

int main() {
  const int G = 46, N = G * G * G, E = N * DEG;
  std::vector<int> src(E);
  unsigned rng = 12345u;
  for (int i = 0; i < N; ++i) {
    const int x = i % G, y = (i / G) % G, z = i / (G * G);
    for (int k = 0; k < DEG; ++k) {
      rng = rng * 1664525u + 1013904223u;
      const int dx = (int)((rng >> 8) % 5) - 2, dy = (int)((rng >> 12) % 5) - 2, dz = (int)((rng >> 16) % 5) - 2;
      src[i * DEG + k] = ((x + dx + G) % G) + G * (((y + dy + G) % G) + G * ((z + dz + G) % G));
    }
  }
  int *d_src;
  float *d_out;
  CK(hipMalloc(&d_src, (size_t)E * 4));
  CK(hipMemcpy(d_src, src.data(), (size_t)E * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_out, (size_t)E * DX * 4));
  CK(hipMemset(d_out, 0, (size_t)E * DX * 4));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  const unsigned grid = (N + NWV - 1) / NWV;
  const char *names[4] = {"16-byte stores to g_xe[E,480]", "atomicAdd(float) into g_h[src]", "unsafeAtomicAdd(float) into g_h[src]",
                          "plain += into g_h[src] (racy)"};
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      CK(hipEventRecord(t0));
      if (mode == 0) scatter_kernel<0><<<grid, 64 * NWV>>>(d_src, N, d_out);
      if (mode == 1) scatter_kernel<1><<<grid, 64 * NWV>>>(d_src, N, d_out);
      if (mode == 2) scatter_kernel<2><<<grid, 64 * NWV>>>(d_src, N, d_out);
      if (mode == 3) scatter_kernel<3><<<grid, 64 * NWV>>>(d_src, N, d_out);
      CK(hipEventRecord(t1));
      CK(hipEventSynchronize(t1));
      float ms;
      CK(hipEventElapsedTime(&ms, t0, t1));
      if (it > 0 && ms < best) best = ms;
    }
    printf("%-40s %8.3f ms   %7.1f GB/s of row data\n", names[mode], best, (double)E * DX * 4 / best / 1e6);
  }
  return 0;
}
