// Issue-rate probe for gfx950 (round 5): cycles per wave64 instruction measured with s_memtime INSIDE the kernel (clock-independent),
// for 1 / 2 / 4 waves per SIMD, of (a) independent v_fma_f32 chains, (b) v_mfma_f32_16x16x32_f16 chains, and (c) a VALU-only wave
// and a matrix-only wave sharing each SIMD -- do the two pipes overlap ACROSS waves, and what does a wave64 VALU instruction cost?
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/gpu/issue_probe/issue_probe.hip -o tools/gpu/issue_probe/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// mode bit 0: this wave runs VALU, bit 1: this wave runs MFMA; role by wave parity when mode == 3 (even waves VALU, odd waves matrix)
template <int NACC>
__global__ __launch_bounds__(1024) void probe(float *out, unsigned long long *cyc, const float *in, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool do_valu = mode == 1 || (mode == 3 && (wave & 4) == 0);   // waves 0-3 land on SIMDs 0-3, waves 4-7 again: pair (w, w + 4) shares a SIMD
  const bool do_mfma = mode == 2 || (mode == 3 && (wave & 4) != 0);
  float v[NACC], u[NACC], w[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) { v[i] = in[threadIdx.x + 64 * i]; u[i] = in[threadIdx.x + 64 * i + 1]; w[i] = in[threadIdx.x + 64 * i + 2]; }
  f32x4 acc[4];
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[threadIdx.x + i]; b[i] = (_Float16)in[threadIdx.x + 8 + i]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = now();
  if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) v[i] = __builtin_fmaf(u[i], w[(i + 1) % NACC], v[i]);
#pragma unroll
      for (int i = 0; i < NACC; ++i) u[i] = __builtin_fmaf(v[i], w[(i + 3) % NACC], u[i]);
    }
  }
  if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < NACC / 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
  }
  if (mode == 4) {   // packed: v_pk_fma_f32, three VGPR-pair operands
    f32x2 pv[NACC / 2], pu[NACC / 2], pw[NACC / 2];
#pragma unroll
    for (int i = 0; i < NACC / 2; ++i) { pv[i] = f32x2{v[2 * i], v[2 * i + 1]}; pu[i] = f32x2{u[2 * i], u[2 * i + 1]}; pw[i] = f32x2{w[2 * i], w[2 * i + 1]}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int i = 0; i < NACC / 2; ++i) pv[i] = __builtin_elementwise_fma(pu[i], pw[(i + 1) % (NACC / 2)], pv[i]);
#pragma unroll
        for (int i = 0; i < NACC / 2; ++i) pu[i] = __builtin_elementwise_fma(pv[i], pw[(i + 3) % (NACC / 2)], pu[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < NACC / 2; ++i) { v[2 * i] = pv[i][0]; v[2 * i + 1] = pv[i][1]; u[2 * i] = pu[i][0]; u[2 * i + 1] = pu[i][1]; }
  }
  if (mode == 5) {   // VALU interleaved 1:1 with scalar ALU work of the same wave: does a SALU instruction take a VALU issue slot?
    int sacc = iters;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        v[i] = __builtin_fmaf(u[i], w[(i + 1) % NACC], v[i]);
        asm volatile("s_mul_i32 %0, %0, 3" : "+s"(sacc));
      }
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        u[i] = __builtin_fmaf(v[i], w[(i + 3) % NACC], u[i]);
        asm volatile("s_add_u32 %0, %0, 7" : "+s"(sacc));
      }
    }
    v[0] += (float)sacc;
  }
  if (mode == 6) {   // VALU interleaved 1:1 with LDS reads of the same wave (results unused until the end)
    __shared__ float lds[4096];
    lds[threadIdx.x] = 0.f;
    float acc_l = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        v[i] = __builtin_fmaf(u[i], w[(i + 1) % NACC], v[i]);
        float t; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t) : "v"((threadIdx.x & 63) * 4), "n"(256 * (i & 7)));
        acc_l += 0.f * 0.f;
        if (it == iters) acc_l += t;
      }
#pragma unroll
      for (int i = 0; i < NACC; ++i) u[i] = __builtin_fmaf(v[i], w[(i + 3) % NACC], u[i]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    v[0] += acc_l;
  }
  const unsigned long long t1 = now();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += v[i] + u[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

int main() {
  float *out, *in;
  unsigned long long *cyc;
  const int grid = 256;   // one workgroup per CU
  hipMalloc(&out, grid * 1024 * sizeof(float));
  hipMalloc(&in, 8192 * sizeof(float));
  hipMemset(in, 0, 8192 * sizeof(float));
  hipMalloc(&cyc, grid * 16 * sizeof(unsigned long long));
  std::vector<unsigned long long> h(grid * 16);
  constexpr int NACC = 8;
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[7] = {"", "VALU only (v_fma_f32, 3 VGPR operands, 8 independent chains)", "matrix only (v_mfma_f32_16x16x32_f16, 4 independent accumulators)",
                          "waves 0-3 VALU + waves 4-7 matrix (one of each per SIMD)", "packed VALU (v_pk_fma_f32, 3 VGPR pairs): wave-instructions of 2 FMAs per lane",
                          "VALU + one SALU instruction after each (same wave)", "VALU, first half each followed by a ds_read_b32 (same wave)"};
  for (int mode = 1; mode <= 6; ++mode)
    for (int waves : {4, 8, 16}) {
      if (mode == 3 && waves != 8) continue;
      for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, grid * 16 * sizeof(unsigned long long));
        float ms;
        hipEventRecord(e0);
        probe<NACC><<<grid, 64 * waves>>>(out, cyc, in, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sv = 0, sm = 0; int nv = 0, nm = 0;
        for (int b = 0; b < grid; ++b)
          for (int w = 0; w < waves; ++w) {
            const bool is_m = mode == 2 || (mode == 3 && (w & 4));
            (is_m ? sm : sv) += (double)h[b * 16 + w];
            (is_m ? nm : nv)++;
          }
        if (rep == 0) continue;
        const double n_valu = 2.0 * NACC * iters, n_mfma = 4.0 * (NACC / 2) * iters;
        printf("%-78s %2d waves/CU (%d per SIMD): %7.3f ms", names[mode], waves, waves / 4, ms);
        if (mode >= 4) { const double n = (mode == 4 ? 2.0 * NACC : 2.0 * NACC) * iters; printf("  %.2f cycles per VALU wave-instruction (%.2f per SIMD)  [effective clock %.2f GHz]\n", sv / nv / n, sv / nv / n / (waves / 4), sv / nv / (ms * 1e-3) / 1e9); continue; }
        if (nv) printf("  VALU %.2f cycles per wave-instruction (%.2f per SIMD)", sv / nv / n_valu, sv / nv / n_valu / (mode == 3 ? 1 : waves / 4));
        if (nm) printf("  MFMA %.2f cycles per wave-instruction (%.2f per SIMD)", sm / nm / n_mfma, sm / nm / n_mfma / (mode == 3 ? 1 : waves / 4));
        printf("  [effective clock %.2f GHz]\n", (nv ? sv / nv : sm / nm) / (ms * 1e-3) / 1e9);
      }
    }
  return 0;
}
