// VALU issue-rate probe: v_fma_f32 vs v_pk_fma_f32 (8 independent accumulator chains per lane, 4 waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_scalar(float *out, float a, float b, int iters) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], a, b);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_packed(float *out, float a, float b, int iters) {
  f32x2 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = f32x2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  const f32x2 a2 = {a, a * 1.0001f}, b2 = {b, b * 0.999f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], a2, b2);
  }
  f32x2 s = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
// three distinct VGPR operands per instruction (the tensor-product bodies' case), scalar broadcast via op_sel
__global__ __launch_bounds__(256) void k_scalar3(float *out, const float *in, int iters) {
  float v[8], u[8], w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = in[threadIdx.x + 64 * i]; u[i] = in[threadIdx.x + 64 * i + 1]; w[i] = in[threadIdx.x + 64 * i + 2]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(u[i], w[(i + 1) & 7], v[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = __builtin_fmaf(v[i], w[(i + 3) & 7], u[i]);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i] + u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_packed3(float *out, const float *in, int iters) {
  f32x2 v[4], u[4], w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = f32x2{in[threadIdx.x + 64 * i], in[threadIdx.x + 64 * i + 3]};
    u[i] = f32x2{in[threadIdx.x + 64 * i + 1], in[threadIdx.x + 64 * i + 4]};
    w[i] = f32x2{in[threadIdx.x + 64 * i + 2], in[threadIdx.x + 64 * i + 5]};
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __builtin_elementwise_fma(u[i], w[(i + 1) & 3], v[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = __builtin_elementwise_fma(v[i], w[(i + 3) & 3], u[i]);
  }
  f32x2 s = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) s += v[i] + u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
// packed with one operand a per-lane scalar broadcast to both halves (op_sel_hi = 0)
__global__ __launch_bounds__(256) void k_packed3b(float *out, const float *in, int iters) {
  f32x2 v[4], u[4];
  float y[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = f32x2{in[threadIdx.x + 64 * i], in[threadIdx.x + 64 * i + 3]};
    u[i] = f32x2{in[threadIdx.x + 64 * i + 1], in[threadIdx.x + 64 * i + 4]};
    y[i] = in[threadIdx.x + 64 * i + 2];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __builtin_elementwise_fma(u[i], f32x2{y[(i + 1) & 3], y[(i + 1) & 3]}, v[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = __builtin_elementwise_fma(v[i], f32x2{y[(i + 3) & 3], y[(i + 3) & 3]}, u[i]);
  }
  f32x2 s = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) s += v[i] + u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
int main() {
  float *out;
  hipMalloc(&out, 256 * 1024 * 4 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, grid = 256 * 4;  // 4 workgroups of 4 waves per CU
  for (int rep = 0; rep < 2; ++rep) {
    float ms;
    hipEventRecord(e0); k_scalar<<<grid, 256>>>(out, 0.999f, 0.001f, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    // 16 fma per iter per lane
    printf("scalar v_fma_f32 : %.3f ms  %.1f TFLOP/s  (%.2f wave-instr/clk/SIMD at 2.4 GHz)\n", ms,
           2.0 * 16 * iters * grid * 256 / ms / 1e9, 16.0 * iters * grid * 4 / (ms * 1e-3 * 2.4e9) / 1024);
    hipEventRecord(e0); k_packed<<<grid, 256>>>(out, 0.999f, 0.001f, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("packed v_pk_fma  : %.3f ms  %.1f TFLOP/s  (%.2f wave-instr/clk/SIMD at 2.4 GHz)\n", ms,
           2.0 * 16 * iters * grid * 256 / ms / 1e9, 8.0 * iters * grid * 4 / (ms * 1e-3 * 2.4e9) / 1024);
  }
  float *in;
  hipMalloc(&in, 4096 * sizeof(float));
  hipMemset(in, 0, 4096 * sizeof(float));
  for (int rep = 0; rep < 2; ++rep) {
    float ms;
    hipEventRecord(e0); k_scalar3<<<grid, 256>>>(out, in, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("scalar 3 VGPR operands : %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * 16 * iters * grid * 256 / ms / 1e9);
    hipEventRecord(e0); k_packed3<<<grid, 256>>>(out, in, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("packed 3 VGPR pairs    : %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * 16 * iters * grid * 256 / ms / 1e9);
    hipEventRecord(e0); k_packed3b<<<grid, 256>>>(out, in, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("packed, splat operand  : %.3f ms  %.1f TFLOP/s\n", ms, 2.0 * 16 * iters * grid * 256 / ms / 1e9);
  }
  return 0;
}
