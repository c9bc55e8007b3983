// tools/mfma_rate.hip -- what the FP4 matrix-core instruction of the scan sustains on the whole
// chip: back-to-back v_mfma_scale_f32_32x32x64_f8f6f4 (FP4 operands; the block-scaled form of rounds 1-2)
// and v_mfma_f32_32x32x64_f8f6f4 (no block scales: what the scan uses now), W waves per SIMD, on random
// codes (the scan's data) and on zeros.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_rate.hip -o tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// SCALED: the block-scaled form the scan used in rounds 1-2 (products +-1 x 2^-7); else the same instruction without block
// scales (the compiler selects v_mfma_f32_32x32x64_f8f6f4 when both scale operands are 0) on codes +-0.5.
template <int CHAINS, bool SCALED = true>
__global__ __launch_bounds__(64) void rate_kernel(const unsigned* __restrict__ src, float* __restrict__ out, int iters,
                                                  unsigned long long* __restrict__ cycles) {
  i32x8 a[4], b[4];
  for (int m = 0; m < 4; m++) {
    a[m] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
    b[m] = a[m];
    for (int s = 0; s < 4; s++) {
      a[m][s] = (int)((src[(threadIdx.x * 4 + m) * 4 + s] & 0x88888888u) | 0x22222222u);
      b[m][s] = (int)((src[4096 + (threadIdx.x * 4 + m) * 4 + s] & 0x88888888u) | 0x22222222u);
    }
  }
  if (src[8191] == 0xdeadbeefu) for (int m = 0; m < 4; m++) a[m] = b[m] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};  // zeros run
  if (src[8191] == 0xfeedf00du)  // A codes 0.0 / 1.0 (half of them zero), B codes +-1.0
    for (int m = 0; m < 4; m++)
      for (int s = 0; s < 4; s++) a[m][s] = (a[m][s] >> 2) & 0x22222222;
  if (!SCALED)  // codes 0x1 / 0x9 = +-0.5
    for (int m = 0; m < 4; m++)
      for (int s = 0; s < 4; s++) {
        a[m][s] = (a[m][s] & (int)0x88888888u) | 0x11111111;
        b[m][s] = (b[m][s] & (int)0x88888888u) | 0x11111111;
      }
  f32x16 acc[CHAINS];
  float sum = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    a[0][0] ^= (int)0x22222222u;  // changes eight codes: keeps the chains from being hoisted out of the loop
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
      f32x16 d = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};
      if (!SCALED) d = f32x16{192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192};
#pragma unroll
      for (int m = 0; m < 4; m++)
        d = SCALED ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m], b[(m + c) & 3], d, 4, 4, 0, 0x7f7f7f7f, 0, 0x78787878)
                   : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m], b[(m + c) & 3], d, 4, 4, 0, 0, 0, 0);
      acc[c] = d;
    }
#pragma unroll
    for (int c = 0; c < CHAINS; c++) asm volatile("" : "+v"(acc[c]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int c = 0; c < CHAINS; c++) sum += acc[c][0] + acc[c][15];
  out[blockIdx.x * 64 + threadIdx.x] = sum;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cycles = t1 - t0;
}

int main() {
  unsigned* src;
  float* out;
  unsigned long long* cyc;
  CK(hipMalloc(&src, 8192 * 4));
  CK(hipMalloc(&out, 65536 * 64 * 4));
  CK(hipMalloc(&cyc, 8));
  unsigned h[8192];
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int zeros = 0; zeros < 4; zeros++) {
    unsigned x = 12345;
    for (int i = 0; i < 8192; i++) { x = x * 1664525u + 1013904223u; h[i] = x ^ (x >> 13); }
    h[8191] = zeros == 1 ? 0xdeadbeefu : zeros == 2 ? 0xfeedf00du : 1u;  // zeros == 3: random codes, unscaled form
    CK(hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice));
    for (int wps = 1; wps <= 4; wps *= 2) {
      const int blocks = 1024 * wps, iters = 20000 / wps;
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        if (zeros == 3) rate_kernel<2, false><<<blocks, 64>>>(src, out, iters, cyc);
        else rate_kernel<2><<<blocks, 64>>>(src, out, iters, cyc);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long hc; CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
        const double mfmas = (double)blocks * iters * 8;
        if (rep) printf("%s  %d wave(s)/SIMD: %7.3f ms  %6.2f PFLOP/s = %6.2f T comparisons/s  (%.1f memtime ticks per MFMA of wave 0)\n",
                        zeros == 1 ? "zeros " : zeros == 2 ? "A 0/1 " : zeros == 3 ? "unscal" : "random", wps, ms, mfmas * 131072 / (ms * 1e-3) / 1e15, mfmas * 256 / (ms * 1e-3) / 1e12,
                        (double)hc / (iters * 8.0));
      }
    }
  }
  return 0;
}
