// tools/valu_beside_mfma.hip -- what VALU instructions cost beside the scan's FP4 MFMAs, by MFMA shape.
// A "step" = 1024 comparisons = 4 x v_mfma_f32_32x32x64_f8f6f4 or 8 x v_mfma_f32_16x16x128_f8f6f4 (no block
// scales, +-0.5 codes, random) with K v_min3_i32 spread evenly BETWEEN the MFMAs, as the scan issues them
// (round 1 measured the block-scaled form with the K instructions in one burst behind the MFMAs, which stalls
// both waves of a SIMD at once and says little about the scan).  Two waves per SIMD on every CU: wall time,
// T comparisons/s and the clock the chip holds (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/valu_beside_mfma.hip -o tools/bin/valu_beside_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int K>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k(const unsigned* __restrict__ src, int* __restrict__ out, int iters,
                                                                                 unsigned long long* __restrict__ stamps) {
  i32x8 a[4], b[4];
  for (int m = 0; m < 4; m++) {
    a[m] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
    b[m] = a[m];
    for (int s = 0; s < 4; s++) {
      a[m][s] = (int)((src[(threadIdx.x * 4 + m) * 4 + s] & 0x88888888u) | 0x11111111u);
      b[m][s] = (int)((src[4096 + (threadIdx.x * 4 + m) * 4 + s] & 0x88888888u) | 0x11111111u);
    }
  }
  int c[8];
  for (int i = 0; i < 8; i++) c[i] = (int)src[threadIdx.x + i * 64];
  int y = (int)src[threadIdx.x + 777], z = (int)src[threadIdx.x + 999];
  constexpr int NM = SHAPE == 32 ? 4 : 8;  // MFMAs per step
  // start values in registers of their own (as the scan has them): the first MFMA of a chain reads them as C and
  // writes another register set, no moves
  f32x16 start = {192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192, 192};
  asm volatile("" : "+v"(start));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
    a[0][0] ^= (int)0x88888888u;
    if (SHAPE == 32) {
      f32x16 e;
#pragma unroll
      for (int m = 0; m < 4; m++) {
        e = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m], b[m], m ? e : start, 4, 4, 0, 0, 0, 0);
        asm volatile("" : "+v"(e));
#pragma unroll
        for (int i = K * m / NM; i < K * (m + 1) / NM; i++) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(c[i & 7]) : "v"(y), "v"(z));
      }
    } else {
      // two tiles at a time, their k halves interleaved: no MFMA directly behind the one it depends on
      f32x4 e[4];
#pragma unroll
      for (int n = 0; n < 8; n++) {
        const int t = 2 * (n >> 2) + (n & 1), m = (n >> 1) & 1;
        const f32x4 st = {start[4 * t], start[4 * t + 1], start[4 * t + 2], start[4 * t + 3]};
        e[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[(t + m) & 3], b[(2 * m + t) & 3], m ? e[t] : st, 4, 4, 0, 0, 0, 0);
        asm volatile("" : "+v"(e[t]));
#pragma unroll
        for (int i = K * n / NM; i < K * (n + 1) / NM; i++) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(c[i & 7]) : "v"(y), "v"(z));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 7) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
  int s = 0;
  for (int i = 0; i < 8; i++) s += c[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

static unsigned long long* g_stamps;
template <int SHAPE, int K>
static void run(const unsigned* src, int* out, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  double clock = 0;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0));
    k<SHAPE, K><<<2048, 64>>>(src, out, iters, g_stamps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) {
      best = ms;
      unsigned long long h[2];
      CK(hipMemcpy(h, g_stamps, 16, hipMemcpyDeviceToHost));
      clock = (double)h[0] / (double)h[1] * 0.1;
    }
  }
  printf("%s  K = %2d v_min3_i32 per step: %7.3f ms  %6.2f T comparisons/s  clock %.3f GHz\n", SHAPE == 32 ? "32x32x64 " : "16x16x128", K,
         best, 2048.0 * iters * 1024.0 / (best * 1e-3) / 1e12, clock);
}

template <int SHAPE>
static void sweep(const unsigned* src, int* out) {
  const int iters = 20000;
  run<SHAPE, 0>(src, out, iters);
  run<SHAPE, 8>(src, out, iters);
  run<SHAPE, 10>(src, out, iters);
  run<SHAPE, 12>(src, out, iters);
  run<SHAPE, 13>(src, out, iters);
  run<SHAPE, 15>(src, out, iters);
  run<SHAPE, 16>(src, out, iters);
  run<SHAPE, 20>(src, out, iters);
  run<SHAPE, 24>(src, out, iters);
}

int main() {
  unsigned* src;
  int* out;
  CK(hipMalloc(&src, 8192 * 4));
  CK(hipMalloc(&out, 2048 * 64 * 4));
  CK(hipMalloc(&g_stamps, 16));
  unsigned h[8192];
  unsigned x = 12345;
  for (int i = 0; i < 8192; i++) { x = x * 1664525u + 1013904223u; h[i] = x ^ (x >> 13); }
  CK(hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice));
  for (int round = 0; round < 2; round++) {
    sweep<32>(src, out);
    sweep<16>(src, out);
  }
  return 0;
}
