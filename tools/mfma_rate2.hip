// tools/mfma_rate2.hip -- round 3: what the FP4 matrix-core instruction sustains on the whole chip by
// SHAPE (v_mfma_f32_32x32x64_f8f6f4 against v_mfma_f32_16x16x128_f8f6f4, both without block scales) and by
// OPERAND CODES (the scan's +-0.5 codes on both sides; "zero-rich" train codes: a quarter +-0.5, the other
// three quarters 0 / x -- the shift-free codes of csrc/mfma_scan.h; zeros), W waves per SIMD, back to back,
// with the clock the chip holds (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_rate2.hip -o tools/bin/mfma_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// DATA 0: +-0.5 codes both sides; 1: train zero-rich (reg 0: +-0.5, reg 1: 0 / 2.0, reg 2: 0 / 1.0, reg 3: 0 / 0.5), query
// +-1.0, +-0.5, +-1.0, +-2.0; 2: zeros
template <int DATA>
__device__ __forceinline__ void load_codes(const unsigned* src, i32x8 (&a)[4], i32x8 (&b)[4]) {
  for (int m = 0; m < 4; m++) {
    a[m] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
    b[m] = a[m];
    for (int s = 0; s < 4; s++) {
      const unsigned wa = src[(threadIdx.x * 4 + m) * 4 + s], wb = src[4096 + (threadIdx.x * 4 + m) * 4 + s];
      if (DATA == 0) {
        a[m][s] = (int)((wa & 0x88888888u) | 0x11111111u);
        b[m][s] = (int)((wb & 0x88888888u) | 0x11111111u);
      } else if (DATA == 1) {
        a[m][s] = s == 0 ? (int)((wa & 0x88888888u) | 0x11111111u) : (int)(wa & (0x88888888u >> s));
        b[m][s] = (int)((wb & 0x88888888u) | (s == 0 ? 0x22222222u : s == 1 ? 0x11111111u : s == 2 ? 0x22222222u : 0x44444444u));
      } else if (DATA == 3) {  // both sides: a code is zero or one magnitude (0.5 / 1.0 / 2.0), half of them zero
        a[m][s] = (int)(wa & (0x11111111u << (s % 3)));
        b[m][s] = (int)(wb & (0x11111111u << ((s + 1) % 3)));
      } else if (DATA == 4) {  // train zero or one magnitude, query +-0.5
        a[m][s] = (int)(wa & (0x11111111u << (s % 3)));
        b[m][s] = (int)((wb & 0x88888888u) | 0x11111111u);
      } else if (DATA == 5) {  // both sides zero or 1.0, a quarter of the bits set (sparser descriptors)
        a[m][s] = (int)(wa & (wa >> 4 | wa << 28) & 0x22222222u);
        b[m][s] = (int)(wb & (wb >> 4 | wb << 28) & 0x22222222u);
      }
    }
  }
}

// ORDER (32 x 32 shape): 0 = chain after chain (four dependent MFMAs, as in the scan), 1 = two chains interleaved with
// consecutive MFMAs sharing the A operand, 2 = sharing the B operand
template <int SHAPE, int DATA, int ORDER = 0>
__global__ __launch_bounds__(64) void rate_kernel(const unsigned* __restrict__ src, float* __restrict__ out, int iters,
                                                  unsigned long long* __restrict__ stamps) {
  i32x8 a[4], b[4];
  load_codes<DATA>(src, a, b);
  float sum = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (SHAPE == 32) {
    f32x16 acc[2];
    for (int it = 0; it < iters; it++) {
      a[0][0] ^= (int)0x88888888u;
      if (ORDER == 0) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const float s0 = 768.f + c;  // distinct start values: the two chains are not one computation done twice
          f32x16 d = {s0, s0, s0, s0, s0, s0, s0, s0, s0, s0, s0, s0, s0, s0, s0, s0};
#pragma unroll
          for (int m = 0; m < 4; m++) d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m], b[(m + c) & 3], d, 4, 4, 0, 0, 0, 0);
          acc[c] = d;
        }
      } else {
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const float s0 = 768.f + c;
          acc[c] = f32x16{s0, s0, s0, s0, s0, s0, s0, s0, s0, s0, s0, s0, s0, s0, s0, s0};
        }
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
          for (int c = 0; c < 2; c++)
            acc[c] = ORDER == 1 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m], b[(m + c) & 3], acc[c], 4, 4, 0, 0, 0, 0)
                                : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(m + c) & 3], b[m], acc[c], 4, 4, 0, 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < 2; c++) asm volatile("" : "+v"(acc[c]));
    }
    for (int c = 0; c < 2; c++) sum += acc[c][0] + acc[c][15];
  } else {
    // the same 2 x 1024 comparisons per iteration: 8 tiles of 16 x 16, two MFMAs (k = 256) each
    f32x4 acc[8];
    for (int it = 0; it < iters; it++) {
      a[0][0] ^= (int)0x88888888u;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const float s0 = 768.f + c;
        f32x4 d = {s0, s0, s0, s0};
#pragma unroll
        for (int m = 0; m < 2; m++) d = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[(2 * c + m) & 3], b[(m + c) & 3], d, 4, 4, 0, 0, 0, 0);
        acc[c] = d;
      }
#pragma unroll
      for (int c = 0; c < 8; c++) asm volatile("" : "+v"(acc[c]));
    }
    for (int c = 0; c < 8; c++) sum += acc[c][0] + acc[c][3];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 64 + threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 7) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

template <int SHAPE, int DATA, int ORDER = 0>
static void run(const char* name, const unsigned* src, float* out, unsigned long long* st) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = 1024 * wps, iters = 40000 / wps;
    float best = 1e30f;
    unsigned long long h[2] = {0, 1};
    for (int rep = 0; rep < 3; rep++) {
      CK(hipEventRecord(e0));
      rate_kernel<SHAPE, DATA, ORDER><<<blocks, 64>>>(src, out, iters, st);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) { best = ms; CK(hipMemcpy(h, st, 16, hipMemcpyDeviceToHost)); }
    }
    const double comps = (double)blocks * iters * 2048.0;
    printf("%-34s %d wave(s)/SIMD: %8.3f ms  %6.2f T comparisons/s = %5.2f PFLOP/s   clock %.3f GHz\n", name, wps, best,
           comps / (best * 1e-3) / 1e12, comps * 512 / (best * 1e-3) / 1e15, (double)h[0] / (double)h[1] * 0.1);
  }
}

int main() {
  unsigned* src;
  float* out;
  unsigned long long* st;
  CK(hipMalloc(&src, 8192 * 4));
  CK(hipMalloc(&out, 4096 * 64 * 4));
  CK(hipMalloc(&st, 16));
  unsigned h[8192];
  unsigned x = 12345;
  for (int i = 0; i < 8192; i++) { x = x * 1664525u + 1013904223u; h[i] = x ^ (x >> 13); }
  CK(hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice));
  for (int round = 0; round < 2; round++) {
    run<32, 0>("32x32x64  +-0.5 codes", src, out, st);
    run<16, 0>("16x16x128 +-0.5 codes", src, out, st);
    run<32, 0, 1>("32x32x64  +-0.5, pairs share A", src, out, st);
    run<32, 0, 2>("32x32x64  +-0.5, pairs share B", src, out, st);
    run<32, 1>("32x32x64  zero-rich train codes", src, out, st);
    run<16, 1>("16x16x128 zero-rich train codes", src, out, st);
    run<32, 3>("32x32x64  0 / x codes both sides", src, out, st);
    run<16, 3>("16x16x128 0 / x codes both sides", src, out, st);
    run<32, 4>("32x32x64  0 / x train, +-0.5 query", src, out, st);
    run<16, 4>("16x16x128 0 / x train, +-0.5 query", src, out, st);
    run<32, 5>("32x32x64  0 / 1 both, quarter set", src, out, st);
    run<32, 2>("32x32x64  zeros", src, out, st);
    run<16, 2>("16x16x128 zeros", src, out, st);
  }
  return 0;
}
