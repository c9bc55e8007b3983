// tools/valu_beside_mfma.hip -- what VALU instructions cost beside the scan's FP4 MFMAs: a step
// of 4 x v_mfma_scale_f32_32x32x64_f8f6f4 plus K instructions of one kind (8 independent
// chains), two waves per SIMD on every CU, cycles per step from wall time at the clock the
// MFMA-only loop implies.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/valu_beside_mfma.hip -o tools/bin/valu_beside_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OP, int K>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k(const unsigned* __restrict__ src, int* __restrict__ out, int iters) {
  i32x8 a[4], b[4];
  for (int m = 0; m < 4; m++) {
    a[m] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
    b[m] = a[m];
    for (int s = 0; s < 4; s++) {
      a[m][s] = (int)((src[(threadIdx.x * 4 + m) * 4 + s] & 0x88888888u) | 0x22222222u);
      b[m][s] = (int)((src[4096 + (threadIdx.x * 4 + m) * 4 + s] & 0x88888888u) | 0x22222222u);
    }
  }
  int c[8];
  for (int i = 0; i < 8; i++) c[i] = (int)src[threadIdx.x + i * 64];
  int y = (int)src[threadIdx.x + 777], z = (int)src[threadIdx.x + 999];
  f32x16 d = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};
  for (int it = 0; it < iters; it++) {
    a[0][0] ^= (int)0x88888888u;
    f32x16 e = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4};
#pragma unroll
    for (int m = 0; m < 4; m++)
      e = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[m], b[m], e, 4, 4, 0, 0x7f7f7f7f, 0, 0x78787878);
#pragma unroll
    for (int i = 0; i < K; i++) {
      int& r = c[i & 7];
      if (OP == 0) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(r) : "v"(y), "v"(z));
      if (OP == 1) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(y), "v"(z));
      if (OP == 2) asm volatile("v_min_i32 %0, %0, %1" : "+v"(r) : "v"(y));
      if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(y));
      if (OP == 4) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r) : "v"(y));
    }
    asm volatile("" : "+v"(e));
    d = e;
  }
  int s = 0;
  for (int i = 0; i < 8; i++) s += c[i];
  out[blockIdx.x * 64 + threadIdx.x] = s + (int)d[0] + (int)d[15];
}

template <int OP, int K>
static float run(const unsigned* src, int* out, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0));
    k<OP, K><<<2048, 64>>>(src, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  return best;
}

template <int OP>
static void sweep(const char* name, const unsigned* src, int* out) {
  const int iters = 8000;
  const float t0 = run<OP, 0>(src, out, iters), t8 = run<OP, 8>(src, out, iters), t16 = run<OP, 16>(src, out, iters),
              t24 = run<OP, 24>(src, out, iters), t32 = run<OP, 32>(src, out, iters), t48 = run<OP, 48>(src, out, iters);
  // two waves per SIMD share it: per SIMD a "step pair" = 8 MFMAs + 2K instructions
  printf("%-12s ms for K = 0, 8, 16, 24, 32, 48: %7.3f %7.3f %7.3f %7.3f %7.3f %7.3f   (relative to K = 0: %.2f %.2f %.2f %.2f %.2f)\n", name, t0, t8,
         t16, t24, t32, t48, t8 / t0, t16 / t0, t24 / t0, t32 / t0, t48 / t0);
}

int main() {
  unsigned* src;
  int* out;
  CK(hipMalloc(&src, 8192 * 4));
  CK(hipMalloc(&out, 2048 * 64 * 4));
  unsigned h[8192];
  unsigned x = 12345;
  for (int i = 0; i < 8192; i++) { x = x * 1664525u + 1013904223u; h[i] = x ^ (x >> 13); }
  CK(hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice));
  sweep<0>("v_min3_i32", src, out);
  sweep<1>("v_med3_f32", src, out);
  sweep<2>("v_min_i32", src, out);
  sweep<4>("v_max_i32", src, out);
  sweep<3>("v_add_u32", src, out);
  return 0;
}
