// valu_rate.hip -- micro-benchmark that pins the gfx950 integer-VALU issue rates
// the roofline of the matcher is priced against ("confirm the
// bcnt issue rate and the sustained clock on the device before quoting
// fractions").  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate
// Run on the GPU box: ./tools/valu_rate
//
// Each kernel runs a long unrolled stream of ONE instruction kind (independent
// or dependent chains) from W waves per SIMD on every CU and reports
// wave-instructions / clk / CU and lane-ops/s; the shader clock is measured in
// the kernel (s_memtime vs the 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
using u32 = uint32_t;
using u64 = uint64_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

enum { OP_XOR, OP_XOR_S, OP_BCNT_IND, OP_BCNT_DEP, OP_MIN, OP_MED3, OP_LSHLOR, OP_ADD3, OP_MIX19, OP_FMA, N_OPS };
static const char* op_name[N_OPS] = {"v_xor_b32 v,v,v", "v_xor_b32 v,s,v", "v_bcnt_u32_b32 (8 indep)", "v_bcnt_u32_b32 (dep chain)",
                                     "v_min_u32", "v_med3_u32", "v_lshl_or_b32", "v_add3_u32", "mix: 8xor(s)+8bcnt+lshl_or+med3+min", "v_fma_f32"};
static const int op_count[N_OPS] = {16, 16, 16, 16, 16, 16, 16, 16, 19 * 4, 16};

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(u32* out, int iters, u32 sval, u64* clk) {
  u32 a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5,
      a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
  u32 b0 = a7 ^ 0x55555555u, b1 = a6 ^ 0x33333333u, b2 = a5 * 3, b3 = a4 * 5, b4 = a3 * 7, b5 = a2 * 9, b6 = a1 * 11, b7 = a0 * 13;
  u32 s0 = sval, s1 = sval * 3, s2 = sval * 5, s3 = sval * 7, s4 = sval * 9, s5 = sval * 11, s6 = sval * 13, s7 = sval * 15;
  u64 t0 = __builtin_amdgcn_s_memtime();
  u64 r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
    if (OP == OP_XOR) {
      asm volatile(
          "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %9\n v_xor_b32 %2, %2, %10\n v_xor_b32 %3, %3, %11\n"
          "v_xor_b32 %4, %4, %12\n v_xor_b32 %5, %5, %13\n v_xor_b32 %6, %6, %14\n v_xor_b32 %7, %7, %15\n"
          "v_xor_b32 %0, %0, %9\n v_xor_b32 %1, %1, %10\n v_xor_b32 %2, %2, %11\n v_xor_b32 %3, %3, %12\n"
          "v_xor_b32 %4, %4, %13\n v_xor_b32 %5, %5, %14\n v_xor_b32 %6, %6, %15\n v_xor_b32 %7, %7, %8\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));
    } else if (OP == OP_XOR_S) {
      asm volatile(
          "v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %9, %1\n v_xor_b32 %2, %10, %2\n v_xor_b32 %3, %11, %3\n"
          "v_xor_b32 %4, %12, %4\n v_xor_b32 %5, %13, %5\n v_xor_b32 %6, %14, %6\n v_xor_b32 %7, %15, %7\n"
          "v_xor_b32 %0, %9, %0\n v_xor_b32 %1, %10, %1\n v_xor_b32 %2, %11, %2\n v_xor_b32 %3, %12, %3\n"
          "v_xor_b32 %4, %13, %4\n v_xor_b32 %5, %14, %5\n v_xor_b32 %6, %15, %6\n v_xor_b32 %7, %8, %7\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7));
    } else if (OP == OP_BCNT_IND) {
      asm volatile(
          "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %9, %1\n v_bcnt_u32_b32 %2, %10, %2\n v_bcnt_u32_b32 %3, %11, %3\n"
          "v_bcnt_u32_b32 %4, %12, %4\n v_bcnt_u32_b32 %5, %13, %5\n v_bcnt_u32_b32 %6, %14, %6\n v_bcnt_u32_b32 %7, %15, %7\n"
          "v_bcnt_u32_b32 %0, %9, %0\n v_bcnt_u32_b32 %1, %10, %1\n v_bcnt_u32_b32 %2, %11, %2\n v_bcnt_u32_b32 %3, %12, %3\n"
          "v_bcnt_u32_b32 %4, %13, %4\n v_bcnt_u32_b32 %5, %14, %5\n v_bcnt_u32_b32 %6, %15, %6\n v_bcnt_u32_b32 %7, %8, %7\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));
    } else if (OP == OP_BCNT_DEP) {
      asm volatile(
          "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %0, %9, %0\n v_bcnt_u32_b32 %0, %10, %0\n v_bcnt_u32_b32 %0, %11, %0\n"
          "v_bcnt_u32_b32 %0, %12, %0\n v_bcnt_u32_b32 %0, %13, %0\n v_bcnt_u32_b32 %0, %14, %0\n v_bcnt_u32_b32 %0, %15, %0\n"
          "v_bcnt_u32_b32 %0, %9, %0\n v_bcnt_u32_b32 %0, %10, %0\n v_bcnt_u32_b32 %0, %11, %0\n v_bcnt_u32_b32 %0, %12, %0\n"
          "v_bcnt_u32_b32 %0, %13, %0\n v_bcnt_u32_b32 %0, %14, %0\n v_bcnt_u32_b32 %0, %15, %0\n v_bcnt_u32_b32 %0, %8, %0\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));
    } else if (OP == OP_MIN) {
      asm volatile(
          "v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %9\n v_min_u32 %2, %2, %10\n v_min_u32 %3, %3, %11\n"
          "v_min_u32 %4, %4, %12\n v_min_u32 %5, %5, %13\n v_min_u32 %6, %6, %14\n v_min_u32 %7, %7, %15\n"
          "v_min_u32 %0, %0, %9\n v_min_u32 %1, %1, %10\n v_min_u32 %2, %2, %11\n v_min_u32 %3, %3, %12\n"
          "v_min_u32 %4, %4, %13\n v_min_u32 %5, %5, %14\n v_min_u32 %6, %6, %15\n v_min_u32 %7, %7, %8\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));
    } else if (OP == OP_MED3) {
      asm volatile(
          "v_med3_u32 %0, %0, %8, %9\n v_med3_u32 %1, %1, %9, %10\n v_med3_u32 %2, %2, %10, %11\n v_med3_u32 %3, %3, %11, %12\n"
          "v_med3_u32 %4, %4, %12, %13\n v_med3_u32 %5, %5, %13, %14\n v_med3_u32 %6, %6, %14, %15\n v_med3_u32 %7, %7, %15, %8\n"
          "v_med3_u32 %0, %0, %9, %10\n v_med3_u32 %1, %1, %10, %11\n v_med3_u32 %2, %2, %11, %12\n v_med3_u32 %3, %3, %12, %13\n"
          "v_med3_u32 %4, %4, %13, %14\n v_med3_u32 %5, %5, %14, %15\n v_med3_u32 %6, %6, %15, %8\n v_med3_u32 %7, %7, %8, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));
    } else if (OP == OP_LSHLOR) {
      asm volatile(
          "v_lshl_or_b32 %0, %0, 1, %8\n v_lshl_or_b32 %1, %1, 1, %9\n v_lshl_or_b32 %2, %2, 1, %10\n v_lshl_or_b32 %3, %3, 1, %11\n"
          "v_lshl_or_b32 %4, %4, 1, %12\n v_lshl_or_b32 %5, %5, 1, %13\n v_lshl_or_b32 %6, %6, 1, %14\n v_lshl_or_b32 %7, %7, 1, %15\n"
          "v_lshl_or_b32 %0, %0, 1, %9\n v_lshl_or_b32 %1, %1, 1, %10\n v_lshl_or_b32 %2, %2, 1, %11\n v_lshl_or_b32 %3, %3, 1, %12\n"
          "v_lshl_or_b32 %4, %4, 1, %13\n v_lshl_or_b32 %5, %5, 1, %14\n v_lshl_or_b32 %6, %6, 1, %15\n v_lshl_or_b32 %7, %7, 1, %8\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));
    } else if (OP == OP_ADD3) {
      asm volatile(
          "v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %9, %10\n v_add3_u32 %2, %2, %10, %11\n v_add3_u32 %3, %3, %11, %12\n"
          "v_add3_u32 %4, %4, %12, %13\n v_add3_u32 %5, %5, %13, %14\n v_add3_u32 %6, %6, %14, %15\n v_add3_u32 %7, %7, %15, %8\n"
          "v_add3_u32 %0, %0, %9, %10\n v_add3_u32 %1, %1, %10, %11\n v_add3_u32 %2, %2, %11, %12\n v_add3_u32 %3, %3, %12, %13\n"
          "v_add3_u32 %4, %4, %13, %14\n v_add3_u32 %5, %5, %14, %15\n v_add3_u32 %6, %6, %15, %8\n v_add3_u32 %7, %7, %8, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));
    } else if (OP == OP_FMA) {
      asm volatile(
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %10, %11\n v_fma_f32 %3, %3, %11, %12\n"
          "v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %13, %14\n v_fma_f32 %6, %6, %14, %15\n v_fma_f32 %7, %7, %15, %8\n"
          "v_fma_f32 %0, %0, %9, %10\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %11, %12\n v_fma_f32 %3, %3, %12, %13\n"
          "v_fma_f32 %4, %4, %13, %14\n v_fma_f32 %5, %5, %14, %15\n v_fma_f32 %6, %6, %15, %8\n v_fma_f32 %7, %7, %8, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));
    } else if (OP == OP_MIX19) {
      // the matcher's per-comparison stream: 4 queries (a0..7 reused as 4 "queries" of 8 words via b regs)
#define ONE(Q0, Q1, Q2, Q3, Q4, Q5, Q6, Q7, BEST, SEC)                                   \
  asm volatile(                                                                          \
      "v_xor_b32 %2, %12, %4\n v_bcnt_u32_b32 %3, %2, 0\n"                               \
      "v_xor_b32 %2, %13, %5\n v_bcnt_u32_b32 %3, %2, %3\n"                              \
      "v_xor_b32 %2, %14, %6\n v_bcnt_u32_b32 %3, %2, %3\n"                              \
      "v_xor_b32 %2, %15, %7\n v_bcnt_u32_b32 %3, %2, %3\n"                              \
      "v_xor_b32 %2, %16, %8\n v_bcnt_u32_b32 %3, %2, %3\n"                              \
      "v_xor_b32 %2, %17, %9\n v_bcnt_u32_b32 %3, %2, %3\n"                              \
      "v_xor_b32 %2, %18, %10\n v_bcnt_u32_b32 %3, %2, %3\n"                             \
      "v_xor_b32 %2, %19, %11\n v_bcnt_u32_b32 %3, %2, %3\n"                             \
      "v_lshl_or_b32 %3, %3, 23, %20\n v_med3_u32 %1, %0, %3, %1\n v_min_u32 %0, %0, %3\n" \
      : "+v"(BEST), "+v"(SEC), "=&v"(tmp0), "=&v"(tmp1)                                  \
      : "v"(Q0), "v"(Q1), "v"(Q2), "v"(Q3), "v"(Q4), "v"(Q5), "v"(Q6), "v"(Q7), "s"(s0), \
        "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7), "s"(i))
      u32 tmp0, tmp1;
      ONE(b0, b1, b2, b3, b4, b5, b6, b7, a0, a1);
      ONE(b1, b2, b3, b4, b5, b6, b7, b0, a2, a3);
      ONE(b2, b3, b4, b5, b6, b7, b0, b1, a4, a5);
      ONE(b3, b4, b5, b6, b7, b0, b1, b2, a6, a7);
#undef ONE
    }
  }
  u64 t1 = __builtin_amdgcn_s_memtime();
  u64 r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

typedef void (*kern_t)(u32*, int, u32, u64*);
static kern_t kernels[N_OPS] = {rate_kernel<OP_XOR>,  rate_kernel<OP_XOR_S>,  rate_kernel<OP_BCNT_IND>, rate_kernel<OP_BCNT_DEP>,
                                rate_kernel<OP_MIN>,  rate_kernel<OP_MED3>,   rate_kernel<OP_LSHLOR>,   rate_kernel<OP_ADD3>,
                                rate_kernel<OP_MIX19>, rate_kernel<OP_FMA>};

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device: %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
  const int max_blocks = cus * 8;
  u32* out;
  u64* clk;
  CK(hipMalloc(&out, (size_t)max_blocks * 256 * 4));
  CK(hipMalloc(&clk, (size_t)max_blocks * 16));
  std::vector<u64> h(2 * max_blocks);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int iters = argc > 1 ? atoi(argv[1]) : 20000;
  printf("%-42s %5s %10s %12s %12s %10s\n", "op", "w/SIMD", "ms", "winst/clk/CU", "Tlaneop/s", "clk MHz");
  for (int op = 0; op < N_OPS; op++) {
    for (int wps : {1, 2, 4, 8}) {
      int blocks = cus * wps;  // 256 threads = 4 waves = 1 wave per SIMD per block
      int it = (op == OP_MIX19) ? iters / 4 : iters;
      kernels[op]<<<blocks, 256>>>(out, 200, 0x1234567u, clk);  // warm
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      kernels[op]<<<blocks, 256>>>(out, it, 0x1234567u, clk);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h.data(), clk, (size_t)blocks * 16, hipMemcpyDeviceToHost));
      std::vector<double> mhz(blocks), cyc(blocks);
      for (int b = 0; b < blocks; b++) {
        mhz[b] = (double)h[2 * b] / (double)h[2 * b + 1] * 100.0;
        cyc[b] = (double)h[2 * b];
      }
      std::nth_element(mhz.begin(), mhz.begin() + blocks / 2, mhz.end());
      std::nth_element(cyc.begin(), cyc.begin() + blocks / 2, cyc.end());
      double winst = (double)op_count[op] * it * wps * 4;  // per CU
      double per_clk = winst / cyc[blocks / 2];
      double tl = (double)op_count[op] * it * (double)blocks * 256 / (ms * 1e-3) / 1e12;
      printf("%-42s %5d %10.3f %12.3f %12.2f %10.0f\n", op_name[op], wps, ms, per_clk, tl, mhz[blocks / 2]);
    }
  }
  return 0;
}
