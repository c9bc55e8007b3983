// scan_variants.hip -- A/B harness for the scan kernel's inner-loop variants on
// MI355X: same random job set, every variant timed interleaved in one process
// (cdna_hip_programming.md rule 24), checksums compared.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ivisual-slam_amd/csrc tools/scan_variants.hip -o tools/scan_variants
#include "vsmatch_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace vsm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void fill_random(u64* p, size_t n, u64 seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  p[i] = z ^ (z >> 31);
}
__global__ void checksum(const uint2* s, size_t n, u64* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  atomicAdd((unsigned long long*)out, (unsigned long long)s[i].x * 3 + s[i].y);
}

typedef void (*scan_fn)(const u32x4*, const u32x4*, const Job*, u32, u32, u32, uint2*, const u32*);
struct Variant { const char* name; scan_fn fn; int R; };

int main(int argc, char** argv) {
  const u32 n_img = 512, N = argc > 1 ? atoi(argv[1]) : 2000, P = argc > 2 ? atoi(argv[2]) : 4096;
  const int reps = 3;
  u64* db;
  CK(hipMalloc(&db, (size_t)n_img * N * 32));
  fill_random<<<(n_img * N * 4 + 255) / 256, 256>>>(db, (size_t)n_img * N * 4, 1234);
  const u32 stride = (N + 63) / 64 * 64;
  std::vector<Job> jobs(2 * P);
  for (u32 p = 0; p < P; p++) {
    u32 i = (p * 7 + 1) % n_img, j = (p * 13 + 5) % n_img;
    jobs[2 * p] = {i * N, N, j * N, N, p * 2 * stride, NO_QLIST, 0, 0};
    jobs[2 * p + 1] = {j * N, N, i * N, N, p * 2 * stride + stride, NO_QLIST, 0, 0};
  }
  Job* d_jobs;
  CK(hipMalloc(&d_jobs, jobs.size() * sizeof(Job)));
  CK(hipMemcpy(d_jobs, jobs.data(), jobs.size() * sizeof(Job), hipMemcpyHostToDevice));
  uint2* scratch;
  const size_t n_scr = (size_t)P * 2 * stride;
  CK(hipMalloc(&scratch, n_scr * sizeof(uint2)));
  u64* d_sum;
  CK(hipMalloc(&d_sum, 8));
  std::vector<Variant> vs = {
      {"R4 fill0 (shipped)", scan_kernel<4, 0>, 4}, {"R4 fill1 s_nop", scan_kernel<4, 1>, 4}, {"R4 fill2 v_mov", scan_kernel<4, 2>, 4},
      {"R2 fill2", scan_kernel<2, 2>, 2},
      {"R8 fill0", scan_kernel<8, 0>, 8}, {"R8 fill1", scan_kernel<8, 1>, 8}, {"R8 fill2", scan_kernel<8, 2>, 8},
      {"R4 fill5 f32 key", scan_kernel<4, 5>, 4}, {"R8 fill5 f32 key", scan_kernel<8, 5>, 8},
      {"R2 fill5 f32 key", scan_kernel<2, 5>, 2}, {"R1 fill5 f32 key", scan_kernel<1, 5>, 1}, {"R1 fill2", scan_kernel<1, 2>, 1},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<std::vector<float>> ms(vs.size());
  std::vector<u64> sums(vs.size());
  const double comps = 2.0 * P * (double)N * N;
  for (int rep = 0; rep < reps + 1; rep++) {
    for (size_t v = 0; v < vs.size(); v++) {
      const u32 tq = 64 * vs[v].R, tiles = (N + tq - 1) / tq, wgs = (tiles + 3) / 4;
      CK(hipMemset(scratch, 0xff, n_scr * sizeof(uint2)));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(vs[v].fn, dim3(2 * P * wgs), dim3(256), 0, 0, (const u32x4*)db, (const u32x4*)db, d_jobs, 2 * P, 1u, (u32)n_scr, scratch, (const u32*)nullptr);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (rep) ms[v].push_back(t);
      if (rep == 1) {
        CK(hipMemset(d_sum, 0, 8));
        // only the N valid entries of each stride-sized slot are written; the rest stay 0xff
        checksum<<<(n_scr + 255) / 256, 256>>>(scratch, n_scr, d_sum);
        CK(hipMemcpy(&sums[v], d_sum, 8, hipMemcpyDeviceToHost));
      }
    }
  }
  printf("N=%u pairs=%u comps/launch=%.3e\n", N, P, comps);
  for (size_t v = 0; v < vs.size(); v++) {
    std::sort(ms[v].begin(), ms[v].end());
    printf("%-22s min %8.3f ms  med %8.3f ms  %8.1f Gcomp/s  frac_of_2457.6=%.3f  checksum %s\n", vs[v].name, ms[v][0], ms[v][reps / 2],
           comps / (ms[v][0] * 1e-3) / 1e9, comps / (ms[v][0] * 1e-3) / 2457.6e9, sums[v] == sums[0] ? "ok" : "MISMATCH");
  }
  return 0;
}
