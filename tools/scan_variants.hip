// scan_variants.hip -- A/B harness for the scan kernel's inner-loop variants on
// MI355X: same random job set, every variant timed interleaved in one process
// (cdna_hip_programming.md rule 24), checksums compared.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ivisual-slam_amd/csrc tools/scan_variants.hip -o tools/scan_variants
#include "history/scan_variants_kernels.h"
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

// The design north_star sketches, for comparison: train descriptors staged through LDS
// (cooperative coalesced 16-B loads, double buffered, one barrier per stage of TT
// descriptors), read back with two broadcast ds_read_b128 per descriptor per wave, queries in
// VGPRs as in the shipped kernel, same update<> instruction sequence (the train words are
// then VGPR operands of v_xor_b32 instead of SGPR operands).
template <int R, int FILL, int TT>
__global__ __launch_bounds__(256) void scan_kernel_lds(const u32x4* __restrict__ qbase,
                                                       const u32x4* __restrict__ tbase,
                                                       const Job* __restrict__ jobs, u32 n_jobs, u32 splits,
                                                       u32 split_stride, uint2* __restrict__ scratch,
                                                       const u32* __restrict__ qlist) {
  static_assert(TT * 2 % 256 == 0 || TT * 2 == 128, "stage must be a whole number of 256-thread loads");
  constexpr int LPT = (TT * 2 + 255) / 256;  // u32x4 loads per thread per stage
  __shared__ u32x4 tile[2][TT * 2];
  constexpr u32 TQ = 64 * R;
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  const u32 wg_of_job = blockIdx.x / n_jobs;
  const u32 job_id = blockIdx.x - wg_of_job * n_jobs;
  const Job job = jobs[job_id];
  if (wg_of_job * 4 * TQ >= job.nq) return;  // whole workgroup idle (block-uniform)
  const u32 q0 = (wg_of_job * 4 + wave) * TQ;
  u32 q[R][8];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const u32 qi = min(q0 + r * 64 + lane, job.nq - 1);
    const u32x4* p = qbase + 2 * (size_t)(job.q_off + qi);
    const u32x4 a = p[0], b = p[1];
    q[r][0] = a.x; q[r][1] = a.y; q[r][2] = a.z; q[r][3] = a.w;
    q[r][4] = b.x; q[r][5] = b.y; q[r][6] = b.z; q[r][7] = b.w;
  }
  u32 best[R], second[R];
#pragma unroll
  for (int r = 0; r < R; r++) best[r] = second[r] = (FILL == 5 ? FKEY_INIT : KEY_INIT);
  Pend pend = {PEND_INIT_D, FKEY_MAGIC};
  const u32x4* tr = tbase + 2 * (size_t)job.t_off;
  const u32 n_vec = 2 * job.nt;  // u32x4 units in the train range
  const u32 stages = (job.nt + TT - 1) / TT;
  u32x4 pre[LPT];
#pragma unroll
  for (int l = 0; l < LPT; l++) {
    const u32 v = threadIdx.x + 256 * l;
    if (v < TT * 2) pre[l] = tr[min(v, n_vec - 1)];
  }
#pragma unroll
  for (int l = 0; l < LPT; l++) {
    const u32 v = threadIdx.x + 256 * l;
    if (v < TT * 2) tile[0][v] = pre[l];
  }
  __syncthreads();
  for (u32 s = 0; s < stages; s++) {
    const u32 buf = s & 1;
    if (s + 1 < stages) {
#pragma unroll
      for (int l = 0; l < LPT; l++) {
        const u32 v = threadIdx.x + 256 * l;
        if (v < TT * 2) pre[l] = tr[min((s + 1) * TT * 2 + v, n_vec - 1)];
      }
    }
    const u32 base = s * TT, cnt = min((u32)TT, job.nt - base);
    u32 j = 0;
    for (; j + 2 <= cnt; j += 2) {
      const u32x4 a0 = tile[buf][2 * j], a1 = tile[buf][2 * j + 1];
      const u32x4 b0 = tile[buf][2 * j + 2], b1 = tile[buf][2 * j + 3];
      const u32x8 ta = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const u32x8 tb = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      update<R, FILL>(q, ta, base + j, best, second, pend);
      update<R, FILL>(q, tb, base + j + 1, best, second, pend);
    }
    for (; j < cnt; j++) {
      const u32x4 a0 = tile[buf][2 * j], a1 = tile[buf][2 * j + 1];
      const u32x8 ta = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      update<R, FILL>(q, ta, base + j, best, second, pend);
    }
    if (s + 1 < stages) {
#pragma unroll
      for (int l = 0; l < LPT; l++) {
        const u32 v = threadIdx.x + 256 * l;
        if (v < TT * 2) tile[buf ^ 1][v] = pre[l];
      }
    }
    __syncthreads();
  }
  uint2* out = scratch + job.out_off;
#pragma unroll
  for (int r = 0; r < R; r++) {
    const u32 qi = q0 + r * 64 + lane;
    if (FILL == 5) {
      best[r] = fkey_to_key(best[r], 0);
      second[r] = fkey_to_key(second[r], 0);
    }
    if (qi < job.nq) out[qi] = make_uint2(best[r], second[r]);
  }
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
      {"LDS R8 f32 key TT128", scan_kernel_lds<8, 5, 128>, 8}, {"LDS R8 f32 key TT256", scan_kernel_lds<8, 5, 256>, 8},
      {"LDS R4 f32 key TT128", scan_kernel_lds<4, 5, 128>, 4}, {"LDS R8 int key TT128", scan_kernel_lds<8, 2, 128>, 8},
      {"LDS R8 int nofill TT128", scan_kernel_lds<8, 0, 128>, 8},
      {"R8 fill7 add,fma pb", scan_kernel<8, 7>, 8}, {"R4 fill7 add,fma pb", scan_kernel<4, 7>, 4},
      {"R8 fill8 skip test", scan_kernel<8, 8>, 8}, {"R4 fill8 skip test", scan_kernel<4, 8>, 4},
      {"R2 fill8 skip test", scan_kernel<2, 8>, 2}, {"R1 fill8 skip test", scan_kernel<1, 8>, 1},
      {"R2 fill7 add,fma pb", scan_kernel<2, 7>, 2}, {"R1 fill7 add,fma pb", scan_kernel<1, 7>, 1},
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
