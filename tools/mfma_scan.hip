// tools/mfma_scan.hip -- the matrix-core scan (csrc/mfma_scan.h) against the popcount scan on the
// same jobs: bitwise comparison of the scratch results and launch times.
//   hipcc --offload-arch=gfx950 -O3 -I visual-slam_amd/csrc tools/mfma_scan.hip -o /tmp/mfma_scan
//   /tmp/mfma_scan [descriptors per image] [pairs] [d_lim] [plant a true match every k-th descriptor, 0 = none] [real | synth] [train rows per job]
// `real` as fifth argument: images are random draws from the reference's 1084 real ORB descriptors
// (tests/golden/ex3_kd{0,1}.npy, run from the repository root) with ~8 bits flipped per copy.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef VSM_PRODUCT  // -DVSM_PRODUCT: the shipped kernel only (csrc/mfma_scan.h), for tuning it
#include "mfma_scan.h"
#include "mfma16_scan.h"
#else
#include "history/mfma_scan_forms.h"
#endif
using namespace vsm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ inline u32 mix(u32 x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// Every plant-th descriptor index k holds the same base pattern in all images with a few bits
// flipped per image (true matches, distance ~ 16); the others are independent random bits.
__global__ void fill_db(u32* p, u32 n_img, u32 N, u32 plant) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)n_img * N * 8) return;
  const u32 w = i & 7, k = (i >> 3) % N, img = (i >> 3) / N;
  const u32 r = mix((u32)i * 2654435761u + 12345u);
  if (plant && k % plant == 0) {
    const u32 base = mix(k * 8 + w + 777u);
    const u32 noise = r & mix(r + 1) & mix(r + 2) & mix(r + 3) & mix(r + 4);
    p[i] = base ^ noise ^ (img & 0);
  } else {
    p[i] = r;
  }
}

// The group form stores an upper bound of the second best distance; the library's consumers make
// it exact with refine_second (vsmatch_kernels.h) where it matters -- here for every entry.
__global__ void refine_all(const u32x4* db, const Job* jobs, uint2* scr, int v2 = 0) {
  const Job job = jobs[blockIdx.x];
  for (u32 q = threadIdx.x; q < job.nq; q += blockDim.x) {
    const uint2 v = scr[job.out_off + q];
#ifdef VSM_PRODUCT
    if (v2) {
      scr[job.out_off + q].y = refine_second_v3(db + 2 * (size_t)(job.q_off + q), db + 2 * (size_t)job.t_off, job.nt, v) << IDX_BITS;
      continue;
    }
#endif
    const u32 s = refine_second(db + 2 * (size_t)(job.q_off + q), db + 2 * (size_t)job.t_off, job.nt, job.nt, v);
    scr[job.out_off + q].y = s << IDX_BITS;
  }
}

#ifdef VSM_PRODUCT
// EXPERIMENT (round 2, last hours; not in the library): the eight 256-query tiles of a job in ONE workgroup of
// eight waves, the train operands of a step expanded by one wave and shared through LDS -- the other seven
// waves read them with four ds_read_b128 instead of 28 VALU instructions.  Train sizes that are multiples of
// 256 rows and query sets of up to 2048 only (no partial step, no remainder tiles): a measurement of what the
// sharing is worth, against the shipped kernel on the same jobs.  Double-buffered: one barrier per 8 steps.
// WAVES = 8: one workgroup per job and CU; WAVES = 4: two workgroups per job (1024 queries each) and per CU, each
// wave expands one step of four -- a workgroup waiting at its barrier leaves the matrix cores to the other one.
// SPW: steps a wave expands per barrier (a barrier per WAVES * SPW steps).
template <int WAVES, int SPW = 1>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(2))) void mfma_scan_shared_kernel(
    const u32x4* __restrict__ qbase, const u32x4* __restrict__ tbase, const Job* __restrict__ jobs,
    uint2* __restrict__ scratch, u32 d_lim) {
  extern __shared__ u32x4 shared_a[];  // [2 buffers][WAVES steps][4 k-quarters][64 lanes]
  constexpr int NB = 8;
  constexpr u32 PARTS = 8 / WAVES, S = WAVES * SPW;
  const Job job = jobs[blockIdx.x / PARTS];
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const u32 c = lane & 31, h = lane >> 5, q0 = ((blockIdx.x % PARTS) * WAVES + wave) * 256;
  if (q0 >= job.nq) {  // fewer than eight tiles: the wave still expands its steps and keeps the barriers
  }
  u32 m88 = 0x88888888u, c11 = 0x11111111u;
  asm volatile("" : "+s"(m88), "+v"(c11));
  i32x8 B[NB][4];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const u32 qi = min(q0 + b * 32 + c, job.nq - 1);
    const u32x4 w = qbase[2 * (size_t)(job.q_off + qi) + h];
    B[b][0] = expand_bits_r(~w.x, m88, c11);
    B[b][1] = expand_bits_r(~w.y, m88, c11);
    B[b][2] = expand_bits_r(~w.z, m88, c11);
    B[b][3] = expand_bits_r(~w.w, m88, c11);
  }
  const int init = d_lim >= 256u ? GROUP_D_256 : GROUP_D_BASE | (int)(d_lim << 15);
  int best[NB], second[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) best[b] = second[b] = init;
  f32x16 zero;
#pragma unroll
  for (int i = 0; i < 16; i++) zero[i] = __int_as_float(GROUP_D_START + i);
  asm volatile("" : "+v"(zero));
  auto med3 = [](int p, int q, int r) {
    return __float_as_int(__builtin_amdgcn_fmed3f(__int_as_float(p), __int_as_float(q), __int_as_float(r)));
  };
  auto group_min = [](const int (&x)[16]) {
    const int m0 = min3i(x[0], x[1], x[2]), m1 = min3i(x[3], x[4], x[5]), m2 = min3i(x[6], x[7], x[8]);
    const int m3 = min3i(x[9], x[10], x[11]), m4 = min3i(x[12], x[13], x[14]);
    return min(min3i(m0, m1, m2), min3i(m3, m4, x[15]));
  };
  const char* tr = (const char*)(tbase + 2 * (size_t)job.t_off);
  const u32 super_steps = job.nt / (32 * S);
  u32 buf = 0;
  auto my_rows = [&](u32 ss, u32 k) { return *(const u32x4*)(tr + ((ss * S + wave * SPW + k) * 32 + c) * 32u + h * 16u); };
  u32x4 wk[SPW];
#pragma unroll
  for (int k = 0; k < SPW; k++) wk[k] = super_steps ? my_rows(0, k) : u32x4{0, 0, 0, 0};
  for (u32 ss = 0; ss < super_steps; ss++, buf ^= 1) {
#pragma unroll
    for (int k = 0; k < SPW; k++) {  // this wave's steps of the super-step: expand, share; the next ones go in flight
      const u32x4 w = wk[k];
      const i32x8 e0 = expand_bits_r(w.x, m88, c11), e1 = expand_bits_r(w.y, m88, c11);
      const i32x8 e2 = expand_bits_r(w.z, m88, c11), e3 = expand_bits_r(w.w, m88, c11);
      u32x4* dst = shared_a + ((buf * S + wave * SPW + k) * 4) * 64 + lane;
      dst[0] = u32x4{(u32)e0[0], (u32)e0[1], (u32)e0[2], (u32)e0[3]};
      dst[64] = u32x4{(u32)e1[0], (u32)e1[1], (u32)e1[2], (u32)e1[3]};
      dst[128] = u32x4{(u32)e2[0], (u32)e2[1], (u32)e2[2], (u32)e2[3]};
      dst[192] = u32x4{(u32)e3[0], (u32)e3[1], (u32)e3[2], (u32)e3[3]};
      if (ss + 1 < super_steps) wk[k] = my_rows(ss + 1, k);
    }
    __syncthreads();
    auto shared_step = [&](u32 st, i32x8 (&A)[4]) {
      const u32x4* src = shared_a + ((buf * S + st) * 4) * 64 + lane;
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const u32x4 v = src[m * 64];
        A[m] = i32x8{(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
      }
    };
    i32x8 A[4], An[4];
    shared_step(0, A);
    f32x16 D = zero;
#pragma unroll
    for (int m = 0; m < 4; m++) D = mfma_step(A[m], B[0][m], D);
    for (u32 st = 0; st < S; st++) {
      const u32 step16 = (ss * S + st) * 16;
#pragma unroll
      for (int b = 0; b < NB; b++) {
        f32x16 Dn = zero;
        if (b + 1 < NB) {
#pragma unroll
          for (int m = 0; m < 4; m++) Dn = mfma_step(A[m], B[b + 1][m], Dn);
        } else if (st + 1 < S) {
#pragma unroll
          for (int m = 0; m < 4; m++) Dn = mfma_step(An[m], B[0][m], Dn);
        }
        if (b == NB - 2 && st + 1 < S) shared_step(st + 1, An);
        asm volatile("" : "+v"(Dn));
        int x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = __float_as_int(D[i]);
        const int m = group_min(x);
        second[b] = med3(best[b], m, second[b]);
        best[b] = min(best[b], m | (int)step16);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        D = Dn;
      }
#pragma unroll
      for (int m = 0; m < 4; m++) A[m] = An[m];
    }
  }
  if (q0 >= job.nq) return;
  uint2* out = scratch + job.out_off;
  const u32 sentinel = d_lim >= 256u ? KEY_INIT : (d_lim << IDX_BITS);
#pragma unroll
  for (int b = 0; b < NB; b++) {
    const u32 bb = (u32)best[b], sb = (u32)second[b];
    const u32 i = bb & 15u;
    const u32 row = 2 * ((bb >> 4) & 1023u) * 16 + 4 * h + (i & 3u) + 8 * (i >> 2);
    const u32 kl = (int)bb < init ? ((((bb >> 15) & 255u) << IDX_BITS) | row) : sentinel;
    const u32 sl = (int)(sb & ~32767u) < init ? (((sb >> 15) & 255u) << IDX_BITS) : sentinel;
    const u32 ko = (u32)__shfl_xor((int)kl, 32), so = (u32)__shfl_xor((int)sl, 32);
    const u32 kb = min(kl, ko), ks = min(min(sl, so), max(kl, ko));
    const u32 qi = q0 + b * 32 + c;
    if (h == 0 && qi < job.nq) out[qi] = make_uint2(kb, ks);
  }
}
#endif

// data of a little-endian .npy file of uint64 (version 1 header)
static std::vector<unsigned long long> load_npy_u64(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  unsigned char head[10];
  if (fread(head, 1, 10, f) != 10 || memcmp(head, "\x93NUMPY", 6) != 0) { fprintf(stderr, "%s: not an npy file\n", path); exit(2); }
  const size_t hlen = head[8] | (head[9] << 8);
  fseek(f, 0, SEEK_END);
  const size_t bytes = (size_t)ftell(f) - 10 - hlen;
  fseek(f, (long)(10 + hlen), SEEK_SET);
  std::vector<unsigned long long> v(bytes / 8);
  if (fread(v.data(), 8, v.size(), f) != v.size()) { fprintf(stderr, "%s: short read\n", path); exit(2); }
  fclose(f);
  return v;
}
static unsigned long long sm64(unsigned long long& x) {
  unsigned long long z = (x += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

int main(int argc, char** argv) {
  const u32 n_img = 512, N = argc > 1 ? atoi(argv[1]) : 2000, P = argc > 2 ? atoi(argv[2]) : 4096;
  const u32 d_lim = argc > 3 ? atoi(argv[3]) : 256;
  const u32 plant = argc > 4 ? atoi(argv[4]) : 4;
  const u32 NT = argc > 6 ? atoi(argv[6]) : N;  // train rows per job (default: the image): the fixed cost of a tile is the time at NT -> 0
  u32* db;
  CK(hipMalloc(&db, (size_t)n_img * N * 32));
  if (argc > 5 && !strcmp(argv[5], "real")) {
    std::vector<unsigned long long> real = load_npy_u64("tests/golden/ex3_kd0.npy"), r1 = load_npy_u64("tests/golden/ex3_kd1.npy");
    real.insert(real.end(), r1.begin(), r1.end());
    const size_t nr = real.size() / 4;
    std::vector<unsigned long long> h_db((size_t)n_img * N * 4);
    unsigned long long st = 12345;
    for (size_t i = 0; i < (size_t)n_img * N; i++) {
      const size_t src = sm64(st) % nr;
      for (int w = 0; w < 4; w++) h_db[4 * i + w] = real[4 * src + w] ^ (sm64(st) & sm64(st) & sm64(st) & sm64(st) & sm64(st));  // ~2 bits per word
    }
    CK(hipMemcpy(db, h_db.data(), h_db.size() * 8, hipMemcpyHostToDevice));
    printf("database: random draws from %zu real ORB descriptors\n", nr);
  } else
  fill_db<<<(u32)(((size_t)n_img * N * 8 + 255) / 256), 256>>>(db, n_img, N, plant);
  const u32 stride = (N + 63) / 64 * 64;
  std::vector<Job> jobs(2 * P);
  for (u32 p = 0; p < P; p++) {
    u32 i = (p * 7 + 1) % n_img, j = (p * 13 + 5) % n_img;
    jobs[2 * p] = {i * N, N, j * N, NT, p * 2 * stride, NO_QLIST, 0, 0};
    jobs[2 * p + 1] = {j * N, N, i * N, NT, p * 2 * stride + stride, NO_QLIST, 0, 0};
  }
  Job* d_jobs;
  CK(hipMalloc(&d_jobs, jobs.size() * sizeof(Job)));
  CK(hipMemcpy(d_jobs, jobs.data(), jobs.size() * sizeof(Job), hipMemcpyHostToDevice));
  const size_t n_scr = (size_t)P * 2 * stride;
  uint2 *scr_ref, *scr;
  CK(hipMalloc(&scr_ref, n_scr * sizeof(uint2)));
  CK(hipMalloc(&scr, n_scr * sizeof(uint2)));
  std::vector<uint2> h_ref(n_scr), h(n_scr);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double comps = 2.0 * P * (double)N * NT;
  const u32 n_jobs = 2 * P;

  // reference: the shipped popcount kernel
  float t_ref = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipMemset(scr_ref, 0xff, n_scr * sizeof(uint2)));
    const u32 tiles = adaptive_tiles(N, 4), groups = (n_jobs + 7) / 8;
    CK(hipEventRecord(e0));
#ifdef VSM_PRODUCT
    scan_kernel_adaptive<8, 4><<<dim3(groups * 8 * tiles), dim3(64)>>>((const u32x4*)db, (const u32x4*)db, d_jobs, n_jobs, 1u, (u32)n_scr, scr_ref, nullptr, tiles);
#else
    scan_kernel_adaptive<8, 1, 4><<<dim3(groups * 8 * tiles), dim3(64)>>>((const u32x4*)db, (const u32x4*)db, d_jobs, n_jobs, 1u, (u32)n_scr, scr_ref, nullptr, tiles);
#endif
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    t_ref = std::min(t_ref, t);
  }
  CK(hipMemcpy(h_ref.data(), scr_ref, n_scr * sizeof(uint2), hipMemcpyDeviceToHost));
  printf("N=%u pairs=%u d_lim=%u comps/launch=%.3e\n", N, P, d_lim, comps);
  printf("%-28s %8.3f ms %9.1f Gcomp/s\n", "popcount adaptive<8,1,4>", t_ref, comps / (t_ref * 1e-3) / 1e9);

  struct V { const char* name; int fmt, nb; };  // fmt 4: key tracking with the skip test, 14: distance-only tracking, 24: group tracking
#ifdef VSM_PRODUCT
  const V vs[] = {{"shipped NB8", 24, 8}, {"16x16 product", 104, 14}, {"shipped NB8", 24, 8}, {"16x16 product", 104, 14}, {"16x16 v3, 224-query tiles", 84, 14}, {"16x16 v3, 192-query tiles", 84, 12}, {"16x16 v3, 208-query tiles", 84, 13},
                  {"shipped NB8", 24, 8}, {"16x16 v3, 224-query tiles", 84, 14}, {"16x16 v3, 192-query tiles", 84, 12},
                  {"16x16x128, 192-query tiles", 74, 12}, {"16x16x128, 128-query tiles", 74, 8},
                  {"16x16x128, 160-query tiles", 74, 10}, {"16x16x128, 224-query tiles", 74, 14},
                  {"shipped NB8", 24, 8}, {"16x16x128, 192-query tiles", 74, 12}, {"zero-rich codes NB8", 64, 8},
                  {"shipped NB4", 24, 4}, {"zero-rich codes NB4", 64, 4}, {"shipped NB2", 24, 2}, {"zero-rich codes NB2", 64, 2},
                  {"NB8, 4-wave workgroups", 44, 8}, {"NB4, 4-wave workgroups", 44, 4},
                  {"EXPERIMENT: LDS-shared operands, 8 waves", 54, 8}, {"EXPERIMENT: LDS-shared operands, 4 waves", 54, 4},
                  {"EXPERIMENT: LDS-shared, 4 waves x 2 steps", 54, 2}, {"shipped NB8 again", 24, 8}};
#else
  const V vs[] = {{"mfma keys NB8", 4, 8}, {"mfma keys NB4", 4, 4}, {"mfma dist NB8", 14, 8}, {"mfma dist NB4", 14, 4}, {"mfma dist NB2", 14, 2},
                  {"mfma group NB8", 24, 8}, {"mfma group NB4", 24, 4}, {"mfma group NB2", 24, 2}};
#endif
  // VSM_FMT=104: only the variants of that format; VSM_LDS_PAD=36864: dynamic LDS per workgroup of the 16x16 product
  // kernel, so that four one-wave workgroups fit a CU -- ONE wave per SIMD: what a wave does to the MFMA pipe alone
  const int only_fmt = getenv("VSM_FMT") ? atoi(getenv("VSM_FMT")) : 0;
  const u32 lds_pad = getenv("VSM_LDS_PAD") ? (u32)atoi(getenv("VSM_LDS_PAD")) : 0u;
  for (const V& v : vs) {
    if (only_fmt && v.fmt != only_fmt) continue;
    float tm = 1e30f;
    const u32 tiles = adaptive_tiles(N, v.fmt == 74 ? 4 : v.nb / 2), groups = (n_jobs + 7) / 8;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipMemset(scr, 0xff, n_scr * sizeof(uint2)));
      CK(hipEventRecord(e0));
      const dim3 g(groups * 8 * tiles), blk(64);
#define ARGS (const u32x4*)db, (const u32x4*)db, d_jobs, n_jobs, 1u, (u32)n_scr, scr, nullptr, tiles, d_lim
#ifdef VSM_PRODUCT
      if (v.fmt == 54) {
        if (N % 256 || N > 2048) { printf("(the LDS experiment needs N = a multiple of 256, at most 2048)\n"); break; }
        CK(hipFuncSetAttribute((const void*)mfma_scan_shared_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        if (v.nb == 8) mfma_scan_shared_kernel<8><<<dim3(n_jobs), dim3(512), 65536>>>((const u32x4*)db, (const u32x4*)db, d_jobs, scr, d_lim);
        else if (v.nb == 4) mfma_scan_shared_kernel<4><<<dim3(n_jobs * 2), dim3(256), 32768>>>((const u32x4*)db, (const u32x4*)db, d_jobs, scr, d_lim);
        else {
          CK(hipFuncSetAttribute((const void*)mfma_scan_shared_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
          mfma_scan_shared_kernel<4, 2><<<dim3(n_jobs * 2), dim3(256), 65536>>>((const u32x4*)db, (const u32x4*)db, d_jobs, scr, d_lim);
        }
      } else
      if (v.fmt == 44) {  // tiles of a job in one workgroup (one CU: they share its L1), job-minor order
        const dim3 g4(n_jobs * ((tiles + 3) / 4)), b4(256);
#define ARGS4 (const u32x4*)db, (const u32x4*)db, d_jobs, n_jobs, 1u, (u32)n_scr, scr, nullptr, 0u, d_lim
        if (v.nb == 8) mfma_scan_kernel<8, 4, 0><<<g4, b4>>>(ARGS4, AcceptMask{nullptr, 0, nullptr}, 1u);
        else mfma_scan_kernel<4, 4, 0><<<g4, b4>>>(ARGS4, AcceptMask{nullptr, 0, nullptr}, 1u);
      } else
      if (v.fmt == 104) {
        const u32 t16 = mfma16_tiles(N);
        mfma16_scan_kernel<<<dim3(groups * 8 * t16), blk, lds_pad>>>((const u32x4*)db, (const u32x4*)db, d_jobs, n_jobs, 1u, (u32)n_scr, scr, t16, d_lim, AcceptMask{nullptr, 0, nullptr});
      } else
      if (v.fmt == 84) {
        const u32 t16 = (N + 16 * v.nb - 1) / (16 * v.nb);
        const dim3 g16(groups * 8 * t16);
#define ARGS16V2 (const u32x4*)db, (const u32x4*)db, d_jobs, n_jobs, scr, t16, d_lim
        if (v.nb == 12) mfma16v3_scan_kernel<12><<<g16, blk>>>(ARGS16V2);
        else if (v.nb == 13) mfma16v3_scan_kernel<13><<<g16, blk>>>(ARGS16V2);
        else mfma16v3_scan_kernel<14><<<g16, blk>>>(ARGS16V2);
      } else
      if (v.fmt == 74) {
        const u32 t16 = (N + 16 * v.nb - 1) / (16 * v.nb);
        const dim3 g16(groups * 8 * t16);
#define ARGS16 (const u32x4*)db, (const u32x4*)db, d_jobs, n_jobs, scr, t16, d_lim
        if (v.nb == 12) mfma16v1_scan_kernel<12><<<g16, blk>>>(ARGS16);
        else if (v.nb == 8) mfma16v1_scan_kernel<8><<<g16, blk>>>(ARGS16);
        else if (v.nb == 10) mfma16v1_scan_kernel<10><<<g16, blk>>>(ARGS16);
        else mfma16v1_scan_kernel<14><<<g16, blk>>>(ARGS16);
      } else
      if (v.fmt == 64) {
        if (v.nb == 8) mfma_scan_kernel<8, 1, 1><<<g, blk>>>(ARGS, AcceptMask{nullptr, 0, nullptr}, 1u);
        else if (v.nb == 4) mfma_scan_kernel<4, 1, 1><<<g, blk>>>(ARGS, AcceptMask{nullptr, 0, nullptr}, 1u);
        else mfma_scan_kernel<2, 1, 1><<<g, blk>>>(ARGS, AcceptMask{nullptr, 0, nullptr}, 1u);
      } else
      if (v.nb == 8) mfma_scan_kernel<8, 1, 0><<<g, blk>>>(ARGS, AcceptMask{nullptr, 0, nullptr}, 1u);
      else if (v.nb == 4) mfma_scan_kernel<4, 1, 0><<<g, blk>>>(ARGS, AcceptMask{nullptr, 0, nullptr}, 1u);
      else mfma_scan_kernel<2, 1, 0><<<g, blk>>>(ARGS, AcceptMask{nullptr, 0, nullptr}, 1u);
#else
      if (v.fmt == 4 && v.nb == 8) mfma_scan_kernel<4, 8><<<g, blk>>>(ARGS, nullptr, AcceptMask{nullptr, 0, nullptr});
      else if (v.fmt == 4) mfma_scan_kernel<4, 4><<<g, blk>>>(ARGS, nullptr, AcceptMask{nullptr, 0, nullptr});
      else if (v.fmt == 24 && v.nb == 8) mfma_scan_kernel<4, 8, 1, 2><<<g, blk>>>(ARGS, nullptr, AcceptMask{nullptr, 0, nullptr});
      else if (v.fmt == 24 && v.nb == 4) mfma_scan_kernel<4, 4, 1, 2><<<g, blk>>>(ARGS, nullptr, AcceptMask{nullptr, 0, nullptr});
      else if (v.fmt == 24) mfma_scan_kernel<4, 2, 1, 2><<<g, blk>>>(ARGS, nullptr, AcceptMask{nullptr, 0, nullptr});
      else if (v.nb == 8) mfma_scan_kernel<4, 8, 1, 1><<<g, blk>>>(ARGS, nullptr, AcceptMask{nullptr, 0, nullptr});
      else if (v.nb == 4) mfma_scan_kernel<4, 4, 1, 1><<<g, blk>>>(ARGS, nullptr, AcceptMask{nullptr, 0, nullptr});
      else mfma_scan_kernel<4, 2, 1, 1><<<g, blk>>>(ARGS, nullptr, AcceptMask{nullptr, 0, nullptr});
#endif
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      tm = std::min(tm, t);
    }
    if (v.fmt == 24 || v.fmt == 44 || v.fmt == 54 || v.fmt == 64 || v.fmt == 74 || v.fmt == 84 || v.fmt == 104) {
      CK(hipEventRecord(e0));
      refine_all<<<n_jobs, 256>>>((const u32x4*)db, d_jobs, scr, v.fmt == 84);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      printf("   (refine_second over EVERY entry: %.3f ms)\n", t);
    }
    CK(hipMemcpy(h.data(), scr, n_scr * sizeof(uint2), hipMemcpyDeviceToHost));
    // expected result under the limit: keys of distance >= d_lim are replaced by the start key
    size_t bad = 0, first_bad = 0, cand = 0;
    for (size_t i = 0; i < n_scr; i++) {
      uint2 e = h_ref[i];
      if (e.x == 0xffffffffu && e.y == 0xffffffffu) {  // never written
        if (h[i].x != 0xffffffffu || h[i].y != 0xffffffffu) { if (!bad) first_bad = i; bad++; }
        continue;
      }
      const u32 job = (u32)(i / stride);
      const u32 t_off_rel = 0;  // float keys index from the start of the range; t0 = 0 here
      (void)job; (void)t_off_rel;
      const u32 sentinel = d_lim >= 256 ? KEY_INIT : (d_lim << IDX_BITS);
      if ((e.x >> IDX_BITS) >= d_lim) e.x = sentinel;
      if ((e.y >> IDX_BITS) >= d_lim) e.y = sentinel;
      if ((e.x >> IDX_BITS) < d_lim) cand++;
      // the second best is compared by distance (its index is never read: keypoints.h:296)
      if (e.x != h[i].x || (e.y >> IDX_BITS) != (h[i].y >> IDX_BITS)) { if (!bad) first_bad = i; bad++; }
    }
    printf("%-28s %8.3f ms %9.1f Gcomp/s  x%.2f  queries with a candidate %zu  %s", v.name, tm, comps / (tm * 1e-3) / 1e9,
           t_ref / tm, cand, bad ? "MISMATCH" : "bit-exact");
    if (bad) printf(" (%zu entries, first %zu: want %08x %08x got %08x %08x)", bad, first_bad, h_ref[first_bad].x, h_ref[first_bad].y, h[first_bad].x, h[first_bad].y);
    printf("\n");
  }
  return 0;
}
