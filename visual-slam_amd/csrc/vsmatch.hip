// vsmatch.hip -- C ABI (include/vsmatch.h) over the gfx950 kernels.
//
// Replaces visnav::matchDescriptors (/root/reference/include/visnav/keypoints.h:247-361)
// and its batched call pattern (src/sfm.cpp:1295-1321).  No CPU fallback: every
// entry point needs a HIP device and fails with VSM_ERR_HIP otherwise.
#include "vsmatch_kernels.h"
#include "mfma_scan.h"
#include "mfma16_scan.h"
#include "landmark_kernels.h"
#include "orb_kernels.h"
#include "gftt_kernels.h"
#include "bow_kernels.h"
#include "bowdb_kernels.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <limits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <cstring>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "vsmatch.h"

using namespace vsm;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail(VSM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                  __FILE__, __LINE__);                                                   \
  } while (0)
#define VSM_TRY(expr)         \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != VSM_OK) return rc_; \
  } while (0)

// Grow-only device buffer.
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return VSM_OK;
    if (p) {
      HIP_TRY(hipFree(p));
      p = nullptr;
      cap = 0;
    }
    size_t want = std::max(bytes + bytes / 4, (size_t)4096);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      want = bytes;
      e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
      p = nullptr;
      return fail(VSM_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    cap = want;
    return VSM_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct PinBuf {
  void* p = nullptr;
  void* dev = nullptr;  // the same memory as the device sees it (mapped buffers only)
  size_t cap = 0;
  unsigned flags = hipHostMallocDefault;
  int reserve(size_t bytes) {
    if (bytes <= cap) return VSM_OK;
    if (p) {
      HIP_TRY(hipHostFree(p));
      p = dev = nullptr;
      cap = 0;
    }
    size_t want = std::max(bytes + bytes / 4, (size_t)4096);
    HIP_TRY(hipHostMalloc(&p, want, flags));
    cap = want;
    if (flags != hipHostMallocDefault && hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) {
      (void)hipGetLastError();
      dev = nullptr;
    }
    return VSM_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = dev = nullptr;
    cap = 0;
  }
};

constexpr size_t DESC_BYTES = 32;

// Scratch of vsm_match_pair_noalloc: a file-scope device array, so that the entry point needs no
// hipMalloc.  Every GPU the library runs on holds its own copy (4 MiB + 256 B) at its own address
// until the process ends; vsm_create asks for that address (hipGetSymbolAddress, with the
// context's device current) and host code only ever uses the pointer it got -- the array's NAME is
// a host-side stand-in, not a device address.  Layout: [Job forward | Job backward | pad to 256 |
// scan results: splits x 2 x stride uint2].
constexpr size_t NOALLOC_MAX_SET = 8192;                   // descriptors per side
constexpr size_t NOALLOC_RESULT_BYTES = (size_t)4 << 20;   // 16 splits x 2 x 8192 x 8 B = 2 MiB at the maximum
constexpr size_t NOALLOC_BYTES = 256 + NOALLOC_RESULT_BYTES;
static __device__ __align__(256) unsigned char g_noalloc_scratch[NOALLOC_BYTES];
constexpr size_t SCRATCH_BUDGET = (size_t)2 << 30;  // scan results per chunk of pairs
constexpr u32 TARGET_WAVES = 4096;                  // below this the train range is split
constexpr u32 LONG_SCAN_WAVES = 98304;              // long scans are split until a launch has this many waves

}  // namespace

// The 16x16x128 forward scan (mfma16_scan.h) on a set of images: tiles per image and an estimate of its time against
// the 32x32x64 scan's, both of the form  (padded queries, weighted by tile size) x (padded train rows + a fixed cost per
// tile and block in rows).  Fitted to tools/mfma_scan.hip on one box (profiles/r03/scan16_product_by_size.txt): per
// padded comparison the 16x16 scan took 0.867 of the time, its fixed cost is 320 rows per block against 115 (two
// merges of four lanes per result, the tail behind the pipelined loop); against the 32x32 kernel WITH its runs of jobs
// (all-pairs lists through the library: tools/m16_rule_probe.py, profiles/r03/m16_rule_probe.txt) 400 rows; with the
// k halves of a tile issued back to back (mfma16_scan.h: + 3 %) 0.843 -- it wins from ~1750 train rows per image up:
// + 8 % at 2000 and + 10 % at 2500 in batches of 8128 pairs (forward kernel), + 9 to + 11 % on the bench's 32768-pair
// launches, a tie at 1400, - 3 % at 1200 and 1600, - 7 % at 1000.
struct Mfma16Stats {
  u32 max_tiles = 0;
  bool uniform = false;  // every image has exactly max_tiles tiles
  bool prefer = false;
};
struct Mfma16Cost {
  double q16 = 0, q32 = 0, t = 0;
  size_t n = 0;
  u32 max_tiles = 0, min_tiles = 0xffffffffu;
  void add(u32 nd, u32 nt_rows = 0xffffffffu) {  // nt_rows: train rows a tile scans, when not the image (train splits)
    // a tile of nb blocks: 11.6 VALU instructions per block and superstep + 32 for the expansion of the train rows,
    // relative to the 14-block tile
    const Mfma16Tiling tg = mfma16_tiling(nd);
    q16 += 16 * (tg.big_tiles * (11.6 * (tg.small + 2) + 32) + (tg.tiles - tg.big_tiles) * (11.6 * tg.small + 32)) / (11.6 + 32 / 14.0);
    // 32x32: full tiles of 256 queries, remainder tiles of 128 (- 3 %) and 64 (- 11 %), the last block left out when empty
    const u32 n64 = (nd + 63) >> 6, full = n64 / 4, rem = n64 & 3;
    q32 += full * 256.0 + ((rem & 2) ? 128 * 1.03 : 0.0) + ((rem & 1) ? 64 * 1.12 : 0.0) - (nd && ((nd - 1) & 63) < 32 ? 32.0 : 0.0);
    t += (std::min(nd, nt_rows) + 31) / 32 * 32;
    n++;
    max_tiles = std::max(max_tiles, tg.tiles);
    min_tiles = std::min(min_tiles, tg.tiles);
  }
  Mfma16Stats stats() const {
    Mfma16Stats s;
    s.max_tiles = max_tiles;
    s.uniform = n > 0 && min_tiles == max_tiles;
    s.prefer = n > 0 && 0.843 * q16 * (t + 400.0 * n) < 0.99 * q32 * (t + 115.0 * n);
    return s;
  }
};

struct vsm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::recursive_mutex mu;
  // database
  const void* db = nullptr;
  DevBuf db_owned;
  DevBuf d_img_off;
  DevBuf gmin;  // one byte per database descriptor (group_min_kernel), valid for `db`
  std::vector<u32> img_off;
  u32 max_n = 0;
  // per full-tile size rmax = 8, 4, 2, 1 (index 0..3): most adaptive tiles any image of the
  // database needs, whether every image needs exactly that many, and the mean over images
  u32 max_tiles[4] = {0, 0, 0, 0};
  bool uniform_tiles[4] = {false, false, false, false};
  double avg_tiles[4] = {0, 0, 0, 0};
  Mfma16Stats m16;  // the same for the 224-query tiles of the 16x16x128 scan, and whether it pays on this database
  // work buffers
  DevBuf stage, scratch, jobs, counts, misc, qlist, flist, accmask, lut, w_pairs, w_offsets, w_out;
  DevBuf image, gftt;  // device image of the ORB/detection entry points; detector work buffers
  DevBuf voc;          // vocabulary tree (vsm_voc_upload)
  size_t voc_nodes = 0, voc_off[6] = {0, 0, 0, 0, 0, 0};
  // BoW database (vsm_bowdb_upload): inverted index by word + the images' own vectors
  DevBuf bowdb, bowdb_work;
  size_t bowdb_off[6] = {0, 0, 0, 0, 0, 0}, bowdb_entries = 0;
  u32 bowdb_images = 0, bowdb_dense = 0;
  bool bowdb_ready = false;
  PinBuf pin, pin_image;
  // vsm_match_pair's results: host memory the emit kernel writes directly (mapped, coherent), so the
  // call needs no device-to-host copy and can poll a flag instead of waiting on the stream
  PinBuf pin_out, pin_in;  // pin_in: its inputs, read by stage_in_kernel (mapped)
  u64 pair_seq = 0;
  void* noalloc = nullptr;  // this device's copy of g_noalloc_scratch (hipGetSymbolAddress in vsm_create)
  // measurement
  bool profiling = false;
  std::vector<hipEvent_t> ev;
  size_t ev_used = 0;
  // per kind: 0 = full / forward scans, 1 = pruned backward scans
  double scan_ms[2] = {0, 0};
  u64 scan_launches[2] = {0, 0}, scan_comps[2] = {0, 0};
  int scan_kernel[2] = {0, 0};  // vsm_last_scan_kernel
  std::vector<int> ev_kind;
  int variant = 0;
};

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

void make_accept(int threshold, double dist_2_best, AcceptParams* prm) {
  prm->threshold = threshold;
  for (int b = 0; b <= 256; b++) {
    // min_second[b] = smallest s in 0..256 with `dist_2_best * b <= s` as keypoints.h:296 evaluates it
    // (both sides double), 257 if there is none.  x is ONE double; the predicate is monotone in s, so
    // the answer is ceil(x) clamped -- confirmed with the predicate itself, so that no rounding
    // argument is needed (a 257 x 257 search of the predicate cost 13 us per call).
    const double x = dist_2_best * b;
    int need = 257;
    if (x <= 0.0) need = 0;               // also -inf, -0.0
    else if (x <= 256.0) {                // NaN and everything above 256 fail both tests: 257
      need = (int)std::ceil(x);
      while (need > 0 && x <= (double)(need - 1)) need--;
      while (need <= 256 && !(x <= (double)need)) need++;
    }
    prm->min_second[b] = (uint16_t)need;
  }
  prm->min_second[257] = 257;
}

// Smallest distance L such that the accept test (keypoints.h:295-296) cannot tell a best or
// second best distance >= L from L itself: a best >= threshold is rejected whatever it is, and a
// second best >= every min_second[b], b < threshold, passes whatever it is.  The matrix-core scan
// starts its (best, second) at L and so skips every block of distances that are all >= L.
u32 accept_limit(const AcceptParams& prm) {
  int lim = std::max(prm.threshold, 1);
  for (int b = 0; b < std::min(prm.threshold, 256); b++) lim = std::max(lim, (int)prm.min_second[b]);
  return (u32)std::min(lim, 256);
}

// Radius gate of find_matches_landmarks (vo_utils.h:136-138): `norm > r` with norm =
// sqrt(squared) correctly rounded is, by monotonicity of sqrt, `squared > T` with T the
// largest double whose square root is <= r.  NaN r: nothing is ever skipped (T = +inf);
// negative r: every finite distance is skipped (T = -inf).
double gate_threshold(double r) {
  const double inf = std::numeric_limits<double>::infinity();
  if (std::isnan(r)) return inf;
  if (r < 0) return -inf;
  if (std::isinf(r)) return inf;
  double x = r * r;
  for (int it = 0; it < 4096 && std::sqrt(x) > r; it++) x = std::nextafter(x, -inf);
  for (int it = 0; it < 4096; it++) {
    const double up = std::nextafter(x, inf);
    if (std::isinf(up) || std::sqrt(up) > r) break;
    x = up;
  }
  return x;
}

int ev_pair(vsm_ctx* c, hipEvent_t* a, hipEvent_t* b) {
  while (c->ev.size() < c->ev_used + 2) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    c->ev.push_back(e);
  }
  *a = c->ev[c->ev_used++];
  *b = c->ev[c->ev_used++];
  return VSM_OK;
}

void stats_reset(vsm_ctx* c) {
  c->ev_used = 0;
  c->ev_kind.clear();
  for (int k = 0; k < 2; k++) {
    c->scan_ms[k] = 0;
    c->scan_launches[k] = 0;
    c->scan_comps[k] = 0;
  }
}

// After the stream has been synchronised: sum the event pairs.
int stats_collect(vsm_ctx* c) {
  for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
    c->scan_ms[c->ev_kind[i / 2]] += ms;
  }
  c->ev_used = 0;
  c->ev_kind.clear();
  return VSM_OK;
}

struct ScanShape {
  bool adaptive;  // one-wave workgroups, full tiles of 64 * rmax queries + 128/64-query remainder tiles
  int rmax;       // adaptive: queries per lane-pair of the full tiles in units of 64 (4, 2 or 1)
  int R;          // fixed tiles (not adaptive): 64 * R queries per wave, 4-wave workgroups (1, or 4 through the knob)
  u32 wgs_per_job, splits;
  u32 tiles;      // most tiles a job has (adaptive: grid of one-wave workgroups = n_jobs x tiles)
  bool uniform;   // every job has exactly `tiles` tiles (grouped block order, see the kernel)
  bool fkey;      // a split scans <= FKEY_MAX_SET rows: float keys (popcount) / 10-bit step tags (matrix cores)
  bool mfma;      // matrix-core scan (mfma_scan.h); else the popcount scan (experiment knob only)
  u32 jobs_per_wave = 1;  // matrix-core adaptive scan: consecutive jobs a wave runs over (mfma_scan_run)
  bool mfma16 = false;    // scans without a query list go through the 16x16x128 kernel (mfma16_scan.h); tiles, uniform: its own
};

// Most tiles the adaptive kernels can need for a job of at most max_nq queries (the tile
// count is not monotone in nq: 192 queries are 2 tiles at rmax = 4, 256 are 1).
u32 adaptive_tiles_bound(u32 max_nq, u32 rmax) {
  const u32 n64 = (max_nq + 63) >> 6;
  u32 best = 0;
  for (u32 k = n64 > 16 ? n64 - 16 : 0; k <= n64; k++) best = std::max(best, adaptive_tiles(k * 64, rmax));
  return best;
}

// Experiment knob (vsm_set_variant), kept for A/B runs and for the parity tests, which drive every
// kernel shape through it:
//   low byte    1 / 4 / 8: latency shape (64-query tiles, 4-wave workgroups) / fixed 256-query
//               popcount tiles / adaptive tiles, whatever the job count says
//   bits 8-15   number of train splits
//   bit 16      no pruning of the backward pass
//   bit 17      popcount scan with integer keys (FILL 2) instead of float keys (FILL 8)
//   bit 18      the OTHER operand codes of the matrix-core scan (mfma_scan.h CODES): +-0.5 codes where the
//               library uses the shift-free zero-rich ones (128- and 64-query tiles) and vice versa (256)
//   bits 20-23  consecutive jobs per wave of the matrix-core batched scan (0 = the library's choice)
//   bit 25      popcount scan everywhere (the formulation of BASELINE.json's north_star)
//   bit 26      popcount scan in the pruned backward pass only
//   bit 27      popcount scan for the latency shape only
//   bits 28-30  adaptive full-tile size: 4 / 2 / 1 = 256 / 128 / 64 queries
//   bit 19 / 24 never / always the 16x16x128 forward scan where its shape allows it (else: the estimate of Mfma16Cost)
constexpr int KNOB_MFMA16_OFF = 0x80000, KNOB_MFMA16_ON = 0x1000000;
constexpr int KNOB_NO_PRUNE = 0x10000, KNOB_INT_KEYS = 0x20000, KNOB_FLIP_CODES = 0x40000, KNOB_POPCOUNT = 0x2000000,
              KNOB_POPCOUNT_BACKWARD = 0x4000000, KNOB_POPCOUNT_LATENCY = 0x8000000;

// Tile form and train splits.  Large batches: adaptive tiles (256 queries or fewer), one-wave
// workgroups, no split.  Small problems (one stereo pair, odometry.cpp:814) are latency bound:
// 64-query tiles and the train range split over grid.y so that all 256 CUs work.
ScanShape choose_shape(u64 n_jobs, u32 max_nq, u32 max_nt, int variant, const u32* exact_max_tiles = nullptr,
                       const bool* uniform_tiles = nullptr, const double* avg_tiles = nullptr,
                       u64 jobs_per_launch = 0, const Mfma16Stats* m16 = nullptr) {
  ScanShape s;
  const int vr = variant & 0xff, vs = (variant >> 8) & 0xff;
  // adaptive tiles when 256-query tiles still leave enough waves to fill the chip (counting the
  // long-scan splits that will be applied below), else the latency shape
  const u64 long_splits = std::max(1u, max_nt / 4096);
  const u64 waves_r4 = n_jobs * ((max_nq + 255) / 256) * long_splits;
  s.adaptive = waves_r4 >= TARGET_WAVES;
  s.R = 1;
  if (vr == 1) s.adaptive = false;
  if (vr == 8) s.adaptive = true;
  if (vr == 4) { s.adaptive = false; s.R = 4; }
  s.rmax = 4;
  if (s.adaptive) {
    // Full tiles of 256 queries (NB 8: the fewest operand expansions per MFMA) unless the database
    // is RAGGED with small images (fewer than four 256-query tiles on average: euroc_V1's 117-1363
    // corners): waves then differ 40-fold in length and 128-query tiles balance better.  Measured
    // with the matrix-core scan (tools/ragged_probe.py, round 2; ms per batch at 256 / 128 / 64):
    // euroc_V1 sizes 0.672 / 0.620 / 0.671, 164 images x 1200-1500 2.09 / 2.13 / 2.37, 200 x 478
    // uniform 0.580 / 0.584 / 0.631, 96 x 2000 dense 2.52 / 2.57 / 2.72, 2048 pairs of 2000 x 2000
    // 0.739 / 0.763 / 0.831.  (Round 1's rule -- smaller tiles until a launch has 131072 waves --
    // was tuned on the popcount scan, whose waves cost the same per query at every tile size.)
    if (uniform_tiles && avg_tiles && !uniform_tiles[1] && avg_tiles[1] < 4.0) s.rmax = 2;
    const int force = (variant >> 28) & 0x7;
    if (force == 1 || force == 2 || force == 4) s.rmax = force;
  }
  const int ri = s.rmax == 4 ? 1 : s.rmax == 2 ? 2 : 3;  // index into the tile statistics
  const u32 tq = 64 * s.R;
  const u32 tiles = s.adaptive ? (exact_max_tiles ? exact_max_tiles[ri] : adaptive_tiles_bound(max_nq, (u32)s.rmax))
                               : (max_nq + tq - 1) / tq;
  s.tiles = std::max(1u, tiles);
  s.uniform = s.adaptive && uniform_tiles && exact_max_tiles && uniform_tiles[ri];
  s.wgs_per_job = std::max(1u, (tiles + 3) / 4);
  const u64 waves = std::max<u64>(1, n_jobs * tiles);
  // (a) latency regime: reach TARGET_WAVES with at least 64 train items per split;
  // (b) long scans (1M x 1M): many short waves balance better than few long ones
  //     (profiles/r01/nn2_sweep.txt: 1 -> 24 splits = 0.80 -> 0.95 of the roofline);
  // (c) a split scans <= FKEY_MAX_SET train items (step tags / float keys).
  u32 sa = 1, sb = 1;
  if (waves < TARGET_WAVES) sa = std::min((u32)((TARGET_WAVES + waves - 1) / waves), std::max(1u, max_nt / 64));
  if (waves < LONG_SCAN_WAVES) sb = std::min((u32)((LONG_SCAN_WAVES + waves - 1) / waves), std::max(1u, max_nt / 4096));
  const u32 sc = (max_nt + FKEY_MAX_SET - 1) / FKEY_MAX_SET;
  s.splits = std::min(std::max(std::max(sa, sb), std::max(sc, 1u)), 256u);
  if (vs) s.splits = (u32)vs;
  const u32 per = (max_nt + s.splits - 1) / s.splits;
  s.fkey = per <= FKEY_MAX_SET;
  s.mfma = s.fkey && !(variant & KNOB_POPCOUNT) && (s.adaptive || (s.R == 1 && !(variant & KNOB_POPCOUNT_LATENCY)));
  // Runs of consecutive jobs per wave (mfma_scan.h, mfma_scan_run): as long as >= 32768 waves are
  // left (16 rounds of the 2048 resident ones: the tail of a launch of four-job waves stays small)
  s.jobs_per_wave = 1;
  if (s.adaptive && s.mfma && s.splits == 1 && s.rmax == 4) {  // the 256-query-tile kernel only (mfma_scan.h)
    const u64 w = (jobs_per_launch ? jobs_per_launch : n_jobs) * (u64)s.tiles;
    while (s.jobs_per_wave < 4 && w / (2 * s.jobs_per_wave) >= 32768) s.jobs_per_wave *= 2;
    const int fj = (variant >> 20) & 0xf;  // knob bits 20-23: forced
    if (fj) s.jobs_per_wave = (u32)fj;
  }
  // the 16x16x128 kernel instead of the 256-query-tile 32x32x64 one (mfma16_scan.h; any number of train splits, step
  // tags of 10 bits as there)
  if (s.adaptive && s.mfma && s.rmax == 4) {
    Mfma16Stats st;
    if (m16) {
      st = *m16;
    } else {
      Mfma16Cost cost;
      cost.add(max_nq, per);
      st = cost.stats();
      st.uniform = false;
    }
    s.mfma16 = (st.prefer || (variant & KNOB_MFMA16_ON)) && !(variant & KNOB_MFMA16_OFF) && st.max_tiles > 0;
    if (s.mfma16) {
      s.tiles = st.max_tiles;
      s.uniform = st.uniform;
      // (no runs of jobs in this kernel: built and measured -- tools/history/mfma16_scan_with_runs_of_jobs.h: the loop
      // costs 100 spilled registers around every job; runs of four then gain 4 % over runs of one of the SAME code, which
      // is 6 % slower than the kernel without the loop: 14.87 against 15.5 T relative to the 32x32 kernel's 14.3 / 14.55)
      s.jobs_per_wave = 1;
    }
  }
  return s;
}

int launch_scan(vsm_ctx* c, const void* qbase, const void* tbase, const Job* d_jobs, u64 n_jobs,
                const ScanShape& sh, u32 split_stride, uint2* d_scratch, u64 comps,
                const u32* d_qlist = nullptr, u32 d_lim = 256,
                AcceptMask acc = AcceptMask{nullptr, 0, nullptr},
                bool timing = true) {  // false: never create events (vsm_match_pair_noalloc)
  const int kind = d_qlist ? 1 : 0;
  if (n_jobs == 0) return VSM_OK;
  u64 blocks = n_jobs * sh.wgs_per_job;
  dim3 block(256);
  const bool m16 = sh.mfma16 && !d_qlist;
  if (sh.adaptive) {
    const u64 groups = sh.mfma ? (n_jobs + sh.jobs_per_wave - 1) / sh.jobs_per_wave : n_jobs;  // mfma_scan_run
    blocks = (sh.uniform ? (groups + 7) / 8 * 8 : groups) * sh.tiles;
    block = dim3(64);
  }
  if (blocks > 0x7fffffffull) return fail(VSM_ERR_INVALID, "scan grid too large (%llu blocks)", (unsigned long long)blocks);
  const dim3 grid((u32)blocks, sh.splits);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool timed = c->profiling && timing;
  if (timed) {
    VSM_TRY(ev_pair(c, &e0, &e1));
    c->ev_kind.push_back(kind);
    HIP_TRY(hipEventRecord(e0, c->stream));
  }
  const u32x4* q = (const u32x4*)qbase;
  const u32x4* t = (const u32x4*)tbase;
  const u32 grouped = sh.uniform ? sh.tiles : 0u;
  // popcount scans: float keys with the skip test (FILL 8) unless a split scans more than 32768
  // rows or the knob asks for integer keys (FILL 2)
  const bool fill8 = sh.fkey && !(c->variant & KNOB_INT_KEYS);
  // Operand codes (mfma_scan.h): the shift-free zero-rich train codes for the 128- and 64-query tiles (16 instead of
  // 28 expansion instructions per 32 train rows, spread over 4 or 2 query blocks: +2 % and +5.7 % on the same box,
  // profiles/r03/scan_codes_and_shapes.txt), the +-0.5 codes for the 256-query tiles (no gain there: the expansion is
  // spread over 8 blocks, and the loop over a run of jobs leaves no register for the queries' constants)
  const bool flip = (c->variant & KNOB_FLIP_CODES) != 0;
#define VSM_LAUNCH_MC(NN, WW, CC)                                                                                      \
  hipLaunchKernelGGL((mfma_scan_kernel<NN, WW, CC>), grid, block, 0, c->stream, q, t, d_jobs, (u32)n_jobs, sh.splits, \
                     split_stride, d_scratch, d_qlist, grouped, d_lim, acc, sh.adaptive ? sh.jobs_per_wave : 1u)
#define VSM_LAUNCH_M(NN, WW)                                  \
  do {                                                        \
    if ((NN == 8) == flip) VSM_LAUNCH_MC(NN, WW, 1);          \
    else VSM_LAUNCH_MC(NN, WW, 0);                            \
  } while (0)
#define VSM_LAUNCH_A(FF, RR)                                                                                        \
  hipLaunchKernelGGL((scan_kernel_adaptive<FF, RR>), grid, block, 0, c->stream, q, t, d_jobs, (u32)n_jobs, sh.splits, \
                     split_stride, d_scratch, d_qlist, grouped)
#define VSM_LAUNCH(RR, FF)                                                                                     \
  hipLaunchKernelGGL((scan_kernel<RR, FF>), grid, block, 0, c->stream, q, t, d_jobs, (u32)n_jobs, sh.splits, \
                     split_stride, d_scratch, d_qlist)
  if (m16) {  // FP4 matrix-core scan on the 16x16x128 shape, 224 queries per wave
    hipLaunchKernelGGL(mfma16_scan_kernel, grid, block, 0, c->stream, q, t, d_jobs, (u32)n_jobs, sh.splits, split_stride, d_scratch, grouped,
                       d_lim, acc);
  } else if (sh.adaptive && sh.mfma) {  // FP4 matrix-core scan on the 32x32x64 shape, 32 * NB queries per wave
    if (sh.rmax == 4) VSM_LAUNCH_M(8, 1);
    else if (sh.rmax == 2) VSM_LAUNCH_M(4, 1);
    else VSM_LAUNCH_M(2, 1);
  } else if (sh.adaptive) {
    if (sh.rmax == 4) { if (fill8) VSM_LAUNCH_A(8, 4); else VSM_LAUNCH_A(2, 4); }
    else if (sh.rmax == 2) { if (fill8) VSM_LAUNCH_A(8, 2); else VSM_LAUNCH_A(2, 2); }
    else { if (fill8) VSM_LAUNCH_A(8, 1); else VSM_LAUNCH_A(2, 1); }
  } else if (sh.mfma) {  // 4-wave workgroups: one stereo pair, the pruned backward pass
    if (sh.R == 2) VSM_LAUNCH_M(4, 4);  // the pruned backward pass: 128-query tiles
    else VSM_LAUNCH_M(2, 4);           // one stereo pair: 64-query tiles
  } else if (sh.R == 4) {
    if (fill8) VSM_LAUNCH(4, 8); else VSM_LAUNCH(4, 2);
  } else {
    if (fill8) VSM_LAUNCH(1, 8); else VSM_LAUNCH(1, 2);
  }
#undef VSM_LAUNCH
#undef VSM_LAUNCH_A
#undef VSM_LAUNCH_M
#undef VSM_LAUNCH_MC
  HIP_TRY(hipGetLastError());
  if (timed) HIP_TRY(hipEventRecord(e1, c->stream));
  c->scan_kernel[kind] = m16 ? 3 : sh.mfma ? 2 : 1;
  c->scan_launches[kind]++;
  c->scan_comps[kind] += comps;
  return VSM_OK;
}

u32 round_up(u32 v, u32 m) { return (v + m - 1) / m * m; }

// The batched matcher on device-resident inputs.  d_desc/d_img_off describe
// the database the pairs index; results go to caller-provided device memory.
int match_pairs_device(vsm_ctx* c, const void* d_desc, const u32* d_img_off, u32 n_img, u32 max_n,
                       const uint2* d_pairs, size_t n_pairs,
                       const AcceptParams& prm, u64* d_out_offsets, int2* d_out_pairs,
                       size_t cap_pairs, size_t* out_total) {
  // `misc`: [u64 running total][u64 comparisons, full/forward scans][u64 comparisons,
  //          pruned backward scans][u32 bad-pair flag] ... from byte 128: nominate_kernel's COMPS_SLOTS counters
  constexpr size_t MISC_BYTES = 128 + COMPS_SLOTS * COMPS_SLOT_STRIDE * sizeof(u64);
  VSM_TRY(c->misc.reserve(MISC_BYTES));
  u64* d_running = (u64*)c->misc.p;
  u64* d_comps = d_running + 1;
  u32* d_bad = (u32*)(d_running + 3);
  u64* d_slots = d_running + 16;
  HIP_TRY(hipMemsetAsync(c->misc.p, 0, MISC_BYTES, c->stream));
  if (n_pairs == 0) {
    HIP_TRY(hipMemsetAsync(d_out_offsets, 0, sizeof(u64), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *out_total = 0;
    return VSM_OK;
  }
  const u32 stride = std::max(64u, round_up(max_n, 64));
  const bool own_db = d_desc == c->db;
  const ScanShape sh = choose_shape(2 * (u64)n_pairs, max_n, max_n, c->variant, own_db ? c->max_tiles : nullptr,
                                    own_db ? c->uniform_tiles : nullptr, own_db ? c->avg_tiles : nullptr,
                                    std::min<u64>(n_pairs, 32768),  // forward jobs of one chunk
                                    own_db ? &c->m16 : nullptr);
  // chunk so that scratch (8 B per query per split, both directions) fits the budget
  const u64 per_pair = (u64)2 * stride * sizeof(uint2) * sh.splits;
  u64 chunk = std::max<u64>(1, SCRATCH_BUDGET / per_pair);
  chunk = std::min<u64>(chunk, n_pairs);
  chunk = std::min<u64>(chunk, 0xffffffffull / ((u64)2 * stride) / 2);
  const u64 split_stride = chunk * 2 * stride;
  if (split_stride * sh.splits > 0xffffffffull) return fail(VSM_ERR_INVALID, "scratch index overflow");
  VSM_TRY(c->scratch.reserve(split_stride * sh.splits * sizeof(uint2)));
  VSM_TRY(c->jobs.reserve(chunk * 2 * sizeof(Job)));
  VSM_TRY(c->counts.reserve(chunk * sizeof(u32)));
  // Pruned backward pass (nominate_kernel): only for unsplit scans and images that fit the LDS bitmap.
  const bool prune = sh.splits == 1 && stride <= NOMINATE_MAX_STRIDE && !(c->variant & KNOB_NO_PRUNE);
  if (prune) VSM_TRY(c->qlist.reserve(chunk * stride * sizeof(u32)));
  if (prune) VSM_TRY(c->flist.reserve(chunk * stride * sizeof(uint2)));
  // accept masks (one bit per forward row) and the accept table on the device, for the matrix-core
  // forward scan's epilogue; the popcount scans do not write masks (nominate_kernel then reads every row)
  const bool masks = prune && sh.mfma && sh.adaptive;
  AcceptMask acc{nullptr, 0, nullptr};
  if (masks) {
    VSM_TRY(c->accmask.reserve(chunk * (stride / 32) * sizeof(u32)));
    VSM_TRY(c->lut.reserve(sizeof prm.min_second));
    HIP_TRY(hipMemcpyAsync(c->lut.p, prm.min_second, sizeof prm.min_second, hipMemcpyHostToDevice, c->stream));
    acc = AcceptMask{(const uint16_t*)c->lut.p, prm.threshold, (u32*)c->accmask.p};
  }
  // Backward jobs of the pruned pass scan ragged nominated lists whose lengths only the device
  // knows: fixed tiles in 4-wave workgroups -- 128 queries per wave on the matrix cores (NB 4: real-like
  // batch 1.98 -> 1.84 ms, dense 0.82 -> 0.75, sparse 0.20 -> 0.18 against 64-query tiles, which the
  // popcount scan keeps; 256-query tiles: 2.18 / 0.74 / 0.25).
  // Adaptive tiles in one-wave workgroups were measured here twice and do not pay: with the popcount
  // scan a sparse batch went 0.45 -> 1.18 ms per launch and a dense one 4.4 -> 5.4 ms; with the
  // matrix-core scan (256-query tiles) sparse 0.137 -> 0.153 ms, dense 1.59 -> 1.62 ms.  One looping
  // workgroup per job instead of eight mostly idle ones: sparse 0.144 -> 0.172 ms, dense 1.50 -> 1.36 ms.
  ScanShape shb = sh;
  shb.adaptive = false;
  shb.R = 1;
  shb.wgs_per_job = std::max(1u, ((max_n + 63) / 64 + 3) / 4);
  shb.mfma = sh.fkey && !(c->variant & (KNOB_POPCOUNT | KNOB_POPCOUNT_BACKWARD));
  if (shb.mfma) {  // matrix cores: 128-query tiles (NB 4, remainders of 64 and fewer), four per workgroup
    shb.R = 2;
    shb.wgs_per_job = std::max(1u, (adaptive_tiles_bound(max_n, 2) + 3) / 4);
  }
  Job* d_jobs = (Job*)c->jobs.p;
  uint2* d_scratch = (uint2*)c->scratch.p;
  u32* d_counts = (u32*)c->counts.p;
  const u32x4* desc = (const u32x4*)d_desc;
  const u32 d_lim = accept_limit(prm);
  const uint8_t* gmin = own_db ? (const uint8_t*)c->gmin.p : nullptr;  // the table belongs to the context's database

  for (u64 first = 0; first < n_pairs; first += chunk) {
    const u32 n = (u32)std::min<u64>(chunk, n_pairs - first);
    hipLaunchKernelGGL(plan_pairs_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream,
                       d_pairs + first, n, d_img_off, n_img, stride, d_jobs, d_bad, d_comps,
                       prune ? 1u : 2u);
    HIP_TRY(hipGetLastError());
    if (prune) {
      VSM_TRY(launch_scan(c, d_desc, d_desc, d_jobs, n, sh, (u32)split_stride, d_scratch, 0, nullptr, d_lim, acc));
      hipLaunchKernelGGL(nominate_kernel, dim3(n), dim3(256), 0, c->stream, d_jobs, n, d_scratch, prm,
                         stride, (u32*)c->qlist.p, (uint2*)c->flist.p, d_slots, (const u32*)acc.mask);
      HIP_TRY(hipGetLastError());
      VSM_TRY(launch_scan(c, d_desc, d_desc, d_jobs + n, n, shb, (u32)split_stride, d_scratch, 0,
                          (const u32*)c->qlist.p, d_lim));
      hipLaunchKernelGGL(count_list_kernel, dim3(n), dim3(256), 0, c->stream, d_jobs, n, d_scratch,
                         (const uint2*)c->flist.p, stride, prm, d_counts, desc,
                         (sh.mfma ? 1u : 0u) | (shb.mfma ? 2u : 0u), gmin);
    } else {
      VSM_TRY(launch_scan(c, d_desc, d_desc, d_jobs, 2 * (u64)n, sh, (u32)split_stride, d_scratch, 0, nullptr, d_lim));
      if (sh.splits > 1 || sh.mfma) {
        const u32 n_entries = n * 2 * stride;
        hipLaunchKernelGGL(merge_refine_kernel, dim3((n_entries + 255) / 256), dim3(256), 0, c->stream,
                           (const Job*)d_jobs, n, stride, d_scratch, sh.splits, (u32)split_stride, desc, desc, prm,
                           sh.mfma ? 1u : 0u);
        HIP_TRY(hipGetLastError());
      }
      hipLaunchKernelGGL(count_kernel, dim3(n), dim3(256), 0, c->stream, d_jobs, d_scratch,
                         1u, (u32)split_stride, prm, d_counts);
    }
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(scan_counts_kernel, dim3(1), dim3(1024), 0, c->stream, d_counts, n,
                       d_out_offsets, first, d_running, prune ? d_slots : nullptr, d_comps + 1);
    HIP_TRY(hipGetLastError());
    if (prune)
      hipLaunchKernelGGL(fill_list_kernel, dim3(n), dim3(256), 0, c->stream, d_jobs, d_scratch, d_out_offsets, first,
                         d_out_pairs, (u64)cap_pairs);
    else
      hipLaunchKernelGGL(fill_kernel, dim3(n), dim3(256), 0, c->stream, d_jobs, d_scratch, 1u,
                         (u32)split_stride, prm, d_out_offsets, first, d_out_pairs, (u64)cap_pairs);
    HIP_TRY(hipGetLastError());
  }
  struct {
    u64 total, comps_fwd, comps_bwd;
    u32 bad;
    u32 pad;
  } h;
  HIP_TRY(hipMemcpyAsync(&h, c->misc.p, 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->profiling) VSM_TRY(stats_collect(c));
  c->scan_comps[0] = h.comps_fwd;
  c->scan_comps[1] = h.comps_bwd;
  *out_total = (size_t)h.total;
  if (h.bad) return fail(VSM_ERR_INVALID, "a pair names an image index >= %u", n_img);
  if (h.total > cap_pairs)
    return fail(VSM_ERR_CAPACITY, "%llu matches need more than cap_pairs=%zu", (unsigned long long)h.total, cap_pairs);
  return VSM_OK;
}

int check_ctx(vsm_ctx* c) {
  if (!c) return fail(VSM_ERR_INVALID, "null context");
  return VSM_OK;
}

int check_offsets(const uint32_t* offsets, size_t n_img) {
  if (!offsets) return fail(VSM_ERR_INVALID, "null offsets");
  if (n_img >= 0xffffffffull) return fail(VSM_ERR_INVALID, "too many images");
  for (size_t i = 0; i < n_img; i++) {
    if (offsets[i + 1] < offsets[i]) return fail(VSM_ERR_INVALID, "offsets not ascending at image %zu", i);
    if (offsets[i + 1] - offsets[i] > VSM_MAX_SET)
      return fail(VSM_ERR_INVALID, "image %zu with %u descriptors exceeds VSM_MAX_SET", i, offsets[i + 1] - offsets[i]);
  }
  return VSM_OK;
}

int set_db(vsm_ctx* c, const void* d_desc, const uint32_t* offsets, size_t n_img) {
  VSM_TRY(check_offsets(offsets, n_img));
  u32 max_n = 0, max_tiles[4] = {0, 0, 0, 0}, min_tiles[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  double sum_tiles[4] = {0, 0, 0, 0};
  Mfma16Cost cost16;
  for (size_t i = 0; i < n_img; i++) {
    max_n = std::max(max_n, offsets[i + 1] - offsets[i]);
    cost16.add(offsets[i + 1] - offsets[i]);
    for (int ri = 0; ri < 4; ri++) {
      const u32 tl = adaptive_tiles(offsets[i + 1] - offsets[i], 8u >> ri);
      max_tiles[ri] = std::max(max_tiles[ri], tl);
      min_tiles[ri] = std::min(min_tiles[ri], tl);
      sum_tiles[ri] += tl;
    }
  }
  if (n_img && !d_desc && offsets[n_img] != 0) return fail(VSM_ERR_INVALID, "null descriptor pointer");
  VSM_TRY(c->d_img_off.reserve((n_img + 1) * sizeof(u32)));
  HIP_TRY(hipMemcpyAsync(c->d_img_off.p, offsets, (n_img + 1) * sizeof(u32), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->img_off.assign(offsets, offsets + n_img + 1);
  c->max_n = max_n;
  for (int ri = 0; ri < 4; ri++) {
    c->max_tiles[ri] = max_tiles[ri];
    c->uniform_tiles[ri] = n_img > 0 && min_tiles[ri] == max_tiles[ri];
    c->avg_tiles[ri] = n_img ? sum_tiles[ri] / (double)n_img : 0.0;
  }
  c->m16 = cost16.stats();
  c->db = d_desc;
  // the group table of the consumer kernels (vsmatch_kernels.h, group_min_kernel): reads every
  // descriptor once, now
  const size_t total = n_img ? offsets[n_img] : 0;
  VSM_TRY(c->gmin.reserve(std::max<size_t>(total, 1)));
  if (total) {
    hipLaunchKernelGGL(group_min_kernel, dim3((u32)((total + 255) / 256)), dim3(256), 0, c->stream, (const u32x4*)d_desc,
                       (const u32*)c->d_img_off.p, (u32)n_img, (u32)total, (uint8_t*)c->gmin.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  return VSM_OK;
}

}  // namespace

extern "C" {

int vsm_version(void) { return 100; }

const char* vsm_last_error(void) { return g_err; }

int vsm_create(int device, vsm_ctx** out) {
  if (!out) return fail(VSM_ERR_INVALID, "null out pointer");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(VSM_ERR_HIP, "no HIP device available (%s); libvsmatch has no CPU fallback",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(VSM_ERR_INVALID, "device %d out of range [0, %d)", device, n);
  DeviceGuard g(device);
  if (!g.ok) return fail(VSM_ERR_HIP, "hipSetDevice(%d) failed", device);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(VSM_ERR_HIP, "device %d is %s; libvsmatch is built for gfx950 only", device, prop.gcnArchName);
  vsm_ctx* c = new (std::nothrow) vsm_ctx;
  if (!c) return fail(VSM_ERR_NOMEM, "out of host memory");
  c->device = device;
  c->pin_out.flags = hipHostMallocMapped | hipHostMallocCoherent;
  c->pin_in.flags = hipHostMallocMapped | hipHostMallocCoherent;
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return fail(VSM_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
  }
  // device address of the static scratch of vsm_match_pair_noalloc ON THIS DEVICE (current here)
  e = hipGetSymbolAddress(&c->noalloc, HIP_SYMBOL(g_noalloc_scratch));
  if (e != hipSuccess || !c->noalloc) {
    (void)hipStreamDestroy(c->stream);
    delete c;
    return fail(VSM_ERR_HIP, "hipGetSymbolAddress(g_noalloc_scratch) failed on device %d: %s", device, hipGetErrorString(e));
  }
  *out = c;
  return VSM_OK;
}

void vsm_destroy(vsm_ctx* c) {
  if (!c) return;
  {
    DeviceGuard g(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    for (DevBuf* b : {&c->db_owned, &c->d_img_off, &c->stage, &c->scratch, &c->jobs, &c->counts, &c->misc, &c->qlist, &c->flist, &c->accmask, &c->lut,
                      &c->w_pairs, &c->w_offsets, &c->w_out, &c->image, &c->gftt, &c->voc, &c->bowdb, &c->bowdb_work, &c->gmin})
      b->release();
    c->pin.release();
    c->pin_image.release();
    c->pin_out.release();
    c->pin_in.release();
    if (c->stream) (void)hipStreamDestroy(c->stream);
  }
  delete c;
}

void* vsm_stream(vsm_ctx* c) { return c ? (void*)c->stream : nullptr; }

int vsm_set_profiling(vsm_ctx* c, int enabled) {
  VSM_TRY(check_ctx(c));
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->profiling = enabled != 0;
  return VSM_OK;
}

int vsm_accept_limit(int threshold, double dist_2_best) {
  AcceptParams prm;
  make_accept(threshold, dist_2_best, &prm);
  return (int)accept_limit(prm);
}

int vsm_set_variant(vsm_ctx* c, int variant) {
  VSM_TRY(check_ctx(c));
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->variant = variant;
  return VSM_OK;
}

int vsm_last_scan_stats(vsm_ctx* c, double* ms, uint64_t* launches, uint64_t* comparisons) {
  VSM_TRY(check_ctx(c));
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  if (ms) *ms = c->scan_ms[0] + c->scan_ms[1];
  if (launches) *launches = c->scan_launches[0] + c->scan_launches[1];
  if (comparisons) *comparisons = c->scan_comps[0] + c->scan_comps[1];
  return VSM_OK;
}

int vsm_last_scan_stats_kind(vsm_ctx* c, int kind, double* ms, uint64_t* launches, uint64_t* comparisons) {
  VSM_TRY(check_ctx(c));
  if (kind < 0 || kind > 1) return fail(VSM_ERR_INVALID, "kind must be 0 (full/forward) or 1 (pruned backward)");
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  if (ms) *ms = c->scan_ms[kind];
  if (launches) *launches = c->scan_launches[kind];
  if (comparisons) *comparisons = c->scan_comps[kind];
  return VSM_OK;
}

int vsm_last_scan_kernel(vsm_ctx* c, int kind) {
  if (!c || kind < 0 || kind > 1) return 0;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  return c->scan_kernel[kind];
}

int vsm_db_upload(vsm_ctx* c, const uint64_t* desc, const uint32_t* offsets, size_t n_img) {
  VSM_TRY(check_ctx(c));
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  VSM_TRY(check_offsets(offsets, n_img));  // before the old database is touched
  const size_t total = offsets[n_img];
  if (total && !desc) return fail(VSM_ERR_INVALID, "null descriptor pointer");
  // From here on the old database is gone (reserve may free it, the copy overwrites it): forget it,
  // so that a failure below leaves "no database", not pointers into freed or mismatched memory.
  c->db = nullptr;
  c->img_off.clear();
  VSM_TRY(c->db_owned.reserve(std::max<size_t>(total * DESC_BYTES, 256)));
  if (total) HIP_TRY(hipMemcpyAsync(c->db_owned.p, desc, total * DESC_BYTES, hipMemcpyHostToDevice, c->stream));
  return set_db(c, c->db_owned.p, offsets, n_img);
}

int vsm_db_attach(vsm_ctx* c, const void* d_desc, const uint32_t* offsets, size_t n_img) {
  VSM_TRY(check_ctx(c));
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  return set_db(c, d_desc, offsets, n_img);
}

int vsm_match_pairs_device(vsm_ctx* c, const void* d_pairs, size_t n_pairs, int threshold,
                           double dist_2_best, void* d_out_offsets, void* d_out_pairs,
                           size_t cap_pairs, size_t* out_total) {
  VSM_TRY(check_ctx(c));
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  if (!out_total || !d_out_offsets) return fail(VSM_ERR_INVALID, "null output pointer");
  if (n_pairs && !d_pairs) return fail(VSM_ERR_INVALID, "null pairs pointer");
  if (cap_pairs && !d_out_pairs) return fail(VSM_ERR_INVALID, "null out_pairs with non-zero capacity");
  if (c->img_off.empty()) return fail(VSM_ERR_INVALID, "no database: call vsm_db_upload/vsm_db_attach first");
  AcceptParams prm;
  make_accept(threshold, dist_2_best, &prm);
  stats_reset(c);
  return match_pairs_device(c, c->db, (const u32*)c->d_img_off.p, (u32)(c->img_off.size() - 1), c->max_n,
                            (const uint2*)d_pairs, n_pairs, prm, (u64*)d_out_offsets,
                            (int2*)d_out_pairs, cap_pairs, out_total);
}

int vsm_match_pairs(vsm_ctx* c, const uint32_t* pairs, size_t n_pairs, int threshold,
                    double dist_2_best, uint64_t* out_offsets, int32_t* out_pairs, size_t cap_pairs,
                    size_t* out_total) {
  VSM_TRY(check_ctx(c));
  if (!out_total || !out_offsets) return fail(VSM_ERR_INVALID, "null output pointer");
  if (n_pairs && !pairs) return fail(VSM_ERR_INVALID, "null pairs pointer");
  if (cap_pairs && !out_pairs) return fail(VSM_ERR_INVALID, "null out_pairs with non-zero capacity");
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  VSM_TRY(c->w_pairs.reserve(std::max<size_t>(n_pairs * 8, 8)));
  VSM_TRY(c->w_offsets.reserve((n_pairs + 1) * 8));
  VSM_TRY(c->w_out.reserve(std::max<size_t>(cap_pairs * 8, 8)));
  if (n_pairs) HIP_TRY(hipMemcpyAsync(c->w_pairs.p, pairs, n_pairs * 8, hipMemcpyHostToDevice, c->stream));
  int rc = vsm_match_pairs_device(c, c->w_pairs.p, n_pairs, threshold, dist_2_best, c->w_offsets.p,
                                  c->w_out.p, cap_pairs, out_total);
  if (rc != VSM_OK && rc != VSM_ERR_CAPACITY) return rc;
  HIP_TRY(hipMemcpyAsync(out_offsets, c->w_offsets.p, (n_pairs + 1) * 8, hipMemcpyDeviceToHost, c->stream));
  const size_t n_copy = std::min(*out_total, cap_pairs);
  if (n_copy) HIP_TRY(hipMemcpyAsync(out_pairs, c->w_out.p, n_copy * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return rc;
}

int vsm_db_upload_multi(vsm_ctx* const* ctxs, size_t n_ctx, const uint64_t* desc, const uint32_t* offsets, size_t n_img) {
  if (!ctxs || n_ctx == 0) return fail(VSM_ERR_INVALID, "no contexts");
  for (size_t k = 0; k < n_ctx; k++) VSM_TRY(check_ctx(ctxs[k]));
  vsm_ctx* c0 = ctxs[0];
  // context 0 stays locked to the end: the other contexts copy out of ITS buffer, which a concurrent call on
  // context 0 could otherwise free or regrow in between (the mutex is recursive: vsm_db_upload takes it again)
  std::lock_guard<std::recursive_mutex> lk0(c0->mu);
  VSM_TRY(vsm_db_upload(c0, desc, offsets, n_img));  // validates everything, host -> first context (synchronised)
  const size_t bytes = (size_t)offsets[n_img] * DESC_BYTES;
  for (size_t k = 1; k < n_ctx; k++) {
    vsm_ctx* c = ctxs[k];
    if (c == c0) continue;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    DeviceGuard g(c->device);
    c->db = nullptr;
    c->img_off.clear();
    VSM_TRY(c->db_owned.reserve(std::max<size_t>(bytes, 256)));
    if (bytes) {
      // device to device on the receiving context's stream: xGMI between the GPUs of a node (the runtime
      // stages through the host where two devices are not peers); from the host again if that fails.
      // set_db synchronises the stream before it reads the copy.
      hipError_t e = c->device == c0->device
                         ? hipMemcpyAsync(c->db_owned.p, c0->db_owned.p, bytes, hipMemcpyDeviceToDevice, c->stream)
                         : hipMemcpyPeerAsync(c->db_owned.p, c->device, c0->db_owned.p, c0->device, bytes, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        HIP_TRY(hipMemcpyAsync(c->db_owned.p, desc, bytes, hipMemcpyHostToDevice, c->stream));
      }
    }
    VSM_TRY(set_db(c, c->db_owned.p, offsets, n_img));
  }
  return VSM_OK;
}

int vsm_match_pairs_multi(vsm_ctx* const* ctxs, size_t n_ctx, const uint32_t* pairs, size_t n_pairs, int threshold,
                          double dist_2_best, uint64_t* out_offsets, int32_t* out_pairs, size_t cap_pairs,
                          size_t* out_total) {
  if (!ctxs || n_ctx == 0) return fail(VSM_ERR_INVALID, "no contexts");
  if (!out_total || !out_offsets) return fail(VSM_ERR_INVALID, "null output pointer");
  if (n_pairs && !pairs) return fail(VSM_ERR_INVALID, "null pairs pointer");
  if (cap_pairs && !out_pairs) return fail(VSM_ERR_INVALID, "null out_pairs with non-zero capacity");
  for (size_t k = 0; k < n_ctx; k++) {
    VSM_TRY(check_ctx(ctxs[k]));
    for (size_t l = 0; l < k; l++)
      if (ctxs[l] == ctxs[k]) return fail(VSM_ERR_INVALID, "context %zu is listed twice", k);
    if (ctxs[k]->img_off.empty()) return fail(VSM_ERR_INVALID, "context %zu has no database: call vsm_db_upload_multi first", k);
  }
  constexpr size_t BLOCK = 4096;  // consecutive pairs share an image: a block keeps it hot in the L2s
  const size_t n_blocks = (n_pairs + BLOCK - 1) / BLOCK;
  struct Part {
    std::vector<uint32_t> pairs;
    std::vector<u64> off;  // local offsets, n_local + 1
    size_t total = 0, cap = 0;  // matches of this share; capacity its device buffer was filled up to
    int rc = VSM_OK;
    std::string err;
  };
  std::vector<Part> parts(n_ctx);
  std::mutex bm;
  std::condition_variable bcv;
  size_t arrived = 0;
  bool released = false;
  u64 grand_total = 0;

  auto worker = [&](size_t k) {
    vsm_ctx* c = ctxs[k];
    Part& pt = parts[k];
    std::lock_guard<std::recursive_mutex> lk(c->mu);  // held to the end: the lists live in the context's buffers
    DeviceGuard g(c->device);
    // No exception leaves this function (it runs on a std::thread under an extern "C" entry point): a failed
    // allocation becomes the part's error code, and the barrier below is reached either way.
    auto guarded = [&](auto&& f) -> int {
      try {
        return f();
      } catch (const std::bad_alloc&) {
        return fail(VSM_ERR_NOMEM, "out of host memory");
      } catch (const std::exception& e) {
        return fail(VSM_ERR_INVALID, "%s", e.what());
      }
    };
    size_t n_local = 0;
    auto run = [&]() -> int {
      // phase 1: this context's blocks k, k + n_ctx, ... as one batch on its GPU, lists stay in its HBM
      for (size_t b = k; b < n_blocks; b += n_ctx) {
        const size_t p0 = b * BLOCK, p1 = std::min(n_pairs, p0 + BLOCK);
        pt.pairs.insert(pt.pairs.end(), pairs + 2 * p0, pairs + 2 * p1);
      }
      n_local = pt.pairs.size() / 2;
      const size_t n_img = c->img_off.size() - 1;
      size_t bound = 0;  // the cross-check emits at most min(n1, n2) pairs per image pair
      for (size_t i = 0; i < n_local; i++) {
        const uint32_t a = pt.pairs[2 * i], b = pt.pairs[2 * i + 1];
        if (a < n_img && b < n_img)
          bound += std::min(c->img_off[a + 1] - c->img_off[a], c->img_off[b + 1] - c->img_off[b]);
      }
      const size_t cap = pt.cap = std::min(cap_pairs, bound);
      VSM_TRY(c->w_pairs.reserve(std::max<size_t>(n_local * 8, 8)));
      VSM_TRY(c->w_offsets.reserve((n_local + 1) * 8));
      VSM_TRY(c->w_out.reserve(std::max<size_t>(cap * 8, 8)));
      if (n_local) HIP_TRY(hipMemcpyAsync(c->w_pairs.p, pt.pairs.data(), n_local * 8, hipMemcpyHostToDevice, c->stream));
      const int rc = vsm_match_pairs_device(c, c->w_pairs.p, n_local, threshold, dist_2_best, c->w_offsets.p, c->w_out.p,
                                            cap, &pt.total);
      if (rc != VSM_OK && rc != VSM_ERR_CAPACITY) return rc;
      pt.off.resize(n_local + 1);
      HIP_TRY(hipMemcpyAsync(pt.off.data(), c->w_offsets.p, (n_local + 1) * 8, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      return rc;
    };
    pt.rc = guarded(run);
    if (pt.rc != VSM_OK) pt.err = g_err;
    // barrier; the last context to arrive lays out the merged list in canonical pair order
    {
      std::unique_lock<std::mutex> bl(bm);
      if (++arrived == n_ctx) {
        bool ok = true;
        for (const Part& q : parts) ok = ok && (q.rc == VSM_OK || q.rc == VSM_ERR_CAPACITY);
        if (ok) {
          u64 run_total = 0;
          for (size_t b = 0; b < n_blocks; b++) {
            const Part& q = parts[b % n_ctx];
            const size_t p0 = b * BLOCK, p1 = std::min(n_pairs, p0 + BLOCK), l0 = (b / n_ctx) * BLOCK;
            for (size_t p = p0; p < p1; p++) {
              out_offsets[p] = run_total;
              run_total += q.off[l0 + (p - p0) + 1] - q.off[l0 + (p - p0)];
            }
          }
          out_offsets[n_pairs] = run_total;
          grand_total = run_total;
        }
        released = true;
        bcv.notify_all();
      } else {
        bcv.wait(bl, [&] { return released; });
      }
    }
    for (const Part& q : parts)
      if (q.rc != VSM_OK && q.rc != VSM_ERR_CAPACITY) return;
    // phase 2: every block of this context is contiguous here and in the merged list
    auto copy_out = [&]() -> int {
      for (size_t b = k; b < n_blocks; b += n_ctx) {
        const size_t p0 = b * BLOCK, p1 = std::min(n_pairs, p0 + BLOCK), l0 = (b / n_ctx) * BLOCK;
        const u64 src0 = pt.off[l0], n = pt.off[l0 + (p1 - p0)] - src0, dst0 = out_offsets[p0];
        // what the context's buffer holds (it was filled up to its own capacity) and what the caller's takes
        const u64 have = src0 < pt.cap ? std::min<u64>(n, pt.cap - src0) : 0;
        const u64 take = dst0 < cap_pairs ? std::min<u64>(have, cap_pairs - dst0) : 0;
        if (take)
          HIP_TRY(hipMemcpyAsync(out_pairs + 2 * dst0, (const char*)c->w_out.p + src0 * 8, take * 8, hipMemcpyDeviceToHost, c->stream));
      }
      HIP_TRY(hipStreamSynchronize(c->stream));
      return VSM_OK;
    };
    const int rc2 = guarded(copy_out);
    if (rc2 != VSM_OK) {
      pt.rc = rc2;
      pt.err = g_err;
    }
  };

  if (n_ctx == 1) {
    worker(0);
  } else {
    std::vector<std::thread> th;
    try {
      th.reserve(n_ctx);
      for (size_t k = 0; k < n_ctx; k++) th.emplace_back(worker, k);
    } catch (const std::exception& e) {
      // a thread could not be started: the ones already running wait at the barrier for n_ctx arrivals --
      // the missing workers arrive here, as failed parts, so that everybody leaves
      std::unique_lock<std::mutex> bl(bm);
      for (size_t k = th.size(); k < n_ctx; k++) {
        parts[k].rc = VSM_ERR_NOMEM;
        parts[k].err = std::string("could not start a host thread: ") + e.what();
      }
      arrived += n_ctx - th.size();
      if (arrived == n_ctx) {
        released = true;
        bcv.notify_all();
      }
    }
    for (std::thread& t : th) t.join();
  }
  for (const Part& q : parts)
    if (q.rc != VSM_OK && q.rc != VSM_ERR_CAPACITY) return fail(q.rc, "%s", q.err.c_str());
  *out_total = (size_t)grand_total;
  if (grand_total > cap_pairs)
    return fail(VSM_ERR_CAPACITY, "%llu matches need more than cap_pairs=%zu", (unsigned long long)grand_total, cap_pairs);
  return VSM_OK;
}

}  // extern "C"
// Waits for the flag word a kernel raises in mapped host memory after its last write: a spin of at most ~200 us
// (one stereo pair is ~30 us end to end), then the caller falls back to hipStreamSynchronize, which sleeps -- with
// the reference's 16 TBB workers calling at once (src/sfm.cpp:1295) sixteen cores would otherwise spin for as long
// as the GPU queues their work.  true: the flag was seen (acquire).
static bool poll_flag(volatile u64* flag, u64 seq) {
  const auto t0 = std::chrono::steady_clock::now();
  for (u32 spin = 0;; spin++) {
    if (*flag == seq) {
      std::atomic_thread_fence(std::memory_order_acquire);
      return true;
    }
    __builtin_ia32_pause();
    if ((spin & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) return false;
  }
}
extern "C" {
int vsm_match_pair(vsm_ctx* c, const uint64_t* d1, size_t n1, const uint64_t* d2, size_t n2,
                   int threshold, double dist_2_best, int32_t* out_pairs, size_t cap_pairs,
                   size_t* out_count) {
  VSM_TRY(check_ctx(c));
  if (!out_count) return fail(VSM_ERR_INVALID, "null out_count");
  if ((n1 && !d1) || (n2 && !d2)) return fail(VSM_ERR_INVALID, "null descriptor pointer");
  if (n1 > VSM_MAX_SET || n2 > VSM_MAX_SET) return fail(VSM_ERR_INVALID, "set larger than VSM_MAX_SET");
  if (cap_pairs && !out_pairs) return fail(VSM_ERR_INVALID, "null out_pairs with non-zero capacity");
  *out_count = 0;
  if (n1 == 0 || n2 == 0) return VSM_OK;  // keypoints.h:251: matches.clear(), nothing to add
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  stats_reset(c);
  AcceptParams prm;
  make_accept(threshold, dist_2_best, &prm);

  // Latency path (one call per odometry keyframe): everything stays in the compute queue.
  // Up: [d1 | d2 | Job f | Job b] in mapped host memory, copied to HBM by stage_in_kernel (an SDMA
  // copy in front of the kernels cost ~14 us of cross-engine waiting).  Down: the emit kernel writes
  // [offsets (2 x u64) | flag, pad to 64 | pairs] straight into mapped host memory and raises the
  // flag last; the host polls it (a device-to-host copy + stream wait cost ~8 us).  With the 13 us
  // the accept table used to take: 57 -> 28 us per 1500 x 1500 call from C.
  const size_t nmin = std::min(n1, n2);
  const size_t up_bytes = (n1 + n2) * DESC_BYTES + 2 * sizeof(Job);
  const size_t down_bytes = 64 + nmin * 8;
  VSM_TRY(c->pin_in.reserve(up_bytes + 16));
  VSM_TRY(c->pin_out.reserve(down_bytes));
  VSM_TRY(c->stage.reserve(up_bytes + 16));
  const bool mapped = c->pin_out.dev != nullptr;
  if (!mapped) VSM_TRY(c->w_out.reserve(down_bytes));
  // The flag word starts from 0: pin_out is also written by other entry points of this context (the ORB kernels
  // put moments there) and a fresh block may hold another context's sequence numbers -- a stale word equal to
  // `seq` would let the host read the list before the emit kernel has run.  Nothing of this context is in
  // flight here (every entry point that uses pin_out waits for its own work before it returns).
  *(volatile u64*)((char*)c->pin_out.p + 16) = 0;
  char* h = (char*)c->pin_in.p;
  memcpy(h, d1, n1 * DESC_BYTES);
  memcpy(h + n1 * DESC_BYTES, d2, n2 * DESC_BYTES);
  const u32 stride = std::max(64u, round_up((u32)std::max(n1, n2), 64));
  Job jf = {0, (u32)n1, (u32)n1, (u32)n2, 0, NO_QLIST, 0, 0};
  Job jb = {(u32)n1, (u32)n2, 0, (u32)n1, stride, NO_QLIST, 0, 0};
  memcpy(h + (n1 + n2) * DESC_BYTES, &jf, sizeof jf);
  memcpy(h + (n1 + n2) * DESC_BYTES + sizeof jf, &jb, sizeof jb);
  if (c->pin_in.dev) {
    const u32 n16 = (u32)((up_bytes + 15) / 16);
    hipLaunchKernelGGL(stage_in_kernel, dim3((n16 + 255) / 256), dim3(256), 0, c->stream, (const u32x4*)c->pin_in.dev,
                       (u32x4*)c->stage.p, n16);
  } else {
    HIP_TRY(hipMemcpyAsync(c->stage.p, h, up_bytes, hipMemcpyHostToDevice, c->stream));
  }
  const Job* d_jobs = (const Job*)((char*)c->stage.p + (n1 + n2) * DESC_BYTES);

  const ScanShape sh = choose_shape(2, (u32)std::max(n1, n2), (u32)std::max(n1, n2), c->variant);
  const u32 split_stride = 2 * stride;
  VSM_TRY(c->scratch.reserve((size_t)split_stride * sh.splits * sizeof(uint2)));
  VSM_TRY(launch_scan(c, c->stage.p, c->stage.p, d_jobs, 2, sh, split_stride, (uint2*)c->scratch.p,
                      2ull * n1 * n2, nullptr, accept_limit(prm)));
  char* d_down = mapped ? (char*)c->pin_out.dev : (char*)c->w_out.p;
  volatile u64* h_flag = (volatile u64*)((char*)c->pin_out.p + 16);
  const u64 seq = ++c->pair_seq;
  if (sh.splits > 1 || sh.mfma)
    hipLaunchKernelGGL(merge_refine_wide_kernel, dim3((split_stride + 15) / 16), dim3(256), 0, c->stream, d_jobs,
                       stride, (uint2*)c->scratch.p, sh.splits, split_stride, (const u32x4*)c->stage.p,
                       (const u32x4*)c->stage.p, prm, sh.mfma ? 1u : 0u);
  hipLaunchKernelGGL(pair_emit_kernel, dim3(1), dim3(1024), 0, c->stream, d_jobs, (const uint2*)c->scratch.p,
                     split_stride, prm, (u64*)d_down, (int2*)(d_down + 64), (u64)nmin, 1u,
                     mapped ? (u64*)(d_down + 16) : (u64*)nullptr, seq);
  HIP_TRY(hipGetLastError());
  char* hd = (char*)c->pin_out.p;
  const bool seen = mapped && !c->profiling && poll_flag(h_flag, seq);
  if (!seen) {
    if (!mapped) HIP_TRY(hipMemcpyAsync(hd, c->w_out.p, down_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  if (c->profiling) VSM_TRY(stats_collect(c));
  u64 offs[2];
  memcpy(offs, hd, 16);
  const size_t count = (size_t)offs[1];
  *out_count = count;
  if (count > cap_pairs) return fail(VSM_ERR_CAPACITY, "%zu matches need more than cap_pairs=%zu", count, cap_pairs);
  if (count) memcpy(out_pairs, hd + 64, count * 8);
  return VSM_OK;
}

int vsm_match_pair_noalloc(vsm_ctx* c, const void* d_d1, size_t n1, const void* d_d2, size_t n2, int threshold,
                           double dist_2_best, void* d_out_pairs, size_t cap_pairs, void* d_out_count) {
  VSM_TRY(check_ctx(c));
  if (!d_out_count) return fail(VSM_ERR_INVALID, "null d_out_count");
  if ((n1 && !d_d1) || (n2 && !d_d2)) return fail(VSM_ERR_INVALID, "null descriptor pointer");
  if (cap_pairs && !d_out_pairs) return fail(VSM_ERR_INVALID, "null d_out_pairs with non-zero capacity");
  if (n1 > NOALLOC_MAX_SET || n2 > NOALLOC_MAX_SET)
    return fail(VSM_ERR_INVALID, "vsm_match_pair_noalloc takes at most %zu descriptors per side (got %zu x %zu): "
                "its scratch is a fixed device array; use vsm_match_pair", NOALLOC_MAX_SET, n1, n2);
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  if (n1 == 0 || n2 == 0) {  // keypoints.h:251: matches.clear(), nothing to add
    HIP_TRY(hipMemsetAsync(d_out_count, 0, sizeof(u64), c->stream));
    return VSM_OK;
  }
  stats_reset(c);
  AcceptParams prm;
  make_accept(threshold, dist_2_best, &prm);
  const u32 nmax = (u32)std::max(n1, n2);
  const u32 stride = std::max(64u, round_up(nmax, 64));
  const ScanShape sh = choose_shape(2, nmax, nmax, c->variant);
  const u32 split_stride = 2 * stride;
  const size_t need = (size_t)split_stride * sh.splits * sizeof(uint2);
  if (need > NOALLOC_RESULT_BYTES)
    return fail(VSM_ERR_INVALID, "vsm_match_pair_noalloc: %u train splits of %zu x %zu need %zu bytes of scan results, "
                "the fixed scratch holds %zu", sh.splits, n1, n2, need, NOALLOC_RESULT_BYTES);
  // everything below only enqueues kernels on the context's stream: no allocation, no copy, no wait
  Job* d_jobs = (Job*)c->noalloc;
  uint2* d_scratch = (uint2*)((char*)c->noalloc + 256);
  const Job jf = {0, (u32)n1, 0, (u32)n2, 0, NO_QLIST, 0, 0};
  const Job jb = {0, (u32)n2, 0, (u32)n1, stride, NO_QLIST, 0, 0};
  hipLaunchKernelGGL(set_jobs_kernel, dim3(1), dim3(64), 0, c->stream, d_jobs, jf, jb, 2u);
  HIP_TRY(hipGetLastError());
  const u32 d_lim = accept_limit(prm);
  // the two sets are separate buffers: one launch per direction (vsm_match_pair stages both sets
  // into one buffer and scans both directions in one launch)
  VSM_TRY(launch_scan(c, d_d1, d_d2, d_jobs, 1, sh, split_stride, d_scratch, (u64)n1 * n2, nullptr, d_lim,
                      AcceptMask{nullptr, 0, nullptr}, false));
  VSM_TRY(launch_scan(c, d_d2, d_d1, d_jobs + 1, 1, sh, split_stride, d_scratch, (u64)n1 * n2, nullptr, d_lim,
                      AcceptMask{nullptr, 0, nullptr}, false));
  if (sh.splits > 1 || sh.mfma)
    hipLaunchKernelGGL(merge_refine_kernel, dim3((split_stride + 255) / 256), dim3(256), 0, c->stream, (const Job*)d_jobs,
                       1u, stride, d_scratch, sh.splits, split_stride, (const u32x4*)d_d1, (const u32x4*)d_d2, prm,
                       sh.mfma ? 1u : 0u);
  hipLaunchKernelGGL(pair_emit_kernel, dim3(1), dim3(256), 0, c->stream, (const Job*)d_jobs, (const uint2*)d_scratch,
                     split_stride, prm, (u64*)d_out_count, (int2*)d_out_pairs, (u64)cap_pairs, 0u, (u64*)nullptr, (u64)0);
  HIP_TRY(hipGetLastError());
  return VSM_OK;
}

int vsm_nn2_device(vsm_ctx* c, const void* d_query, size_t nq, const void* d_train, size_t nt,
                   void* d_best_idx, void* d_best_dist, void* d_second_dist) {
  VSM_TRY(check_ctx(c));
  if (nq == 0) return VSM_OK;
  if (!d_query || (nt && !d_train) || !d_best_idx || !d_best_dist || !d_second_dist)
    return fail(VSM_ERR_INVALID, "null pointer");
  if (nt > VSM_MAX_SET) return fail(VSM_ERR_INVALID, "train set of %zu exceeds VSM_MAX_SET", nt);
  if (nq > 0xffffff00ull) return fail(VSM_ERR_INVALID, "query set too large");
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  stats_reset(c);
  const ScanShape sh = choose_shape(1, (u32)nq, (u32)nt, c->variant);
  const u32 split_stride = round_up((u32)nq, 64);
  if ((u64)split_stride * sh.splits > 0xffffffffull) return fail(VSM_ERR_INVALID, "scratch index overflow");
  VSM_TRY(c->scratch.reserve((size_t)split_stride * sh.splits * sizeof(uint2)));
  VSM_TRY(c->jobs.reserve(sizeof(Job)));
  // The job goes to the device as a kernel argument: this entry point returns before its work has
  // run, and a host staging buffer could be overwritten by the next call before it was copied.
  const Job j = {0, (u32)nq, 0, (u32)nt, 0, NO_QLIST, 0, 0};
  hipLaunchKernelGGL(set_jobs_kernel, dim3(1), dim3(64), 0, c->stream, (Job*)c->jobs.p, j, j, 1u);
  HIP_TRY(hipGetLastError());
  const void* d_tr = nt ? d_train : d_query;
  VSM_TRY(launch_scan(c, d_query, d_tr, (const Job*)c->jobs.p, 1, sh, split_stride, (uint2*)c->scratch.p,
                      (u64)nq * nt, nullptr, 256));
  hipLaunchKernelGGL(finalize_nn2_kernel, dim3((u32)((nq + 255) / 256)), dim3(256), 0, c->stream,
                     (const uint2*)c->scratch.p, (u32)nq, sh.splits, split_stride, (int*)d_best_idx,
                     (int*)d_best_dist, (int*)d_second_dist, (const u32x4*)d_query, (const u32x4*)d_tr, (u32)nt,
                     sh.mfma ? 1u : 0u);
  HIP_TRY(hipGetLastError());
  if (c->profiling) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    VSM_TRY(stats_collect(c));
  }
  return VSM_OK;
}

int vsm_nn2(vsm_ctx* c, const uint64_t* query, size_t nq, const uint64_t* train, size_t nt,
            int32_t* best_idx, int32_t* best_dist, int32_t* second_dist) {
  VSM_TRY(check_ctx(c));
  if (nq == 0) return VSM_OK;
  if (!query || (nt && !train) || !best_idx || !best_dist || !second_dist)
    return fail(VSM_ERR_INVALID, "null pointer");
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  VSM_TRY(c->stage.reserve((nq + nt) * DESC_BYTES + 256));
  VSM_TRY(c->w_out.reserve(nq * 12));
  char* dq = (char*)c->stage.p;
  char* dt = dq + nq * DESC_BYTES;
  char* dout = (char*)c->w_out.p;
  HIP_TRY(hipMemcpyAsync(dq, query, nq * DESC_BYTES, hipMemcpyHostToDevice, c->stream));
  if (nt) HIP_TRY(hipMemcpyAsync(dt, train, nt * DESC_BYTES, hipMemcpyHostToDevice, c->stream));
  VSM_TRY(vsm_nn2_device(c, dq, nq, dt, nt, dout, dout + nq * 4, dout + nq * 8));
  HIP_TRY(hipMemcpyAsync(best_idx, dout, nq * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(best_dist, dout + nq * 4, nq * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(second_dist, dout + nq * 8, nq * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return VSM_OK;
}

int vsm_match_landmarks(vsm_ctx* c, const uint64_t* kp_desc, const double* kp_xy, size_t nk,
                        const double* proj_xy, size_t np, const uint32_t* obs_off,
                        const uint64_t* obs_desc, double match_max_dist_2d, int threshold,
                        double dist_2_best, int32_t* out_pairs, size_t cap_pairs, size_t* out_count) {
  VSM_TRY(check_ctx(c));
  if (!out_count) return fail(VSM_ERR_INVALID, "null out_count");
  *out_count = 0;
  if (nk && (!kp_desc || !kp_xy)) return fail(VSM_ERR_INVALID, "null keypoint pointer");
  if (np && (!proj_xy || !obs_off)) return fail(VSM_ERR_INVALID, "null projected-point pointer");
  if (nk > VSM_MAX_SET || np > VSM_MAX_SET) return fail(VSM_ERR_INVALID, "set larger than VSM_MAX_SET");
  if (cap_pairs && !out_pairs) return fail(VSM_ERR_INVALID, "null out_pairs with non-zero capacity");
  if (nk == 0 || np == 0) return VSM_OK;  // vo_utils.h:104: md.matches.clear(), nothing passes
  for (size_t p = 0; p < np; p++)
    if (obs_off[p + 1] < obs_off[p]) return fail(VSM_ERR_INVALID, "obs_off not ascending at projected point %zu", p);
  if (obs_off[0] != 0) return fail(VSM_ERR_INVALID, "obs_off[0] must be 0");
  const size_t n_obs = obs_off[np];
  if (n_obs && !obs_desc) return fail(VSM_ERR_INVALID, "null observation descriptor pointer");
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  stats_reset(c);
  AcceptParams prm;
  make_accept(threshold, dist_2_best, &prm);
  const double gate_t = gate_threshold(match_max_dist_2d);

  // One pinned staging block up: [kp_desc | obs_desc | kp_xy | proj_xy | obs_off]
  // (every section 32-byte aligned), one block down: [count u64, pad | pairs].
  auto al = [](size_t v) { return (v + 31) & ~(size_t)31; };
  const size_t o_kd = 0, o_od = o_kd + nk * DESC_BYTES, o_kx = o_od + al(std::max<size_t>(n_obs, 1) * DESC_BYTES);
  const size_t o_px = o_kx + al(nk * 16), o_of = o_px + al(np * 16), up_bytes = o_of + al((np + 1) * 4);
  const size_t cap = std::min(cap_pairs, nk);
  // (the latency path of vsm_match_pair: inputs staged by a kernel from mapped host memory, results
  // written straight into mapped host memory with a flag the host polls)
  const size_t down_bytes = 64 + nk * 8;  // [count u64 | flag u64 at 16 | pad to 64 | pairs]
  VSM_TRY(c->pin_in.reserve(up_bytes + 16));
  VSM_TRY(c->pin_out.reserve(down_bytes));
  VSM_TRY(c->stage.reserve(up_bytes + 16));
  const bool mapped = c->pin_out.dev != nullptr;
  if (!mapped) VSM_TRY(c->w_out.reserve(down_bytes));
  *(volatile u64*)((char*)c->pin_out.p + 16) = 0;  // the flag starts from 0 (see vsm_match_pair)
  char* h = (char*)c->pin_in.p;
  memcpy(h + o_kd, kp_desc, nk * DESC_BYTES);
  if (n_obs) memcpy(h + o_od, obs_desc, n_obs * DESC_BYTES);
  memcpy(h + o_kx, kp_xy, nk * 16);
  memcpy(h + o_px, proj_xy, np * 16);
  memcpy(h + o_of, obs_off, (np + 1) * 4);
  if (c->pin_in.dev) {
    const u32 n16 = (u32)((up_bytes + 15) / 16);
    hipLaunchKernelGGL(stage_in_kernel, dim3((n16 + 255) / 256), dim3(256), 0, c->stream, (const u32x4*)c->pin_in.dev,
                       (u32x4*)c->stage.p, n16);
  } else {
    HIP_TRY(hipMemcpyAsync(c->stage.p, h, up_bytes, hipMemcpyHostToDevice, c->stream));
  }
  char* d = (char*)c->stage.p;

  // Latency-bound: spread the projected points over enough waves to cover the chip
  // (1024 SIMDs), at least 16 points per wave; 4 waves (= 4 splits) per workgroup.
  const u32 tiles = (u32)((nk + 63) / 64);
  u32 splits = std::max<u32>(1, std::min<u32>((u32)((np + 15) / 16), (2048 + tiles - 1) / tiles));
  splits = std::min<u32>(round_up(splits, 4), 1024);
  const u32 split_stride = round_up((u32)nk, 64);
  VSM_TRY(c->scratch.reserve((size_t)split_stride * splits * sizeof(uint2)));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c->profiling) {
    VSM_TRY(ev_pair(c, &e0, &e1));
    c->ev_kind.push_back(0);
    HIP_TRY(hipEventRecord(e0, c->stream));
  }
  hipLaunchKernelGGL(landmark_scan_kernel, dim3(tiles, splits / 4), dim3(256), 0, c->stream,
                     (const u32x4*)(d + o_kd), (const double2*)(d + o_kx), (u32)nk,
                     (const double*)(d + o_px), (u32)np, (const u32*)(d + o_of),
                     (const u32x4*)(d + o_od), gate_t, splits, split_stride, (uint2*)c->scratch.p);
  HIP_TRY(hipGetLastError());
  if (c->profiling) HIP_TRY(hipEventRecord(e1, c->stream));
  c->scan_launches[0]++;
  hipLaunchKernelGGL(merge_splits_kernel, dim3((split_stride + 255) / 256), dim3(256), 0, c->stream,
                     (uint2*)c->scratch.p, split_stride, splits, split_stride);
  char* d_down = mapped ? (char*)c->pin_out.dev : (char*)c->w_out.p;
  const u64 seq = ++c->pair_seq;
  hipLaunchKernelGGL(landmark_emit_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint2*)c->scratch.p,
                     (u32)nk, prm, (u64*)d_down, (int2*)(d_down + 64), (u64)cap,
                     mapped ? (u64*)(d_down + 16) : (u64*)nullptr, seq);
  HIP_TRY(hipGetLastError());
  char* hd = (char*)c->pin_out.p;
  volatile u64* h_flag = (volatile u64*)(hd + 16);
  const bool seen = mapped && !c->profiling && poll_flag(h_flag, seq);
  if (!seen) {
    if (!mapped) HIP_TRY(hipMemcpyAsync(hd, c->w_out.p, 64 + cap * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  if (c->profiling) VSM_TRY(stats_collect(c));
  u64 count;
  memcpy(&count, hd, 8);
  *out_count = (size_t)count;
  if (count > cap_pairs) return fail(VSM_ERR_CAPACITY, "%llu matches need more than cap_pairs=%zu", (unsigned long long)count, cap_pairs);
  if (count) memcpy(out_pairs, hd + 64, count * 8);
  return VSM_OK;
}

static int check_image(const uint8_t* img, int width, int height, size_t pitch) {
  if (!img) return fail(VSM_ERR_INVALID, "null image pointer");
  if (width <= 0 || height <= 0 || pitch < (size_t)width || width > 32768 || height > 32768)
    return fail(VSM_ERR_INVALID, "bad image geometry %d x %d, pitch %zu", width, height, pitch);
  return VSM_OK;
}

// Rows packed to `width` into the pinned image buffer, one async H2D into c->image.  The
// pinned and device image buffers are used by nothing else, so later staging in the same
// call cannot disturb the copy in flight.
static int stage_image(vsm_ctx* c, const uint8_t* img, int width, int height, size_t pitch) {
  const size_t img_bytes = (size_t)width * height;
  VSM_TRY(c->pin_image.reserve(img_bytes));
  VSM_TRY(c->image.reserve(img_bytes));
  char* h = (char*)c->pin_image.p;
  if (pitch == (size_t)width) memcpy(h, img, img_bytes);
  else
    for (int y = 0; y < height; y++) memcpy(h + (size_t)y * width, img + (size_t)y * pitch, (size_t)width);
  // (a copy kernel reading mapped host memory, as vsm_match_pair uses for its 96 KB, gains nothing at 361 KB)
  HIP_TRY(hipMemcpyAsync(c->image.p, h, img_bytes, hipMemcpyHostToDevice, c->stream));
  return VSM_OK;
}

static int check_corners(const double* corners_xy, size_t n, int width, int height, bool need_border) {
  for (size_t i = 0; i < n; i++) {
    const double x = corners_xy[2 * i], y = corners_xy[2 * i + 1];
    // `const int cx = p[0]` (keypoints.h:158, 213) is undefined outside the int range
    if (!(std::fabs(x) < 1073741824.0) || !(std::fabs(y) < 1073741824.0))
      return fail(VSM_ERR_INVALID, "corner %zu is not finite or out of range", i);
    if (need_border) {
      // computeDescriptors reads up to 19 pixels away (keypoints.h:227-228); the reference only
      // ever passes corners that detectKeypoints kept (InBounds(.., EDGE_THRESHOLD), :145)
      const int cx = (int)x, cy = (int)y;
      if (cx < VSM_ORB_EDGE_THRESHOLD || cx >= width - VSM_ORB_EDGE_THRESHOLD || cy < VSM_ORB_EDGE_THRESHOLD ||
          cy >= height - VSM_ORB_EDGE_THRESHOLD)
        return fail(VSM_ERR_INVALID, "corner %zu (%d, %d) is closer than %d pixels to the image border: "
                    "the reference reads outside the image there", i, cx, cy, VSM_ORB_EDGE_THRESHOLD);
    }
  }
  return VSM_OK;
}

// computeAngles + computeDescriptors on the image already staged in c->image (mutex held,
// device current, corners validated).  As in vsm_match_pair everything stays in the compute queue: the
// kernels read the corners and (cos, sin) from, and write the moments and descriptors to, MAPPED host
// memory (each value crosses the bus once; a copy kernel or SDMA copy per direction and stage cost
// 5-10 us each with its waits).  Without a mapping the same buffers are copied.
static int orb_on_device(vsm_ctx* c, int width, int height, const double* corners_xy, size_t n,
                         int rotate_features, const double* angles_in, double* angles_out, uint64_t* desc_out) {
  // host staging: [corners | (cos, sin)] up, [moments | descriptors] down
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_xy = 0, o_cs = al(n * 16), up_bytes = o_cs + al(n * 16);
  const size_t o_mom = 0, o_desc = al(n * 8), down_bytes = o_desc + n * DESC_BYTES;
  VSM_TRY(c->pin_in.reserve(up_bytes));
  VSM_TRY(c->pin_out.reserve(down_bytes));
  const bool map_in = c->pin_in.dev != nullptr, map_out = c->pin_out.dev != nullptr;
  if (!map_in) VSM_TRY(c->stage.reserve(up_bytes));
  if (!map_out) VSM_TRY(c->w_out.reserve(down_bytes));
  char* h = (char*)c->pin_in.p;
  char* hd = (char*)c->pin_out.p;
  const char* d = map_in ? (const char*)c->pin_in.dev : (const char*)c->stage.p;  // what the kernels read
  char* dd = map_out ? (char*)c->pin_out.dev : (char*)c->w_out.p;                 // what they write
  const uint8_t* d_img = (const uint8_t*)c->image.p;
  memcpy(h + o_xy, corners_xy, n * 16);
  if (!map_in) HIP_TRY(hipMemcpyAsync(c->stage.p, h + o_xy, n * 16, hipMemcpyHostToDevice, c->stream));
  double* cs = (double*)(h + o_cs);
  std::vector<double> ang_local;
  double* ang = angles_out;
  if (!ang) {
    ang_local.resize(n);
    ang = ang_local.data();
  }
  if (angles_in) {
    for (size_t i = 0; i < n; i++) ang[i] = angles_in[i];  // computeDescriptors: kd.corner_angles as given
  } else if (rotate_features) {
    hipLaunchKernelGGL(orb_moments_kernel, dim3((u32)((n + 3) / 4)), dim3(256), 0, c->stream, d_img, width,
                       height, (u32)width, (const double2*)(d + o_xy), (u32)n, (int2*)(dd + o_mom), (const u32*)nullptr, (size_t)0);
    HIP_TRY(hipGetLastError());
    if (!map_out) HIP_TRY(hipMemcpyAsync(hd + o_mom, dd + o_mom, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int32_t* mom = (const int32_t*)(hd + o_mom);
    for (size_t i = 0; i < n; i++)  // keypoints.h:195, host libm
      ang[i] = std::atan2((double)mom[2 * i + 1], (double)mom[2 * i]);
  } else {
    for (size_t i = 0; i < n; i++) ang[i] = 0;  // keypoints.h:161
  }
  if (!desc_out) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VSM_OK;
  }
  for (size_t i = 0; i < n; i++) {  // keypoints.h:219, host libm
    cs[2 * i] = std::cos(ang[i]);
    cs[2 * i + 1] = std::sin(ang[i]);
  }
  if (!map_in) HIP_TRY(hipMemcpyAsync((char*)c->stage.p + o_cs, h + o_cs, n * 16, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(orb_descriptor_kernel, dim3((u32)n), dim3(256), 0, c->stream, d_img, width, height,
                     (u32)width, (const double2*)(d + o_xy), (const double2*)(d + o_cs), (u32)n,
                     (u64*)(dd + o_desc), (const u32*)nullptr, (size_t)0);
  HIP_TRY(hipGetLastError());
  if (!map_out) HIP_TRY(hipMemcpyAsync(hd + o_desc, dd + o_desc, n * DESC_BYTES, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(desc_out, hd + o_desc, n * DESC_BYTES);
  return VSM_OK;
}

static int orb_impl(vsm_ctx* c, const uint8_t* img, int width, int height, size_t pitch,
                    const double* corners_xy, size_t n, int rotate_features, const double* angles_in,
                    double* angles_out, uint64_t* desc_out) {
  VSM_TRY(check_ctx(c));
  if (n == 0) return VSM_OK;
  if (!corners_xy) return fail(VSM_ERR_INVALID, "null corner pointer");
  if (!angles_out && !desc_out) return fail(VSM_ERR_INVALID, "both outputs are null");
  VSM_TRY(check_image(img, width, height, pitch));
  if (n > VSM_MAX_SET) return fail(VSM_ERR_INVALID, "more than VSM_MAX_SET corners");
  VSM_TRY(check_corners(corners_xy, n, width, height, desc_out != nullptr));
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  VSM_TRY(stage_image(c, img, width, height, pitch));
  return orb_on_device(c, width, height, corners_xy, n, rotate_features, angles_in, angles_out, desc_out);
}

// The sequential pass of cv::goodFeaturesToTrack over the ordered candidates (key = value << 32 |
// pixel index, descending): a candidate is kept unless an already kept one lies closer than
// min_distance (grid of cells of cvRound(min_distance) pixels, 3 x 3 cells searched), until
// max_corners are kept.  Resumable: feed() may be called again with the candidates that follow.
struct MinDistancePass {
  u32 w;
  int cell = 1, gw = 0, gh = 0, max_corners;
  double md2;
  bool suppress, done = false;
  std::vector<int> head, next;
  std::vector<u32> kept;
  std::vector<double> xy;
  MinDistancePass(int width, int height, int max_corners_, double min_distance)
      : w((u32)width), max_corners(max_corners_), md2(min_distance * min_distance), suppress(min_distance >= 1) {
    if (suppress) {
      cell = (int)std::lrint(min_distance);  // cvRound
      gw = (width + cell - 1) / cell;
      gh = (height + cell - 1) / cell;
      head.assign((size_t)gw * gh, -1);
    }
    const size_t expect = max_corners > 0 ? std::min<size_t>((size_t)max_corners, 8192) : 2048;
    next.reserve(expect);
    kept.reserve(expect);
    xy.reserve(2 * expect);
  }
  void feed(const u64* cand, size_t nc) {
    for (size_t k = 0; k < nc && !done; k++) {
      const u32 idx = (u32)cand[k];
      const int y = (int)(idx / w), x = (int)(idx % w);
      if (suppress) {
        const int xc = x / cell, yc = y / cell;
        const int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1), x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
        bool good = true;
        for (int yy = y1; yy <= y2 && good; yy++)
          for (int xx = x1; xx <= x2 && good; xx++)
            for (int m = head[(size_t)yy * gw + xx]; m >= 0; m = next[m]) {
              const float ddx = (float)x - (float)(kept[m] % w), ddy = (float)y - (float)(kept[m] / w);
              if (ddx * ddx + ddy * ddy < md2) {
                good = false;
                break;
              }
            }
        if (!good) continue;
        next.push_back(head[(size_t)yc * gw + xc]);
        head[(size_t)yc * gw + xc] = (int)kept.size();
        kept.push_back(idx);
      }
      xy.push_back((double)x);
      xy.push_back((double)y);
      if (max_corners > 0 && (int)(xy.size() / 2) == max_corners) done = true;
    }
  }
};

// cv::goodFeaturesToTrack on the image staged in c->image: per-pixel work on the device,
// candidate ordering and the sequential minimum-distance pass on the host.
static int gftt_on_device(vsm_ctx* c, int width, int height, int max_corners, double quality,
                          double min_distance, std::vector<double>& xy) {
  xy.clear();
  const size_t n = (size_t)width * height;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  // device work area: gradients (later reused for the ordered candidates) | eig | candidates | misc
  const size_t o_grad = 0, o_eig = al(n * 8), o_cand = o_eig + al(n * 4), o_misc = o_cand + al(n * 8);
  VSM_TRY(c->gftt.reserve(o_misc + 256));
  char* d = (char*)c->gftt.p;
  u32* d_misc = (u32*)(d + o_misc);  // [0] max key, [1] candidate count
  HIP_TRY(hipMemsetAsync(d_misc, 0, 8, c->stream));
  const dim3 grid((u32)((width + 63) / 64), (u32)((height + 3) / 4)), block(256);
  const dim3 tiles((u32)((width + GFTT_TILE_W - 1) / GFTT_TILE_W), (u32)((height + GFTT_TILE_H - 1) / GFTT_TILE_H));
  hipLaunchKernelGGL(gftt_sobel_mineig_kernel, tiles, block, 0, c->stream, (const uint8_t*)c->image.p, width, height,
                     (u32)width, (float*)(d + o_eig), d_misc, (size_t)0);
  hipLaunchKernelGGL(gftt_candidates_kernel, grid, block, 0, c->stream, (const float*)(d + o_eig), width, height,
                     (const u32*)d_misc, quality, (u64*)(d + o_cand), d_misc + 1, (u32)n);
  // order them on the device (the gradient area is free again); sized for the largest count the
  // kernel sorts -- it reads the real count itself, so no host round trip is needed in between
  const u32 sort_blocks = (u32)((std::min<size_t>(n, GFTT_RANK_SORT_MAX) + GFTT_RANK_BLOCK - 1) / GFTT_RANK_BLOCK);
  // The count and the first GFTT_FIRST_COPY ordered candidates come down speculatively: written by the
  // sort kernel itself into mapped host memory (two copies and their waits cost ~15 us), else copied.
  constexpr size_t GFTT_FIRST_COPY = 4096;
  VSM_TRY(c->pin_out.reserve(256 + GFTT_FIRST_COPY * 8));
  const bool mapped = c->pin_out.dev != nullptr;
  char* hp = (char*)c->pin_out.p;
  char* hp_dev = (char*)c->pin_out.dev;
  hipLaunchKernelGGL(gftt_rank_sort_kernel, dim3(sort_blocks), dim3(GFTT_RANK_BLOCK), 0, c->stream, (const u64*)(d + o_cand),
                     (const u32*)(d_misc + 1), (u32)n, (u64*)(d + o_grad), 0u, mapped ? (u32*)hp_dev + 1 : (u32*)nullptr,
                     mapped ? (u64*)(hp_dev + 256) : (u64*)nullptr, (u32)GFTT_FIRST_COPY);
  HIP_TRY(hipGetLastError());
  if (!mapped) {
    HIP_TRY(hipMemcpyAsync(hp, d_misc, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(hp + 256, d + o_grad, std::min(n, GFTT_FIRST_COPY) * 8, hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  const size_t nc = std::min<size_t>(((const u32*)hp)[1], n);
  if (nc == 0) return VSM_OK;
  const bool device_sorted = nc <= GFTT_RANK_SORT_MAX;
  const size_t first = device_sorted ? std::min(nc, GFTT_FIRST_COPY) : 0;
  MinDistancePass pass(width, height, max_corners, min_distance);
  pass.feed((const u64*)(hp + 256), first);
  if (!pass.done && nc > first) {  // the pass wants more candidates than the speculative copy held
    VSM_TRY(c->pin.reserve(256 + nc * 8));  // nothing is in flight here
    hp = (char*)c->pin.p;
    HIP_TRY(hipMemcpyAsync(hp + 256, d + (device_sorted ? o_grad : o_cand), nc * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    u64* cand = (u64*)(hp + 256);
    // value descending, ties by the higher pixel address first (cv greaterThanPtr) = key descending
    if (!device_sorted) std::sort(cand, cand + nc, std::greater<u64>());
    pass.feed(cand + first, nc - first);
  }
  xy.swap(pass.xy);
  return VSM_OK;
}

// keypoints.h:144-148: keep the points with InBounds(x, y, EDGE_THRESHOLD)
static void edge_filter(std::vector<double>& xy, int width, int height) {
  size_t m = 0;
  for (size_t i = 0; i + 1 < xy.size(); i += 2) {
    const float x = (float)xy[i], y = (float)xy[i + 1], b = (float)VSM_ORB_EDGE_THRESHOLD;
    if (b <= x && x < (float)width - b && b <= y && y < (float)height - b) {
      xy[m++] = xy[i];
      xy[m++] = xy[i + 1];
    }
  }
  xy.resize(m);
}

static int detect_impl(vsm_ctx* c, const uint8_t* img, int width, int height, size_t pitch, int max_corners,
                       double quality, double min_distance, bool edge, int describe, int rotate_features,
                       double* corners_out, double* angles_out, uint64_t* desc_out, size_t cap, size_t* out_count) {
  VSM_TRY(check_ctx(c));
  if (!out_count) return fail(VSM_ERR_INVALID, "null out_count");
  *out_count = 0;
  VSM_TRY(check_image(img, width, height, pitch));
  if (!(quality > 0) || !(min_distance >= 0) || max_corners < 0)  // the CV_Assert of cv::goodFeaturesToTrack
    return fail(VSM_ERR_INVALID, "need qualityLevel > 0, minDistance >= 0, maxCorners >= 0");
  if (cap && !corners_out) return fail(VSM_ERR_INVALID, "null corners_out with non-zero capacity");
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  VSM_TRY(stage_image(c, img, width, height, pitch));
  std::vector<double> xy;
  VSM_TRY(gftt_on_device(c, width, height, max_corners, quality, min_distance, xy));
  if (edge) edge_filter(xy, width, height);
  const size_t n = xy.size() / 2;
  *out_count = n;
  if (n > cap) return fail(VSM_ERR_CAPACITY, "%zu corners need more than cap=%zu", n, cap);
  if (n == 0) return VSM_OK;
  memcpy(corners_out, xy.data(), n * 16);
  if (!describe) return VSM_OK;
  if (!angles_out || !desc_out) return fail(VSM_ERR_INVALID, "null angles_out or desc_out");
  return orb_on_device(c, width, height, xy.data(), n, rotate_features, nullptr, angles_out, desc_out);
}

int vsm_good_features_to_track(vsm_ctx* c, const uint8_t* img, int width, int height, size_t pitch,
                               int max_corners, double quality_level, double min_distance,
                               double* corners_out, size_t cap, size_t* out_count) {
  return detect_impl(c, img, width, height, pitch, max_corners, quality_level, min_distance, false, 0, 0,
                     corners_out, nullptr, nullptr, cap, out_count);
}

int vsm_detect_keypoints(vsm_ctx* c, const uint8_t* img, int width, int height, size_t pitch,
                         int num_features, double* corners_out, size_t cap, size_t* out_count) {
  if (num_features < 0) return fail(VSM_ERR_INVALID, "num_features must be >= 0");
  return detect_impl(c, img, width, height, pitch, num_features, 0.01, 8.0, true, 0, 0, corners_out, nullptr,
                     nullptr, cap, out_count);
}

int vsm_detect_and_describe(vsm_ctx* c, const uint8_t* img, int width, int height, size_t pitch,
                            int num_features, int rotate_features, double* corners_out, double* angles_out,
                            uint64_t* desc_out, size_t cap, size_t* out_count) {
  if (num_features < 0) return fail(VSM_ERR_INVALID, "num_features must be >= 0");
  return detect_impl(c, img, width, height, pitch, num_features, 0.01, 8.0, true, 1, rotate_features,
                     corners_out, angles_out, desc_out, cap, out_count);
}

int vsm_orb_describe(vsm_ctx* c, const uint8_t* img, int width, int height, size_t pitch,
                     const double* corners_xy, size_t n, int rotate_features, double* angles_out,
                     uint64_t* desc_out) {
  return orb_impl(c, img, width, height, pitch, corners_xy, n, rotate_features, nullptr, angles_out, desc_out);
}

int vsm_orb_angles(vsm_ctx* c, const uint8_t* img, int width, int height, size_t pitch,
                   const double* corners_xy, size_t n, int rotate_features, double* angles_out) {
  if (n && !angles_out) return fail(VSM_ERR_INVALID, "null angles_out");
  return orb_impl(c, img, width, height, pitch, corners_xy, n, rotate_features, nullptr, angles_out, nullptr);
}

int vsm_orb_descriptors(vsm_ctx* c, const uint8_t* img, int width, int height, size_t pitch,
                        const double* corners_xy, const double* angles, size_t n, uint64_t* desc_out) {
  if (n && (!angles || !desc_out)) return fail(VSM_ERR_INVALID, "null angles or desc_out");
  return orb_impl(c, img, width, height, pitch, corners_xy, n, 1, angles, nullptr, desc_out);
}

int vsm_voc_upload(vsm_ctx* c, const uint64_t* node_desc, const uint32_t* child_off, const uint32_t* child_idx,
                   const uint32_t* node_id, const uint32_t* word_id, const double* weight, size_t n_nodes) {
  VSM_TRY(check_ctx(c));
  if (n_nodes == 0 || n_nodes >= 0x7fffffffull) return fail(VSM_ERR_INVALID, "vocabulary needs 1 .. 2^31 nodes");
  if (!node_desc || !child_off || !node_id || !word_id || !weight) return fail(VSM_ERR_INVALID, "null vocabulary array");
  if (child_off[0] != 0) return fail(VSM_ERR_INVALID, "child_off[0] must be 0");
  for (size_t i = 0; i < n_nodes; i++) {
    if (child_off[i + 1] < child_off[i]) return fail(VSM_ERR_INVALID, "child_off not ascending at node %zu", i);
    if (node_id[i] >= n_nodes) return fail(VSM_ERR_INVALID, "node_id[%zu] out of range", i);
  }
  const size_t n_child = child_off[n_nodes];
  if (n_child && !child_idx) return fail(VSM_ERR_INVALID, "null child_idx");
  for (size_t k = 0; k < n_child; k++)
    if (child_idx[k] >= n_nodes) return fail(VSM_ERR_INVALID, "child_idx[%zu] out of range", k);
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t* o = c->voc_off;  // desc | child_off | child_idx | node_id | word_id | weight
  o[0] = 0;
  o[1] = o[0] + al(n_nodes * DESC_BYTES);
  o[2] = o[1] + al((n_nodes + 1) * 4);
  o[3] = o[2] + al(std::max<size_t>(n_child, 1) * 4);
  o[4] = o[3] + al(n_nodes * 4);
  o[5] = o[4] + al(n_nodes * 4);
  VSM_TRY(c->voc.reserve(o[5] + al(n_nodes * 8)));
  char* d = (char*)c->voc.p;
  HIP_TRY(hipMemcpyAsync(d + o[0], node_desc, n_nodes * DESC_BYTES, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d + o[1], child_off, (n_nodes + 1) * 4, hipMemcpyHostToDevice, c->stream));
  if (n_child) HIP_TRY(hipMemcpyAsync(d + o[2], child_idx, n_child * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d + o[3], node_id, n_nodes * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d + o[4], word_id, n_nodes * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d + o[5], weight, n_nodes * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->voc_nodes = n_nodes;
  return VSM_OK;
}

int vsm_bow_words(vsm_ctx* c, const uint64_t* desc, size_t n, uint32_t* word_out, double* weight_out) {
  VSM_TRY(check_ctx(c));
  if (n == 0) return VSM_OK;
  if (!desc || !word_out || !weight_out) return fail(VSM_ERR_INVALID, "null pointer");
  if (n > VSM_MAX_SET) return fail(VSM_ERR_INVALID, "more than VSM_MAX_SET features");
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  if (c->voc_nodes == 0) return fail(VSM_ERR_INVALID, "no vocabulary: call vsm_voc_upload first");
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_w = 0, o_v = al(n * 4), o_f = o_v + al(n * 8), down = o_f + 256;
  VSM_TRY(c->pin.reserve(n * DESC_BYTES + down));
  VSM_TRY(c->stage.reserve(n * DESC_BYTES));
  VSM_TRY(c->w_out.reserve(down));
  char* h = (char*)c->pin.p;
  memcpy(h, desc, n * DESC_BYTES);
  HIP_TRY(hipMemcpyAsync(c->stage.p, h, n * DESC_BYTES, hipMemcpyHostToDevice, c->stream));
  char* dd = (char*)c->w_out.p;
  HIP_TRY(hipMemsetAsync(dd + o_f, 0, 4, c->stream));
  const char* v = (const char*)c->voc.p;
  const size_t* o = c->voc_off;
  hipLaunchKernelGGL(bow_words_kernel, dim3((u32)((n + 255) / 256)), dim3(256), 0, c->stream,
                     (const u32x4*)c->stage.p, (u32)n, (const u32x4*)(v + o[0]), (const u32*)(v + o[1]),
                     (const u32*)(v + o[2]), (const u32*)(v + o[3]), (const u32*)(v + o[4]),
                     (const double*)(v + o[5]), (u32)c->voc_nodes, (u32*)(dd + o_w), (double*)(dd + o_v),
                     (u32*)(dd + o_f));
  HIP_TRY(hipGetLastError());
  char* hd = h + n * DESC_BYTES;
  HIP_TRY(hipMemcpyAsync(hd, dd, down, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  u32 flags;
  memcpy(&flags, hd + o_f, 4);
  if (flags & 1u)
    return fail(VSM_ERR_INVALID, "a feature reaches a node none of whose children is closer than 256: "
                                 "the reference's loop (bow_voc.h:77-88) does not terminate there");
  memcpy(word_out, hd + o_w, n * 4);
  memcpy(weight_out, hd + o_v, n * 8);
  return VSM_OK;
}

// ---- BoW database (bow_db.h) ------------------------------------------------------------------

}  // extern "C"

namespace {

// The BoW database entry points build host-side tables whose size the caller's data decides: no
// exception leaves the C ABI.
template <class F>
int no_throw(F&& f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return fail(VSM_ERR_NOMEM, "out of host memory");
  } catch (const std::exception& e) {
    return fail(VSM_ERR_INVALID, "%s", e.what());
  }
}

constexpr u32 BOWDB_MAX_WORD = 1u << 26;              // dense word table: 4 B per word id
constexpr size_t BOWDB_MAX_IMAGES = (size_t)1 << 24;  // 64 lists of a query batch are counted in 32 bits
constexpr size_t BOWDB_WORK_BUDGET = (size_t)256 << 20;  // score rows + candidate lists per launch

int check_bow_vectors(const uint32_t* off, const uint32_t* words, const double* values, size_t n, const char* what) {
  if (!off) return fail(VSM_ERR_INVALID, "null %s offsets", what);
  if (off[0] != 0) return fail(VSM_ERR_INVALID, "%s offsets must start at 0", what);
  for (size_t i = 0; i < n; i++)
    if (off[i + 1] < off[i]) return fail(VSM_ERR_INVALID, "%s offsets not ascending at %zu", what, i);
  const size_t total = off[n];
  if (total && (!words || !values)) return fail(VSM_ERR_INVALID, "null %s words or values", what);
  for (size_t k = 0; k < total; k++)
    if (!std::isfinite(values[k])) return fail(VSM_ERR_INVALID, "%s value %zu is not finite", what, k);
  return VSM_OK;
}

// Queries already on the device (d_q*: offsets, words, values of n_q vectors); limit: nullptr = every
// image, else per query; limit_iota: query k sees the images before k.  Results into host arrays
// with the caller's stride num_results.
int bowdb_query_device(vsm_ctx* c, const u32* d_qoff, const u32* d_qwords, const double* d_qvals, size_t n_q,
                       const u32* d_limit, bool limit_iota, size_t num_results, int32_t* out_idx, double* out_score,
                       uint32_t* out_n) {
  const size_t n_img = c->bowdb_images;
  const u32 nr = (u32)std::min<size_t>(num_results, n_img);
  for (size_t i = 0; i < n_q * num_results; i++) out_idx[i] = -1;
  for (size_t i = 0; i < n_q * num_results; i++) out_score[i] = 0.0;
  for (size_t i = 0; i < n_q; i++) out_n[i] = 0;
  if (n_q == 0 || nr == 0) return VSM_OK;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  // score row in LDS when it fits (one LDS round trip per word instead of one through HBM)
  const bool lds_row = n_img <= BOWDB_LDS_MAX_IMAGES;
  const size_t lds_bytes = BOWDB_LDS_FIXED + (lds_row ? n_img * 8 : 0);
  auto kernel = lds_row ? bowdb_query_kernel<true> : bowdb_query_kernel<false>;
  if (lds_bytes > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  const size_t per_query = n_img * (lds_row ? 24 : 32);  // candidates + survivors (+ the row)
  const size_t chunk = std::max<size_t>(1, std::min<size_t>(n_q, BOWDB_WORK_BUDGET / std::max<size_t>(per_query, 1)));
  const size_t o_rows = 0, o_cs = o_rows + (lds_row ? 0 : al(chunk * n_img * 8)), o_ss = o_cs + al(chunk * n_img * 8),
               o_ci = o_ss + al(chunk * n_img * 8), o_si = o_ci + al(chunk * n_img * 4), o_oi = o_si + al(chunk * n_img * 4),
               o_os = o_oi + al(n_q * nr * 4), o_on = o_os + al(n_q * nr * 8), total = o_on + al(n_q * 4);
  VSM_TRY(c->bowdb_work.reserve(total));
  char* w = (char*)c->bowdb_work.p;
  const char* d = (const char*)c->bowdb.p;
  const size_t* o = c->bowdb_off;
  for (size_t q0 = 0; q0 < n_q; q0 += chunk) {
    const u32 nq = (u32)std::min(chunk, n_q - q0);
    hipLaunchKernelGGL(kernel, dim3(nq), dim3(64), lds_bytes, c->stream, d_qoff, d_qwords, d_qvals, (u32)q0, d_limit,
                       limit_iota ? 1u : 0u, (const u32*)(d + o[0]), c->bowdb_dense, (const u32*)(d + o[1]),
                       (const double*)(d + o[2]), (u32)n_img, (double*)(w + o_rows), (u32*)(w + o_ci),
                       (double*)(w + o_cs), (u32*)(w + o_si), (double*)(w + o_ss), nr, (int*)(w + o_oi),
                       (double*)(w + o_os), (u32*)(w + o_on));
    HIP_TRY(hipGetLastError());
  }
  const size_t down = total - o_oi;
  VSM_TRY(c->pin.reserve(down));
  char* h = (char*)c->pin.p;
  HIP_TRY(hipMemcpyAsync(h, w + o_oi, down, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const int32_t* hi = (const int32_t*)h;
  const double* hs = (const double*)(h + (o_os - o_oi));
  const uint32_t* hn = (const uint32_t*)(h + (o_on - o_oi));
  for (size_t q = 0; q < n_q; q++) {
    out_n[q] = hn[q];
    for (u32 r = 0; r < hn[q]; r++) {  // only the first out_n entries of a row were written
      out_idx[q * num_results + r] = hi[q * nr + r];
      out_score[q * num_results + r] = hs[q * nr + r];
    }
  }
  return VSM_OK;
}

}  // namespace

extern "C" {

int vsm_bowdb_upload(vsm_ctx* c, const uint32_t* vec_off, const uint32_t* words, const double* values, size_t n_img) {
 return no_throw([&]() -> int {
  VSM_TRY(check_ctx(c));
  if (n_img > BOWDB_MAX_IMAGES) return fail(VSM_ERR_INVALID, "more than 2^24 images");
  VSM_TRY(check_bow_vectors(vec_off, words, values, n_img, "database"));
  const size_t total = vec_off[n_img];
  u32 max_word = 0;
  for (size_t k = 0; k < total; k++) max_word = std::max(max_word, words[k]);
  if (total && max_word >= BOWDB_MAX_WORD) return fail(VSM_ERR_INVALID, "word id %u: the word table is dense, ids must be < 2^26", max_word);
  const size_t n_dense = total ? (size_t)max_word + 1 : 0;
  // insert() of the images in order (bow_db.h:55-60) = a counting sort by word that keeps image order
  std::vector<u32> start(n_dense + 1, 0), inv_img(total);
  std::vector<double> inv_val(total);
  {
    std::vector<u32> seen(n_dense, 0);  // last image (+1) that named the word: the reference's BowVector has a word once
    for (size_t i = 0; i < n_img; i++)
      for (size_t k = vec_off[i]; k < vec_off[i + 1]; k++) {
        if (seen[words[k]] == i + 1) return fail(VSM_ERR_INVALID, "image %zu names word %u twice", i, words[k]);
        seen[words[k]] = (u32)i + 1;
        start[words[k] + 1]++;
      }
    for (size_t w = 0; w < n_dense; w++) start[w + 1] += start[w];
    std::vector<u32> fill(start.begin(), start.end() - 1);
    for (size_t i = 0; i < n_img; i++)
      for (size_t k = vec_off[i]; k < vec_off[i + 1]; k++) {
        const u32 at = fill[words[k]]++;
        inv_img[at] = (u32)i;
        inv_val[at] = values[k];
      }
  }
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  c->bowdb_ready = false;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t* o = c->bowdb_off;  // inv_start | inv_img | inv_val | vec_off | words | values
  o[0] = 0;
  o[1] = o[0] + al((n_dense + 1) * 4);
  o[2] = o[1] + al(std::max<size_t>(total, 1) * 4);
  o[3] = o[2] + al(std::max<size_t>(total, 1) * 8);
  o[4] = o[3] + al((n_img + 1) * 4);
  o[5] = o[4] + al(std::max<size_t>(total, 1) * 4);
  VSM_TRY(c->bowdb.reserve(o[5] + al(std::max<size_t>(total, 1) * 8)));
  char* d = (char*)c->bowdb.p;
  HIP_TRY(hipMemcpyAsync(d + o[0], start.data(), (n_dense + 1) * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(d + o[3], vec_off, (n_img + 1) * 4, hipMemcpyHostToDevice, c->stream));
  if (total) {
    HIP_TRY(hipMemcpyAsync(d + o[1], inv_img.data(), total * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d + o[2], inv_val.data(), total * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d + o[4], words, total * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d + o[5], values, total * 8, hipMemcpyHostToDevice, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));  // the sources are pageable host memory of this call
  c->bowdb_images = (u32)n_img;
  c->bowdb_dense = (u32)n_dense;
  c->bowdb_entries = total;
  c->bowdb_ready = true;
  return VSM_OK;
 });
}

int vsm_bowdb_query(vsm_ctx* c, const uint32_t* q_off, const uint32_t* q_words, const double* q_values, size_t n_q,
                    const uint32_t* db_limit, size_t num_results, int32_t* out_idx, double* out_score, uint32_t* out_n) {
 return no_throw([&]() -> int {
  VSM_TRY(check_ctx(c));
  if (n_q == 0) return VSM_OK;
  if (n_q >= 0x7fffffffull) return fail(VSM_ERR_INVALID, "more than 2^31 queries");
  if (!out_n || (num_results && (!out_idx || !out_score))) return fail(VSM_ERR_INVALID, "null output");
  VSM_TRY(check_bow_vectors(q_off, q_words, q_values, n_q, "query"));
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  if (!c->bowdb_ready) return fail(VSM_ERR_INVALID, "no BoW database: call vsm_bowdb_upload first");
  const size_t total = q_off[n_q];
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_off = 0, o_w = al((n_q + 1) * 4), o_v = o_w + al(std::max<size_t>(total, 1) * 4),
               o_l = o_v + al(std::max<size_t>(total, 1) * 8), up = o_l + al(n_q * 4);
  VSM_TRY(c->pin.reserve(up));
  VSM_TRY(c->stage.reserve(up));
  char* h = (char*)c->pin.p;
  memcpy(h + o_off, q_off, (n_q + 1) * 4);
  if (total) {
    memcpy(h + o_w, q_words, total * 4);
    memcpy(h + o_v, q_values, total * 8);
  }
  if (db_limit) memcpy(h + o_l, db_limit, n_q * 4);
  HIP_TRY(hipMemcpyAsync(c->stage.p, h, up, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // bowdb_query_device reuses the pinned buffer for the results
  const char* s = (const char*)c->stage.p;
  return bowdb_query_device(c, (const u32*)(s + o_off), (const u32*)(s + o_w), (const double*)(s + o_v), n_q,
                            db_limit ? (const u32*)(s + o_l) : nullptr, false, num_results, out_idx, out_score, out_n);
 });
}

int vsm_bow_select_pairs(vsm_ctx* c, const int64_t* frame_id, size_t num_candidates, uint32_t (*pairs)[2], size_t cap,
                         size_t* n_pairs) {
 return no_throw([&]() -> int {
  VSM_TRY(check_ctx(c));
  if (!n_pairs) return fail(VSM_ERR_INVALID, "null n_pairs");
  *n_pairs = 0;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  if (!c->bowdb_ready) return fail(VSM_ERR_INVALID, "no BoW database: call vsm_bowdb_upload first");
  const size_t n = c->bowdb_images;
  if (n == 0 || num_candidates == 0) return VSM_OK;
  if (!frame_id) return fail(VSM_ERR_INVALID, "null frame_id");
  const size_t nr = std::min(num_candidates, n);
  std::vector<int32_t> idx(n * nr);
  std::vector<double> score(n * nr);
  std::vector<uint32_t> cnt(n);
  const char* d = (const char*)c->bowdb.p;
  const size_t* o = c->bowdb_off;
  // sfm.cpp:1373-1392: image k is queried against the images inserted before it, then inserted
  VSM_TRY(bowdb_query_device(c, (const u32*)(d + o[3]), (const u32*)(d + o[4]), (const double*)(d + o[5]), n, nullptr,
                             true, nr, idx.data(), score.data(), cnt.data()));
  size_t m = 0;
  for (size_t k = 0; k < n; k++)
    for (u32 r = 0; r < cnt[k]; r++) {
      const int32_t j = idx[k * nr + r];
      if (frame_id[k] == frame_id[j]) continue;  // sfm.cpp:1386: stereo partners are not paired here
      if (m < cap && pairs) {
        pairs[m][0] = (uint32_t)k;
        pairs[m][1] = (uint32_t)j;
      }
      m++;
    }
  *n_pairs = m;
  if (m > cap) return fail(VSM_ERR_CAPACITY, "%zu pairs, capacity %zu", m, cap);
  return VSM_OK;
 });
}

}  // extern "C"

// ---- the detection + ORB front end over a whole set of images (sfm.cpp:1191-1201) --------------

namespace {

unsigned host_threads(size_t n_items) {
  unsigned t = 8;
  if (const char* e = getenv("VSM_HOST_THREADS")) {
    const long v = strtol(e, nullptr, 10);
    if (v >= 1 && v <= 64) t = (unsigned)v;
  }
  const unsigned hw = std::thread::hardware_concurrency();
  if (hw) t = std::min(t, hw);
  return (unsigned)std::min<size_t>(std::max(1u, t), std::max<size_t>(n_items, 1));
}

// f(i) for i in [0, n) on host threads (the per-image host steps of the front end: packing rows,
// the sequential minimum-distance pass, atan2 / cos / sin).  The first exception is rethrown here.
template <class F>
void parallel_items(size_t n, size_t min_per_thread, F&& f) {
  const unsigned t = host_threads(n / std::max<size_t>(min_per_thread, 1));
  if (t <= 1) {
    for (size_t i = 0; i < n; i++) f(i);
    return;
  }
  std::atomic<size_t> next{0};
  std::exception_ptr err;
  std::mutex err_mu;
  auto body = [&]() {
    try {
      for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i);
    } catch (...) {
      std::lock_guard<std::mutex> lk(err_mu);
      if (!err) err = std::current_exception();
      next.store(n);
    }
  };
  std::vector<std::thread> th;
  for (unsigned k = 1; k < t; k++) th.emplace_back(body);
  body();
  for (auto& x : th) x.join();
  if (err) std::rethrow_exception(err);
}

constexpr size_t BATCH_DEVICE_BUDGET = (size_t)8 << 30;  // images + detector work areas of one chunk
constexpr size_t BATCH_MAX_CHUNK = 16384;                // images per chunk (grid.z)
constexpr size_t BATCH_FIRST_COPY = 4096;                // ordered candidates per image in the speculative download

}  // namespace

extern "C" {

int vsm_detect_and_describe_batch(vsm_ctx* c, const uint8_t* const* imgs, size_t n_img, int width, int height, size_t pitch,
                                  int num_features, int rotate_features, uint32_t* out_offsets, double* corners_out,
                                  double* angles_out, uint64_t* desc_out, size_t cap, int keep_as_database) {
 return no_throw([&]() -> int {
  VSM_TRY(check_ctx(c));
  if (!out_offsets) return fail(VSM_ERR_INVALID, "null out_offsets");
  for (size_t i = 0; i <= n_img; i++) out_offsets[i] = 0;
  if (num_features < 0) return fail(VSM_ERR_INVALID, "num_features must be >= 0");
  if (keep_as_database && num_features == 0)
    return fail(VSM_ERR_INVALID, "keep_as_database needs num_features > 0 (the database is sized before detection)");
  if (n_img >= 0x7fffffffull / std::max(1, num_features)) return fail(VSM_ERR_INVALID, "too many images");
  if (n_img && !imgs) return fail(VSM_ERR_INVALID, "null image list");
  for (size_t i = 0; i < n_img; i++) VSM_TRY(check_image(imgs[i], width, height, pitch));
  const bool want_out = corners_out || angles_out || desc_out;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  if (keep_as_database) {  // as vsm_db_upload: from here on the old database is gone
    c->db = nullptr;
    c->img_off.clear();
    VSM_TRY(c->db_owned.reserve(std::max<size_t>(n_img * (size_t)num_features * DESC_BYTES, 256)));
  }
  if (n_img == 0) return keep_as_database ? set_db(c, c->db_owned.p, out_offsets, 0) : VSM_OK;

  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t n = (size_t)width * height;
  size_t chunk = std::max<size_t>(1, std::min({n_img, BATCH_MAX_CHUNK, BATCH_DEVICE_BUDGET / (n * 21 + 8)}));
  if (const char* e = getenv("VSM_BATCH_CHUNK")) {  // tests: several chunks without thousands of images
    const long v = strtol(e, nullptr, 10);
    if (v >= 1) chunk = std::min(chunk, (size_t)v);
  }
  // detector work areas of a chunk: gradients (later the ordered candidates) | eig | candidates | (max key, count) pairs
  const size_t o_grad = 0, o_eig = al(chunk * n * 8), o_cand = o_eig + al(chunk * n * 4), o_misc = o_cand + al(chunk * n * 8);
  VSM_TRY(c->gftt.reserve(o_misc + al(chunk * 8)));
  VSM_TRY(c->image.reserve(chunk * n));
  VSM_TRY(c->pin_image.reserve(chunk * n));
  char* d = (char*)c->gftt.p;
  const uint8_t* d_img = (const uint8_t*)c->image.p;
  const size_t first = std::min(n, BATCH_FIRST_COPY);
  size_t total = 0;          // keypoints so far, over all chunks
  bool overflow = false;     // the caller's arrays are too small: keep counting, stop writing
  std::vector<std::vector<double>> xy(chunk);
  std::vector<u64> more;     // an image whose pass needs more than the speculative copy
  const size_t lds_sort_bytes = (size_t)std::min(GFTT_LDS_SORT_BLOCK, gftt_sort_size((u32)std::min<size_t>(n, GFTT_RANK_SORT_MAX))) * 8;
  if (lds_sort_bytes > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void*)gftt_lds_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sort_bytes));

  const bool trace = getenv("VSM_TRACE_BATCH") != nullptr;
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "  batch: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  for (size_t i0 = 0; i0 < n_img; i0 += chunk) {
    const size_t nb = std::min(chunk, n_img - i0);
    // -- images up: rows packed to `width` into pinned memory by host threads, one copy
    char* hi = (char*)c->pin_image.p;
    parallel_items(nb, 4, [&](size_t b) {
      const uint8_t* src = imgs[i0 + b];
      char* dst = hi + b * n;
      if (pitch == (size_t)width) memcpy(dst, src, n);
      else
        for (int y = 0; y < height; y++) memcpy(dst + (size_t)y * width, src + (size_t)y * pitch, (size_t)width);
    });
    lap("rows into pinned memory");
    HIP_TRY(hipMemcpyAsync(c->image.p, hi, nb * n, hipMemcpyHostToDevice, c->stream));
    // -- detection on the device, every image of the chunk in each launch
    u32* d_misc = (u32*)(d + o_misc);
    HIP_TRY(hipMemsetAsync(d_misc, 0, nb * 8, c->stream));
    const dim3 grid((u32)((width + 63) / 64), (u32)((height + 3) / 4), (u32)nb), block(256);
    const dim3 tiles((u32)((width + GFTT_TILE_W - 1) / GFTT_TILE_W), (u32)((height + GFTT_TILE_H - 1) / GFTT_TILE_H), (u32)nb);
    hipLaunchKernelGGL(gftt_sobel_mineig_kernel, tiles, block, 0, c->stream, d_img, width, height, (u32)width,
                       (float*)(d + o_eig), d_misc, n);
    hipLaunchKernelGGL(gftt_candidates_kernel, grid, block, 0, c->stream, (const float*)(d + o_eig), width, height,
                       (const u32*)d_misc, 0.01, (u64*)(d + o_cand), d_misc + 1, (u32)n);
    const u32 sort_blocks = (u32)((std::min<size_t>(n, GFTT_RANK_SORT_MAX) + GFTT_RANK_BLOCK - 1) / GFTT_RANK_BLOCK);
    // order: one workgroup per image in LDS; the brute-force kernel only for images beyond its capacity
    hipLaunchKernelGGL(gftt_lds_sort_kernel, dim3((u32)nb), dim3(1024), lds_sort_bytes, c->stream, (const u64*)(d + o_cand),
                       (const u32*)(d_misc + 1), (u32)n, (u64*)(d + o_grad));
    if (!gftt_lds_sorts((u32)std::min<size_t>(n, GFTT_RANK_SORT_MAX), (u32)n))  // small images: the padded size may not fit
      hipLaunchKernelGGL(gftt_rank_sort_kernel, dim3(sort_blocks, 1, (u32)nb), dim3(GFTT_RANK_BLOCK), 0, c->stream,
                         (const u64*)(d + o_cand), (const u32*)(d_misc + 1), (u32)n, (u64*)(d + o_grad), 1u, (u32*)nullptr,
                         (u64*)nullptr, 0u);
    HIP_TRY(hipGetLastError());
    // -- one download: the (max key, count) pairs and the first ordered candidates of every image
    const size_t h_cand = al(nb * 8);
    VSM_TRY(c->pin.reserve(h_cand + nb * first * 8));
    char* hp = (char*)c->pin.p;
    HIP_TRY(hipMemcpyAsync(hp, d_misc, nb * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpy2DAsync(hp + h_cand, first * 8, d + o_grad, n * 8, first * 8, nb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    lap("upload, detection, download");
    // -- the sequential pass + the EDGE_THRESHOLD filter, images in parallel on host threads
    std::vector<MinDistancePass> passes;
    passes.reserve(nb);
    for (size_t b = 0; b < nb; b++) passes.emplace_back(width, height, num_features, 8.0);
    std::vector<uint8_t> unfinished(nb, 0);
    const u32* h_misc = (const u32*)hp;
    parallel_items(nb, 4, [&](size_t b) {
      const size_t nc = std::min<size_t>(h_misc[2 * b + 1], n);
      const bool device_sorted = nc <= GFTT_RANK_SORT_MAX;
      const size_t have = device_sorted ? std::min(nc, first) : 0;
      passes[b].feed((const u64*)(hp + h_cand) + b * first, have);
      unfinished[b] = !passes[b].done && nc > have;
      if (!unfinished[b]) {
        xy[b].swap(passes[b].xy);
        edge_filter(xy[b], width, height);
      }
    });
    lap("minimum-distance pass");
    for (size_t b = 0; b < nb; b++) {  // rare: fetch the image's whole candidate list
      if (!unfinished[b]) continue;
      const size_t nc = std::min<size_t>(h_misc[2 * b + 1], n);
      const bool device_sorted = nc <= GFTT_RANK_SORT_MAX;
      const size_t have = device_sorted ? std::min(nc, first) : 0;
      more.resize(nc);
      HIP_TRY(hipMemcpy(more.data(), d + (device_sorted ? o_grad : o_cand) + b * n * 8, nc * 8, hipMemcpyDeviceToHost));
      if (!device_sorted) std::sort(more.begin(), more.end(), std::greater<u64>());
      passes[b].feed(more.data() + have, nc - have);
      xy[b].swap(passes[b].xy);
      edge_filter(xy[b], width, height);
    }
    lap("unfinished passes");
    size_t n_chunk = 0;
    for (size_t b = 0; b < nb; b++) {
      out_offsets[i0 + b] = (uint32_t)total + (uint32_t)n_chunk;
      n_chunk += xy[b].size() / 2;
    }
    out_offsets[i0 + nb] = (uint32_t)(total + n_chunk);
    if (total + n_chunk >= 0x7fffffffull) return fail(VSM_ERR_INVALID, "more than 2^31 keypoints");
    if (want_out && total + n_chunk > cap) overflow = true;
    const bool write_out = want_out && !overflow;
    if (n_chunk && (write_out || keep_as_database)) {
      // -- ORB: corners + image index per keypoint up, moments down, atan2 / cos / sin on host threads
      // (the caller's libm, as the reference), (cos, sin) up, descriptors -- into the database or down
      const size_t o_xy = 0, o_ki = al(n_chunk * 16), o_cs = o_ki + al(n_chunk * 4), up_bytes = o_cs + al(n_chunk * 16);
      const size_t o_mom = 0, o_desc = al(n_chunk * 8), down_bytes = o_desc + n_chunk * DESC_BYTES;
      VSM_TRY(c->pin.reserve(up_bytes + down_bytes));
      VSM_TRY(c->stage.reserve(up_bytes));
      VSM_TRY(c->w_out.reserve(down_bytes));
      char* h = (char*)c->pin.p;
      char* hd = h + up_bytes;
      char* ds = (char*)c->stage.p;
      char* dd = (char*)c->w_out.p;
      parallel_items(nb, 16, [&](size_t b) {
        const size_t k = out_offsets[i0 + b] - total;
        memcpy(h + o_xy + k * 16, xy[b].data(), xy[b].size() * 8);
        u32* ki = (u32*)(h + o_ki) + k;
        for (size_t i = 0; i < xy[b].size() / 2; i++) ki[i] = (u32)b;
      });
      lap("edge filter, corner lists");
      HIP_TRY(hipMemcpyAsync(ds, h, o_cs, hipMemcpyHostToDevice, c->stream));  // corners and image indices
      double* cs = (double*)(h + o_cs);
      std::vector<double> ang_local;
      double* ang = (write_out && angles_out) ? angles_out + total : nullptr;
      if (!ang) {
        ang_local.resize(n_chunk);
        ang = ang_local.data();
      }
      if (rotate_features) {
        hipLaunchKernelGGL(orb_moments_kernel, dim3((u32)((n_chunk + 3) / 4)), dim3(256), 0, c->stream, d_img, width, height,
                           (u32)width, (const double2*)(ds + o_xy), (u32)n_chunk, (int2*)(dd + o_mom),
                           (const u32*)(ds + o_ki), n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(hd + o_mom, dd + o_mom, n_chunk * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        lap("moments");
        const int32_t* mom = (const int32_t*)(hd + o_mom);
        constexpr size_t SLAB = 4096;
        parallel_items((n_chunk + SLAB - 1) / SLAB, 1, [&](size_t sl) {
          for (size_t i = sl * SLAB; i < std::min(n_chunk, (sl + 1) * SLAB); i++) {
            ang[i] = std::atan2((double)mom[2 * i + 1], (double)mom[2 * i]);  // keypoints.h:195
            cs[2 * i] = std::cos(ang[i]);                                     // keypoints.h:219
            cs[2 * i + 1] = std::sin(ang[i]);
          }
        });
      } else {
        for (size_t i = 0; i < n_chunk; i++) {
          ang[i] = 0;  // keypoints.h:161
          cs[2 * i] = std::cos(0.0);
          cs[2 * i + 1] = std::sin(0.0);
        }
      }
      lap("atan2, cos, sin");
      HIP_TRY(hipMemcpyAsync(ds + o_cs, h + o_cs, n_chunk * 16, hipMemcpyHostToDevice, c->stream));
      u64* d_desc = keep_as_database ? (u64*)((char*)c->db_owned.p + total * DESC_BYTES) : (u64*)(dd + o_desc);
      hipLaunchKernelGGL(orb_descriptor_kernel, dim3((u32)n_chunk), dim3(256), 0, c->stream, d_img, width, height, (u32)width,
                         (const double2*)(ds + o_xy), (const double2*)(ds + o_cs), (u32)n_chunk, d_desc,
                         (const u32*)(ds + o_ki), n);
      HIP_TRY(hipGetLastError());
      if (write_out && desc_out)
        HIP_TRY(hipMemcpyAsync(hd + o_desc, d_desc, n_chunk * DESC_BYTES, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (write_out && desc_out) memcpy(desc_out + 4 * total, hd + o_desc, n_chunk * DESC_BYTES);
      if (write_out && corners_out) memcpy(corners_out + 2 * total, h + o_xy, n_chunk * 16);
      lap("descriptors, copies out");
    }
    total += n_chunk;
  }
  if (overflow) return fail(VSM_ERR_CAPACITY, "%zu keypoints need more than cap=%zu (out_offsets holds the counts)", total, cap);
  if (keep_as_database) return set_db(c, c->db_owned.p, out_offsets, n_img);
  return VSM_OK;
 });
}

}  // extern "C"
