// omc_api.cpp -- host side of libomc_hip.so: the C ABI declared in include/omc.h.
//
// Mirrors the call sites of the reference driver (OMC.jl = /root/reference/src/OptimalMatrixCompletion.jl):
//   omc_relax_*            <- matrix_completion_SDP_relaxation  OMC.jl:1431-1943 (called at 747-754)
//   omc_separation_batch   <- eigs(U U' - Y) at OMC.jl:1274 and 2466-2477
//   omc_evaluate_objective <- evaluate_objective OMC.jl:2330-2359
//   omc_altmin_batch       <- alternating_minimization OMC.jl:1979-2279
// Argument checks reproduce the reference's `error(...)` conditions and return negative codes instead of
// throwing.  No torch types, no host pointer retained after return.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <chrono>
#include <string>
#include <vector>
#include <algorithm>
#include <atomic>
#include <thread>
#include <mutex>

#include "omc.h"
#include "omc_device.h"
#include "omc_altmin.h"
#include "omc_shor.h"
#include "omc_shor_relax.h"
#include <unordered_map>
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <rccl/rccl.h>   // types only: the library is loaded with dlopen on first use, so that libomc_hip.so has no hard dependency on it

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                                         \
  do {                                                                                                    \
    hipError_t e_ = (x);                                                                                  \
    if (e_ != hipSuccess) {                                                                               \
      g_err = std::string(#x) + ": " + hipGetErrorString(e_);                                             \
      return (int)e_ > 0 ? (int)e_ : 999;                                                                 \
    }                                                                                                     \
  } while (0)

struct DevBuf {
  void* p = nullptr; size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { g_err = std::string("hipMalloc: ") + hipGetErrorString(e); return (int)e; }
    cap = want;
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() { return (T*)p; }
};

// Tuning knobs.  The environment is read in ONE place -- tuning_from_env, called by omc_instance_create and by omc_tuning_reload_env -- and a
// caller that does not want its environment to matter sets the knobs through omc_tuning_set; nothing else in the library calls getenv
// (behaviour of the shipped library depended on the caller's environment at every solve, one knob even per iteration).
static const char* const OMC_TUNING_KEYS[] = {
  "OMC_ALTMIN_NOLDS",
  "OMC_CERT_SUB",
  "OMC_COLD_CHECK",
  "OMC_COLPROX_FIRST",
  "OMC_COLPROX_KEEPB",
  "OMC_CONE_512",
  "OMC_CP_MAXPASS",
  "OMC_CP_SERIES",
  "OMC_DEBUG_MAX_SWEEPS",
  "OMC_DENSE_CHECK",
  "OMC_GLOBAL_NOLDS",
  "OMC_GRAPH_MAX",
  "OMC_GRAPH_TAILS",
  "OMC_GRAPH_SHARED_EVENTS",
  "OMC_GROUPS",
  "OMC_JACOBI_TAU",
  "OMC_NO_COLPROX_PAIR",
  "OMC_NO_COLPROX_WIDE",
  "OMC_NO_GRAPH",
  "OMC_NO_SEP_SUB",
  "OMC_NO_SLOT_LIST",
  "OMC_NO_SUBSPACE",
  "OMC_NO_WARMSTART",
  "OMC_NO_WS_SPLIT",
  "OMC_NO_YX",
  "OMC_SEGV_TRACE",
  "OMC_REFILL_EVERY",
  "OMC_SHOR_DEBUG",
  "OMC_SHOR_EXPLICIT",
  "OMC_SHOR_NO_SUBSPACE",
  "OMC_SLOTS",
  "OMC_SMALL_COLD",
  "OMC_STREAMS",
  "OMC_SUB_ADAPT",
  "OMC_SUB_CHUNK",
  "OMC_SUB_DEBUG",
  "OMC_SUB_GUARD",
  "OMC_SUB_LAZY",
  "OMC_SUB_QMAX",
  "OMC_SUB_TOL",
  "OMC_TIMING_STRIDE",
};
struct Tuning {
  std::unordered_map<std::string, std::string> kv;
  const char* get(const char* name) const { auto it = kv.find(name); return it == kv.end() ? nullptr : it->second.c_str(); }
};
// diagnostics (OMC_SEGV_TRACE=1 at instance creation): print the native frames of a crashing thread before the default action runs
static void omc_segv_trace(int sig) {
  void* fr[64];
  const int nf = backtrace(fr, 64);
  const char msg[] = "libomc_hip: fatal signal, native frames:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(fr, nf, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
static void tuning_from_env(Tuning& t) {
  t.kv.clear();
  for (const char* key : OMC_TUNING_KEYS) if (const char* v = getenv(key)) t.kv[key] = v;
}

struct omc_instance {
  int n = 0, m = 0, k = 0, device = 0, nnz = 0, cmax = 0;
  double gamma = 0, sumA2 = 0;
  std::vector<int> col_ptr, col_idx; std::vector<double> col_val;
  std::vector<double> A; std::vector<uint8_t> mask;
  std::vector<double> Ncnt;
  std::vector<int> row_ptr, row_idx; std::vector<double> row_val;
  DevBuf drow_ptr, drow_idx, drow_val, aR, arkind, arcut, arbi, arbj, arcoef, arrhs, acutx, aU0, aU, aV, aobj, aint, aG, aG2;
  hipStream_t stream = nullptr;
  // per slot group: main / column / small-cone streams and fork, join, done events (see omc_relax_solve)
  Tuning tun;
  hipStream_t gs[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
  hipEvent_t gev[2][5] = {{nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr}};
  hipEvent_t gevc[2][5] = {{nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr}};      // the same fork / join events for bodies recorded under stream capture: an event is never used both inside a captured graph and eagerly
  hipEvent_t ev_main = nullptr;
  DevBuf dA, dmask, dcol_ptr, dcol_idx, dcol_val, dNcnt, dwY, dsolo, dwide;
  int nwide = 0;      // columns outside the pairs with at most 64 observed rows (k_colprox_wide)
  int nsolo = 0;      // columns that k_colprox_pair leaves to k_colprox (more than 32 observed rows, unpaired last column)
  // batch workspace
  DevBuf bYx, bY, bYp, bU, bD1, bD3, bW1, bE3, bQb, brr, bsm, bdS, balpha, balphaX, bsval, bMchk, bsmall, bchk;
  DevBuf bR, brkind, brcut, brbi, brbj, brcoef, brrhs, bcutx, bG, blam;
  DevBuf bobjcol, baaF, baaG, baaZ, baaS, baaI, bMbufC, bVrowC, bchkS, bchkI;
  DevBuf bslotlist, bgap, bvotes, blamDX, bXsC, bsubSC, bsubIC;
  DevBuf bsubz, bscal, bbx, bint, bcp, bcone, bglob, bXout, bThout, bXin, bMbuf, bVrow, bXs, bsubS, bsubI;
  long long sub_tot[8] = {0};
  int ws_lpp = 0, ws_use_lds = 0; size_t ws_lds = 0;
  OmcWS ws{};
  omc_relax_params params{};
  bool staged = false;
  int cone_use_lds = 0, glob_use_lds = 0, small_use_lds = 0; size_t cone_lds = 0, glob_lds = 0, small_lds = 0;
  double last_solve_seconds = 0; long long total_sweeps = 0; int last_iters_total = 0;
  std::thread worker; std::atomic<int> worker_running{0}, nodes_done{0}; int worker_rc = 0; std::string worker_err;
  // appending nodes to a staged / running batch (omc_relax_reserve, omc_relax_append): descriptor and output arrays are sized for node_cap nodes,
  // the strides (rows, row-subspace columns, cuts per node) for reserve_cuts cuts; Btot_live is what the solve loop reads at its refill points
  int reserve_nodes = 0, reserve_cuts = 0, node_cap = 0, staged_cut_type = 0; std::atomic<int> Btot_live{0}; bool append_closed = true; std::mutex append_mu; hipStream_t append_stream = nullptr;
  std::atomic<int> hold{0};      // omc_relax_hold: a solve that has run dry waits for omc_relax_append instead of ending
  std::vector<int> done_q; size_t done_read = 0; std::mutex done_mu; hipStream_t fetch_stream = nullptr;      // nodes harvested so far, in harvest order (omc_relax_fetch_done)
  void* comm = nullptr; int comm_rank = 0, comm_world = 1; DevBuf bcomm, amobj; int amobj_B = 0;
  std::vector<double> rho_scale_per_node; DevBuf brho, brhon, blamD, bslotint, boY, boU, boal, bobx, boscal, boint;
  int Btot = 0;
  // Shor minors (a10 / a11): row bitsets and per-pair popcounts, built at the first call
  DevBuf sbits, scb, scx, scz, soff, stot, sout, shi, slo, sexist, shist, sohi, solo, scnt;
  bool shor_ready = false; int shor_W = 0; long long shor_pairs = 0;
  double shor_last_ms = 0; long long shor_last_candidates = 0;
  // Shor-mode relaxation (omc_relax_stage_shor): index structures of the distinct lists, explicit X / W / Theta state, view of the workspace
  // through which the base eigen-kernels project the order-(n+m) cone
  bool shor_req = false, shor_on = false, shor_keep_V = false, shor_via_base = false; std::vector<int> shor_slackrow; DevBuf soV; double shor_rho = 0.05, shor_r4 = 0.0, shor_r5 = 2.0;      // r4 = 0: automatic per list
  ShWS sh{}; OmcWS wbig{}; int big_lpp = 0, big_use_lds = 0, big_cone_lds_ok = 0; size_t big_lds = 0, big_cone_lds = 0;
  DevBuf sgInts, sgBytes, sgGroups, sgNodeGroup, sAh, sX, sW, sTh, sV1, sV2, sV3, sD0, sP0, sMbufB, sVrowB, sTq, sPq, sNq, sD5x, sD5t, snu5, sP5x,
      scolpart, sminpart, sminpart2, sfroB, svvB, se1, se2, soX, soW, soTh, sbigscr, sXsB, ssubSB, ssubIB;
  long long big_sub_tot[8] = {0};
  // warm-start pool (omc_state_pool_create / omc_relax_set_warm)
  int pool_cap = 0; DevBuf pY, pD1, pD3, pU, palpha, psval, pXs, ptheta, pscal, bwarmL, bwarmS; std::vector<int> warm_load, warm_save;
  // kernel stats
  int64_t launches[OMC_KERNEL_NCLASS] = {0}; double ms[OMC_KERNEL_NCLASS] = {0}; int64_t units[OMC_KERNEL_NCLASS] = {0};
  size_t nnz_rows() const { return row_idx.size(); }
  std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
  std::vector<int> ev_class;
};

extern "C" {

const char* omc_last_error(void) { return g_err.c_str(); }
int omc_version(void) { return OMC_VERSION; }
int omc_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

void omc_relax_params_default(omc_relax_params* p) {
  p->eps_gap = 1e-6; p->eps_feas = 1e-7; p->max_iters = 3000; p->check_every = 25;
  p->rho_scale = 1.0; p->rho_f_ratio = 0.1; p->relax = 1.6; p->time_limit = 3600.0;
  p->reference_quirk_q1 = 1; p->breakpoints = OMC_SMALLEST_1_EIGVEC; p->stall_checks = 8;
  p->bump_max = 2; p->bump_ratio = 4.0; p->bump_factor = 4.0; p->bump_after = 100; p->bump_window = 4; p->slots = 0;
  p->accel = 0; p->aa_mem = 10; p->aa_every = 10; p->aa_start = 50; p->aa_reg = 1e-10; p->aa_safeguard = 1.0; p->first_wins = 0;
  p->early_stop_after = 400; p->early_stop_factor = 1.5;
}

static int upload(DevBuf& b, const void* src, size_t bytes, hipStream_t s) {
  int rc = b.ensure(bytes ? bytes : 8);
  if (rc) return rc;
  if (bytes) HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, s));
  return 0;
}

int omc_instance_create(int n, int m, int k, const double* A, const uint8_t* mask, double gamma, int device,
                        omc_instance** out) {
  if (!out) return fail(OMC_ERR_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (!A || !mask) return fail(OMC_ERR_ARGUMENT, "A / mask is NULL");
  if (n <= 0 || m <= 0) return fail(OMC_ERR_DIMENSION, "Dimension mismatch: A must have size (n, m) (OMC.jl:240-246)");
  if (!(n <= m)) return fail(OMC_ERR_DIMENSION, "Input matrix A must have size (n, m) with n <= m (OMC.jl:249-254)");
  if (k < 1 || k > n) return fail(OMC_ERR_ARGUMENT, "rank k must satisfy 1 <= k <= n");
  if (k > 8) return fail(OMC_ERR_UNSUPPORTED, "k > 8 is not supported by this build");
  if (!(gamma > 0.0)) return fail(OMC_ERR_ARGUMENT, "gamma must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(OMC_ERR_NO_DEVICE, "no HIP device visible: the HIP path is the product, there is no CPU fallback");
  if (device < 0 || device >= ndev) return fail(OMC_ERR_ARGUMENT, "device index out of range");
  HIPCHK(hipSetDevice(device));
  omc_instance* h = new omc_instance();
  tuning_from_env(h->tun);
  if (h->tun.get("OMC_SEGV_TRACE")) { signal(SIGSEGV, omc_segv_trace); signal(SIGABRT, omc_segv_trace); }
  struct Guard { omc_instance*& p; ~Guard() { if (p) omc_instance_destroy(p); } } guard{h};     // every early return below frees the handle
  h->n = n; h->m = m; h->k = k; h->gamma = gamma; h->device = device;
  h->A.assign(A, A + (size_t)n * m);
  h->mask.resize((size_t)n * m);
  for (size_t e = 0; e < (size_t)n * m; ++e) h->mask[e] = mask[e] ? 1 : 0;
  h->col_ptr.assign(m + 1, 0);
  for (int j = 0; j < m; ++j) {
    int c = 0;
    for (int i = 0; i < n; ++i)
      if (h->mask[(size_t)j * n + i]) { h->col_idx.push_back(i); h->col_val.push_back(A[(size_t)j * n + i]); ++c; }
    h->col_ptr[j + 1] = h->col_ptr[j] + c;
    h->cmax = std::max(h->cmax, c);
  }
  h->nnz = h->col_ptr[m];
  h->sumA2 = 0;
  for (double v : h->col_val) h->sumA2 += v * v;
  h->Ncnt.assign((size_t)n * n, 0.0);
  for (int j = 0; j < m; ++j)
    for (int p = h->col_ptr[j]; p < h->col_ptr[j + 1]; ++p)
      for (int q = h->col_ptr[j]; q < h->col_ptr[j + 1]; ++q)
        h->Ncnt[(size_t)h->col_idx[q] * n + h->col_idx[p]] += 1.0;
  // CSR copy (rows -> observed columns) for the U-step of altmin
  h->row_ptr.assign(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    int c = 0;
    for (int j = 0; j < m; ++j)
      if (h->mask[(size_t)j * n + i]) { h->row_idx.push_back(j); h->row_val.push_back(A[(size_t)j * n + i]); ++c; }
    h->row_ptr[i + 1] = h->row_ptr[i] + c;
  }
  HIPCHK(hipStreamCreate(&h->stream));
  int rc = 0;
  if ((rc = upload(h->drow_ptr, h->row_ptr.data(), sizeof(int) * (n + 1), h->stream))) return rc;
  if ((rc = upload(h->drow_idx, h->row_idx.data(), sizeof(int) * h->nnz_rows(), h->stream))) return rc;
  if ((rc = upload(h->drow_val, h->row_val.data(), sizeof(double) * h->nnz_rows(), h->stream))) return rc;
  if ((rc = upload(h->dA, h->A.data(), sizeof(double) * n * m, h->stream))) return rc;
  if ((rc = upload(h->dmask, h->mask.data(), (size_t)n * m, h->stream))) return rc;
  if ((rc = upload(h->dcol_ptr, h->col_ptr.data(), sizeof(int) * (m + 1), h->stream))) return rc;
  {
    std::vector<int> solo, wide;
    const bool no_wide = h->tun.get("OMC_NO_COLPROX_WIDE") != nullptr;      // read at creation: the column lists are built here
    auto place = [&](int j) { const int c = h->col_ptr[j + 1] - h->col_ptr[j]; if (c == 0) return; if (c <= 64 && !no_wide) wide.push_back(j); else solo.push_back(j); };
    for (int j0 = 0; j0 < m; j0 += 2) {
      const int j1 = j0 + 1;
      const int c0 = h->col_ptr[j0 + 1] - h->col_ptr[j0], c1 = (j1 < m) ? h->col_ptr[j1 + 1] - h->col_ptr[j1] : 0;
      if (j1 < m && c0 <= 32 && c1 <= 32) continue;
      place(j0); if (j1 < m) place(j1);
    }
    h->nsolo = (int)solo.size(); h->nwide = (int)wide.size();
    if (h->nsolo && (rc = upload(h->dsolo, solo.data(), sizeof(int) * solo.size(), h->stream))) return rc;
    if (h->nwide && (rc = upload(h->dwide, wide.data(), sizeof(int) * wide.size(), h->stream))) return rc;
  }
  if ((rc = upload(h->dcol_idx, h->col_idx.data(), sizeof(int) * h->nnz, h->stream))) return rc;
  if ((rc = upload(h->dcol_val, h->col_val.data(), sizeof(double) * h->nnz, h->stream))) return rc;
  if ((rc = upload(h->dNcnt, h->Ncnt.data(), sizeof(double) * n * n, h->stream))) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  int lrc = omc_set_max_lds();
  if (lrc) return fail(lrc, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
  { int arc = omc_altmin_set_lds(); if (arc) return fail(5000 + arc, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for the altmin kernel"); }
  omc_relax_params_default(&h->params);
  *out = h;
  h = nullptr;          // released to the caller (the guard holds a reference to this pointer)
  return 0;
}

int omc_instance_create_bits(int n, int m, int k, const double* A, const uint64_t* chunks, double gamma, int device,
                             omc_instance** out) {
  if (!chunks) return fail(OMC_ERR_ARGUMENT, "chunks is NULL");
  if (n <= 0 || m <= 0) return fail(OMC_ERR_DIMENSION, "Dimension mismatch (OMC.jl:240-246)");
  std::vector<uint8_t> mask((size_t)n * m);
  for (size_t e = 0; e < (size_t)n * m; ++e) mask[e] = (uint8_t)((chunks[e >> 6] >> (e & 63)) & 1ull);
  return omc_instance_create(n, m, k, A, mask.data(), gamma, device, out);
}

void omc_instance_destroy(omc_instance* h) {
  if (!h) return;
  if (h->worker.joinable()) h->worker.join();
  (void)hipSetDevice(h->device);
  (void)omc_comm_destroy(h);
  h->bcomm.release(); h->amobj.release();
  DevBuf* all[] = {&h->dwide, &h->bYx, &h->dsolo, &h->dA, &h->dmask, &h->dcol_ptr, &h->dcol_idx, &h->dcol_val, &h->dNcnt, &h->dwY, &h->bY, &h->bYp, &h->bU,
                   &h->bD1, &h->bD3, &h->bW1, &h->bE3, &h->bQb, &h->brr, &h->bsm, &h->bdS, &h->bsmall, &h->bchk,
                   &h->balpha, &h->balphaX, &h->bsval, &h->bMchk,
                   &h->bR, &h->brkind, &h->brcut, &h->brbi, &h->brbj, &h->brcoef, &h->brrhs, &h->bcutx, &h->bG, &h->blam,
                   &h->bsubz, &h->bscal, &h->bbx, &h->bint, &h->bcp, &h->bcone, &h->bglob, &h->bXout, &h->bThout, &h->bXin, &h->bMbuf, &h->bVrow, &h->bXs, &h->bsubS, &h->bsubI, &h->bslotlist, &h->bgap, &h->bvotes, &h->blamDX, &h->bXsC, &h->bsubSC, &h->bsubIC,
                   &h->brho, &h->brhon, &h->blamD, &h->bslotint, &h->boY, &h->boU, &h->boal, &h->bobx, &h->boscal, &h->boint, &h->drow_ptr, &h->drow_idx, &h->drow_val, &h->aR, &h->arkind, &h->arcut, &h->arbi, &h->arbj, &h->arcoef, &h->arrhs, &h->acutx,
                   &h->aU0, &h->aU, &h->aV, &h->aobj, &h->aint, &h->aG, &h->aG2,
                   &h->bobjcol, &h->baaF, &h->baaG, &h->baaZ, &h->baaS, &h->baaI, &h->bMbufC, &h->bVrowC, &h->bchkS, &h->bchkI, &h->sbits, &h->scb, &h->scx, &h->scz, &h->soff, &h->stot, &h->sout, &h->shi, &h->slo, &h->sexist, &h->shist, &h->sohi, &h->solo, &h->scnt,
                   &h->sgInts, &h->sgBytes, &h->sgGroups, &h->sgNodeGroup, &h->sAh, &h->sX, &h->sW, &h->sTh, &h->sV1, &h->sV2, &h->sV3, &h->sD0, &h->sP0, &h->sMbufB, &h->sVrowB,
                   &h->sTq, &h->sPq, &h->sNq, &h->sD5x, &h->sD5t, &h->snu5, &h->sP5x, &h->scolpart, &h->sminpart, &h->sminpart2, &h->sfroB, &h->svvB, &h->se1, &h->se2,
                   &h->soX, &h->soW, &h->soTh, &h->sbigscr, &h->soV, &h->sXsB, &h->ssubSB, &h->ssubIB,
                   &h->pY, &h->pD1, &h->pD3, &h->pU, &h->palpha, &h->psval, &h->pXs, &h->ptheta, &h->pscal, &h->bwarmL, &h->bwarmS};
  for (DevBuf* b : all) b->release();
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  for (int g = 0; g < 2; ++g) {
    for (int q = 0; q < 4; ++q) if (h->gs[g][q]) (void)hipStreamDestroy(h->gs[g][q]);
    for (int q = 0; q < 5; ++q) { if (h->gev[g][q]) (void)hipEventDestroy(h->gev[g][q]); if (h->gevc[g][q]) (void)hipEventDestroy(h->gevc[g][q]); }
  }
  if (h->ev_main) (void)hipEventDestroy(h->ev_main);
  if (h->append_stream) (void)hipStreamDestroy(h->append_stream);
  if (h->fetch_stream) (void)hipStreamDestroy(h->fetch_stream);
  delete h;
}

// piece table of the disjunctive cuts (OMC.jl:1580-1678): lo <= v <= hi, g(v) = slope*v + icpt
static int cut_piece(int cut_type, int dir, double vhat, int q1, double* lo, double* hi, double* slope, double* icpt) {
  const double a = fabs(vhat);
  if (cut_type == OMC_CUT_LINEAR) {
    if (dir == OMC_DIR_LEFT) { *lo = -1; *hi = vhat; *slope = vhat - 1; *icpt = vhat; return 0; }      // 1582-1591
    if (dir == OMC_DIR_RIGHT) { *lo = vhat; *hi = 1; *slope = vhat + 1; *icpt = -vhat; return 0; }     // 1592-1601
  } else if (cut_type == OMC_CUT_LINEAR2) {
    if (dir == OMC_DIR_LEFT) { *lo = -1; *hi = -a; *slope = -(1 + a); *icpt = -a; return 0; }          // 1604-1613
    if (dir == OMC_DIR_MIDDLE) { *lo = -a; *hi = a; *slope = 0; *icpt = vhat * vhat; return 0; }       // 1614-1623
    if (dir == OMC_DIR_RIGHT) { *lo = a; *hi = 1; *slope = 1 + a; *icpt = -a; return 0; }              // 1624-1633
  } else if (cut_type == OMC_CUT_LINEAR3) {
    if (dir == OMC_DIR_LEFT) { *lo = -1; *hi = -a; *slope = -(1 + a); *icpt = -a; return 0; }          // 1636-1645
    if (dir == OMC_DIR_INNER_LEFT) { *lo = -a; *hi = 0; *slope = -a; *icpt = 0; return 0; }            // 1646-1655
    if (dir == OMC_DIR_INNER_RIGHT) { *lo = 0; *hi = a; *slope = a; *icpt = 0; return 0; }             // 1656-1665
    if (dir == OMC_DIR_RIGHT) {                                                                        // 1666-1675
      *lo = a; *hi = 1;
      if (q1) { *slope = a; *icpt = 0; } else { *slope = 1 + a; *icpt = -a; }
      return 0;
    }
  }
  return -1;
}

static hipEvent_t next_event(omc_instance* h) {
  if (h->ev_used == h->ev_pool.size()) {
    hipEvent_t e; (void)hipEventCreate(&e); h->ev_pool.push_back(e);
  }
  return h->ev_pool[h->ev_used++];
}
#define TIMED_ON(strm, cls, units_, call)             \
  do {                                               \
    hipEvent_t e0_ = next_event(h), e1_ = next_event(h); \
    (void)hipEventRecord(e0_, strm);                 \
    call;                                            \
    (void)hipEventRecord(e1_, strm);                 \
    h->ev_class.push_back(cls);                      \
    h->launches[cls] += 1; h->units[cls] += (units_); \
  } while (0)
#define TIMED(cls, units_, call) TIMED_ON(h->stream, cls, units_, call)

// rows (OMC.jl:1558-1685) and row subspace of a set of nodes, on the host: shared by omc_relax_stage and omc_relax_append
struct NodePack {
  std::vector<std::vector<int>> rk, rc, rbi, rbj;
  std::vector<std::vector<double>> rcoef, rrhs, Qn;
  std::vector<int> rrv;
  int Rmax = 0, rmax = 1, Lmax = 0;
};
static int pack_nodes(omc_instance* h, const omc_relax_params& P, int cut_type, int B, const int* L, const double* cut_x, const double* cut_Uhat,
                      const int8_t* cut_dir, const double* U_lower, const double* U_upper, NodePack& pk) {
  const int n = h->n, k = h->k;
  // ---- rows (host) -----------------------------------------------------------------------------------
  int Lmax = 0; long Ltot = 0;
  for (int b = 0; b < B; ++b) {
    int Lb = L ? L[b] : 0;
    if (Lb < 0) return fail(OMC_ERR_ARGUMENT, "negative cut count");
    Lmax = std::max(Lmax, Lb); Ltot += Lb;
  }
  if (Ltot > 0 && (!cut_x || !cut_Uhat || !cut_dir)) return fail(OMC_ERR_ARGUMENT, "cut arrays are NULL but L > 0");
  // box rows: entries whose bound is not implied by ||U_j|| <= 1 (OMC.jl:1561, defaults 1442-1449)
  auto& rk = pk.rk; auto& rc = pk.rc; auto& rbi = pk.rbi; auto& rbj = pk.rbj; auto& rcoef = pk.rcoef; auto& rrhs = pk.rrhs;
  rk.assign(B, {}); rc.assign(B, {}); rbi.assign(B, {}); rbj.assign(B, {}); rcoef.assign(B, {}); rrhs.assign(B, {});
  int Rmax = 0;
  long cutbase = 0;
  for (int b = 0; b < B; ++b) {
    auto add = [&](int kind, int cut, int bi, int bj, const double* coef, double rhs) {
      rk[b].push_back(kind); rc[b].push_back(cut); rbi[b].push_back(bi); rbj[b].push_back(bj);
      for (int j = 0; j < k; ++j) rcoef[b].push_back(coef ? coef[j] : 0.0);
      rrhs[b].push_back(rhs);
    };
    add(ROW_TRACE, -1, -1, -1, nullptr, (double)k);                                 // OMC.jl:1558
    for (int j = 0; j < k; ++j)
      for (int i = 0; i < n; ++i) {
        double lo = U_lower ? U_lower[(size_t)b * n * k + (size_t)j * n + i] : ((i >= n - k + j) ? 0.0 : -1.0);
        double hi = U_upper ? U_upper[(size_t)b * n * k + (size_t)j * n + i] : 1.0;
        if (lo > hi) return fail(OMC_ERR_ARGUMENT, "U_lower > U_upper");
        double cf[8] = {0};
        if (lo > -1.0) { cf[0] = -1.0; add(ROW_BOX, -1, i, j, cf, -lo); }
        if (hi < 1.0) { cf[0] = 1.0; add(ROW_BOX, -1, i, j, cf, hi); }
      }
    const int Lb = L ? L[b] : 0;
    for (int l = 0; l < Lb; ++l) {
      const double* x = cut_x + (size_t)(cutbase + l) * n;
      const double* Uh = cut_Uhat + (size_t)(cutbase + l) * n * k;
      const int8_t* dr = cut_dir + (size_t)(cutbase + l) * k;
      double slope[8], icsum = 0.0;
      for (int j = 0; j < k; ++j) {
        double vhat = 0.0;                                                         // OMC.jl:1577
        for (int i = 0; i < n; ++i) vhat += Uh[(size_t)j * n + i] * x[i];
        double lo, hi, sl, ic;
        if (cut_piece(cut_type, dr[j], vhat, P.reference_quirk_q1, &lo, &hi, &sl, &ic))
          return fail(OMC_ERR_INVALID_ENUM, "direction code invalid for this cut type (OMC.jl:1580-1678)");
        slope[j] = -sl; icsum += ic;
        double cf[8] = {0};
        cf[j] = 1.0; add(ROW_BOUND, l, -1, j, cf, hi);                              // v_j <= hi
        cf[j] = -1.0; add(ROW_BOUND, l, -1, j, cf, -lo);                            // -v_j <= -lo
      }
      add(ROW_CUT, l, -1, -1, slope, icsum);                                        // OMC.jl:1680-1683
    }
    cutbase += Lb;
    Rmax = std::max(Rmax, (int)rk[b].size());
  }
  // ---- subspace of the U functionals: modified Gram-Schmidt (twice) over the row vectors in row order ------
  auto& Qn = pk.Qn; auto& rrv = pk.rrv;
  Qn.assign(B, {}); rrv.assign(B, 0);
  int rmax = 1;
  cutbase = 0;
  for (int b = 0; b < B; ++b) {
    std::vector<double>& Qv = Qn[b];
    int r = 0;
    std::vector<double> v(n);
    const int Lb = L ? L[b] : 0;
    for (size_t rw = 0; rw < rk[b].size(); ++rw) {
      const int kind = rk[b][rw];
      if (kind == ROW_TRACE) continue;
      for (int j = 0; j < k; ++j) {
        const double cf = rcoef[b][rw * k + j];
        bool nz = false;
        if (kind == ROW_BOX) {
          if (j != rbj[b][rw]) continue;
          std::fill(v.begin(), v.end(), 0.0); v[rbi[b][rw]] = rcoef[b][rw * k] > 0 ? 1.0 : -1.0; nz = true;
        } else if (cf != 0.0) {
          const double* x = cut_x + (size_t)(cutbase + rc[b][rw]) * n;
          double nx = 0.0;
          for (int i = 0; i < n; ++i) nx += x[i] * x[i];
          nx = sqrt(nx) * fabs(cf);
          if (nx > 0) { for (int i = 0; i < n; ++i) v[i] = cf * x[i] / nx; nz = true; }
        }
        if (!nz) continue;
        for (int pass = 0; pass < 2; ++pass)
          for (int a = 0; a < r; ++a) {
            double d = 0.0;
            for (int i = 0; i < n; ++i) d += Qv[(size_t)a * n + i] * v[i];
            for (int i = 0; i < n; ++i) v[i] -= d * Qv[(size_t)a * n + i];
          }
        double nv = 0.0;
        for (int i = 0; i < n; ++i) nv += v[i] * v[i];
        nv = sqrt(nv);
        if (nv > 1e-10 && r < n) {
          Qv.resize((size_t)(r + 1) * n);
          for (int i = 0; i < n; ++i) Qv[(size_t)r * n + i] = v[i] / nv;
          ++r;
        }
      }
    }
    rrv[b] = r; rmax = std::max(rmax, r);
    cutbase += Lb;
  }
  pk.Rmax = Rmax; pk.rmax = rmax; pk.Lmax = Lmax;
  return 0;
}

int omc_relax_stage(omc_instance* h, int B, const omc_relax_params* params, int cut_type, const int* L,
                    const double* cut_x, const double* cut_Uhat, const int8_t* cut_dir, const double* U_lower,
                    const double* U_upper) {
  if (!h) return fail(OMC_ERR_ARGUMENT, "handle is NULL");
  if (B <= 0) return fail(OMC_ERR_ARGUMENT, "B must be positive");
  if (cut_type != OMC_CUT_LINEAR && cut_type != OMC_CUT_LINEAR2 && cut_type != OMC_CUT_LINEAR3)
    return fail(OMC_ERR_INVALID_ENUM, "Invalid input for disjunctive cuts type: must be linear, linear2 or linear3 (OMC.jl:1456-1462)");
  HIPCHK(hipSetDevice(h->device));
  h->staged = false;
  const bool shor = h->shor_req;      // set by omc_relax_stage_shor for this call only
  h->shor_req = false; h->shor_on = false; h->shor_via_base = false;
  if (params) h->params = *params; else omc_relax_params_default(&h->params);
  if (shor) {      // the bound of a Shor node lags its primal value for the first ~1000 iterations: no early stop, a longer stall window, one bump
    h->params.early_stop_factor = 0.0;
    h->params.stall_checks = std::max(h->params.stall_checks, 40);
    h->params.bump_max = std::min(h->params.bump_max, 1);
    h->params.first_wins = 0; h->params.accel = 0;
  }
  const omc_relax_params& P = h->params;
  if (P.breakpoints != OMC_SMALLEST_1_EIGVEC && P.breakpoints != OMC_SMALLEST_2_EIGVEC)
    return fail(OMC_ERR_INVALID_ENUM, "Invalid input for disjunctive cuts breakpoints (OMC.jl:2440-2446)");
  const int n = h->n, m = h->m, k = h->k;
  // ---- rows and row subspaces (host) ---------------------------------------------------------------------
  NodePack pk;
  { int rcp = pack_nodes(h, P, cut_type, B, L, cut_x, cut_Uhat, cut_dir, U_lower, U_upper, pk); if (rcp) return rcp; }
  auto& rk = pk.rk; auto& rc = pk.rc; auto& rbi = pk.rbi; auto& rbj = pk.rbj; auto& rcoef = pk.rcoef; auto& rrhs = pk.rrhs; auto& Qn = pk.Qn; auto& rrv = pk.rrv;
  int Rmax = pk.Rmax, rmax = pk.rmax, Lmax = pk.Lmax;
  long cutbase = 0;
  // omc_relax_reserve: room for nodes appended later (default U bounds, at most reserve_cuts cuts each)
  const int extra_nodes = shor ? 0 : h->reserve_nodes, extra_cuts = shor ? 0 : h->reserve_cuts;
  h->reserve_nodes = 0; h->reserve_cuts = 0;
  if (extra_nodes > 0) {
    Lmax = std::max(Lmax, extra_cuts);
    Rmax = std::max(Rmax, 1 + k * (k + 1) / 2 + extra_cuts * (2 * k + 1));
    rmax = std::max(rmax, std::min(n, k + extra_cuts));
  }
  // ---- workspace -------------------------------------------------------------------------------------
  OmcWS& w = h->ws;
  memset(&w, 0, sizeof(w));
  // continuous batching: S slots relax B nodes; a slot that finishes is harvested and re-used for the next pending node
  const int Bcap = B + extra_nodes;      // with omc_relax_reserve the slots are sized for the nodes that may still come
  int S = (P.slots > 0) ? std::min(P.slots, Bcap) : std::min(Bcap, 256);
  if (h->tun.get("OMC_SLOTS")) S = std::max(1, std::min(Bcap, atoi(h->tun.get("OMC_SLOTS"))));
  h->Btot = B; h->Btot_live.store(B); h->node_cap = B + extra_nodes; h->staged_cut_type = cut_type; { std::lock_guard<std::mutex> lk(h->append_mu); h->append_closed = false; }
  { std::lock_guard<std::mutex> lk(h->done_mu); h->done_q.clear(); h->done_read = 0; }
  h->hold.store(0);
  w.b0 = 0; w.nB = S;
  w.B = S; w.Btot = B; w.max_iters = P.max_iters; w.n = n; w.m = m; w.k = k; w.nnz = h->nnz; w.Rmax = Rmax; w.Lmax = std::max(Lmax, 1); w.rmax = rmax;
  w.jacobi_tau = h->tun.get("OMC_JACOBI_TAU") ? atof(h->tun.get("OMC_JACOBI_TAU")) : 0.0;
  w.max_sweeps = h->tun.get("OMC_DEBUG_MAX_SWEEPS") ? atoi(h->tun.get("OMC_DEBUG_MAX_SWEEPS")) : 30;
  w.breakpoints = P.breakpoints; w.stall_checks = P.stall_checks > 0 ? P.stall_checks : 1000000;
  w.gamma = h->gamma; w.sumA2 = h->sumA2;
  {
    const double sh = 1.0 + h->gamma * (double)k / (double)n;
    w.rho = P.rho_scale * 0.5 * h->gamma * h->sumA2 / ((double)m * sh * sh);
  }
  if (shor) w.rho = h->shor_rho * P.rho_scale;      // penalty of the cone blocks in the scaled variables of the Shor splitting
  if (!(w.rho > 0.0)) w.rho = 1.0;
  w.clip_hi = 1.0; w.inv_s2 = 1.0; w.shor = 0;
  w.rho_f_ratio = P.rho_f_ratio;
  w.bump_max = P.bump_max; w.bump_ratio = P.bump_ratio; w.bump_factor = P.bump_factor; w.bump_after = P.bump_after;
  w.bump_gap = P.bump_window * std::max(1, P.check_every);
  w.accel = P.accel ? 1 : 0; w.aa_mem = std::max(2, std::min(P.aa_mem, AA_MAXMEM)); w.aa_every = std::max(1, P.aa_every);
  w.aa_start = std::max(2, P.aa_start); w.aa_reg = P.aa_reg; w.aa_safeguard = P.aa_safeguard;
  {
    std::vector<double> hr((size_t)B + extra_nodes, w.rho);
    if (h->rho_scale_per_node.size() == (size_t)B)
      for (int b = 0; b < B; ++b) hr[b] = w.rho / P.rho_scale * h->rho_scale_per_node[b];
    h->rho_scale_per_node.clear();
    int r0 = upload(h->brhon, hr.data(), sizeof(double) * hr.size(), h->stream); if (r0) return r0;
    HIPCHK(hipStreamSynchronize(h->stream));          // hr dies with this block: the copy must have read it
    w.rho_node = h->brhon.as<double>();
    r0 = h->brho.ensure(sizeof(double) * (size_t)S); if (r0) return r0;
    w.rho_b = h->brho.as<double>();
  }
  w.relax = P.relax; w.eps_gap = P.eps_gap; w.eps_feas = P.eps_feas;
  w.check_every = std::max(1, P.check_every); w.early_stop_after = P.early_stop_after; w.early_stop_factor = (P.first_wins ? 0.0 : P.early_stop_factor);
  {
    int r0 = h->bgap.ensure(sizeof(double) * 2 * (size_t)S); if (r0) return r0;
    r0 = h->bvotes.ensure(sizeof(int) * (size_t)S); if (r0) return r0;
    r0 = h->bslotlist.ensure(sizeof(int) * (size_t)S); if (r0) return r0;
    w.gap_prev = h->bgap.as<double>(); w.gap_rate = h->bgap.as<double>() + S; w.slow_votes = h->bvotes.as<int>();
    w.slot_list = nullptr;      // set on the per-iteration copies only (omc_relax_solve)
  }
  std::vector<double> wY((size_t)n * n);
  for (size_t e = 0; e < (size_t)n * n; ++e) wY[e] = shor ? 3.0 : P.rho_f_ratio * h->Ncnt[e] + 2.0;      // Shor mode: big cone, clip, small cone
  int rc_ = 0;
  if ((rc_ = upload(h->dwY, wY.data(), sizeof(double) * n * n, h->stream))) return rc_;
  w.col_ptr = h->dcol_ptr.as<int>(); w.col_idx = h->dcol_idx.as<int>(); w.col_val = h->dcol_val.as<double>();
  w.cp_pair = h->tun.get("OMC_NO_COLPROX_PAIR") ? 0 : 1; w.cone_512 = h->tun.get("OMC_CONE_512") ? 1 : 0; w.cp_series = h->tun.get("OMC_CP_SERIES") ? atoi(h->tun.get("OMC_CP_SERIES")) : 6; w.cp_maxpass = h->tun.get("OMC_CP_MAXPASS") ? atoi(h->tun.get("OMC_CP_MAXPASS")) : 60; w.cp_nsolo = h->nsolo; w.cp_solo = h->nsolo ? h->dsolo.as<int>() : nullptr;
  w.cp_nwide = h->nwide; w.cp_wide = h->nwide ? h->dwide.as<int>() : nullptr;
  w.Ncnt = h->dNcnt.as<double>(); w.wY1 = h->dwY.as<double>();
  w.row_ptr = h->drow_ptr.as<int>(); w.row_idx = h->drow_idx.as<int>();
#define ENS(buf, bytes) do { int r_ = (buf).ensure(bytes); if (r_) return r_; } while (0)
  const size_t sB = (size_t)S;      // state arrays: one per slot
  const size_t sN = (size_t)B + (size_t)extra_nodes;      // descriptor and output arrays: one per node (plus the room reserved for appended nodes)
  ENS(h->bY, sB * n * n * 8); ENS(h->bYp, sB * n * n * 8); ENS(h->bU, sB * n * k * 8);
  ENS(h->bD1, sB * n * n * 8); ENS(h->bD3, sB * n * n * 8); ENS(h->bW1, sB * n * n * 8); ENS(h->bE3, sB * n * n * 8);
  ENS(h->bdS, sB * rmax * rmax * 8);
  ENS(h->bsm, sB * ((size_t)4 * rmax * k + 3 * k * k) * 8);
  ENS(h->balpha, sB * h->nnz * 8); ENS(h->balphaX, sB * h->nnz * 8); ENS(h->bsval, sB * m * 8);
  ENS(h->bMchk, sB * n * n * 8); ENS(h->bchk, sB * n * k * 8 + (32 + 8 * sB) * 8);
  ENS(h->bG, sB * Rmax * Rmax * 8); ENS(h->blam, sB * Rmax * 8);
  ENS(h->bscal, sB * 16 * 8); ENS(h->bbx, sB * n * 8); ENS(h->bint, sB * 10 * sizeof(int));
  w.np16 = (n + 15) & ~15;
  ENS(h->bMbuf, sB * w.np16 * w.np16 * 8); ENS(h->bVrow, sB * w.np16 * w.np16 * 8);
  ENS(h->blamD, sB * m * n * 8);
  HIPCHK(hipMemsetAsync(h->blamD.p, 0, sB * m * n * 8, h->stream));
  w.lamD = h->blamD.as<double>();
  w.lamDX = nullptr;
  if (n > 144 || h->tun.get("OMC_DENSE_CHECK") || shor) {      // large orders: the certificate matrix takes Lambda Lambda' from one MFMA product
    ENS(h->blamDX, sB * m * n * 8);
    HIPCHK(hipMemsetAsync(h->blamDX.p, 0, sB * m * n * 8, h->stream));
    w.lamDX = h->blamDX.as<double>();
  }
  w.Mbuf = h->bMbuf.as<double>(); w.Vrow = h->bVrow.as<double>();
  {   // tracked subspace of the cone block (k_cone_sub)
    ENS(h->bXs, sB * w.np16 * 16 * 8); ENS(h->bsubS, sB * (17 + 256) * 8); ENS(h->bsubI, sB * 14 * sizeof(int));
    HIPCHK(hipMemsetAsync(h->bXs.p, 0, sB * w.np16 * 16 * 8, h->stream));
    HIPCHK(hipMemsetAsync(h->bsubS.p, 0, sB * (17 + 256) * 8, h->stream));
    HIPCHK(hipMemsetAsync(h->bsubI.p, 0, sB * 14 * sizeof(int), h->stream));
    w.Xs = h->bXs.as<double>(); w.sub_theta = h->bsubS.as<double>(); w.trM = h->bsubS.as<double>() + sB * 16;
    w.sub_on = h->bsubI.as<int>(); w.cone_done = h->bsubI.as<int>() + sB; w.sub_stat = h->bsubI.as<int>() + 2 * sB; w.sub_wait = h->bsubI.as<int>() + 10 * sB; w.sub_nfail = h->bsubI.as<int>() + 11 * sB;
    w.V3 = h->tun.get("OMC_SMALL_COLD") ? nullptr : h->bsubS.as<double>() + sB * 17; w.v3valid = h->bsubI.as<int>() + 12 * sB;
    w.ws_first = h->bsubI.as<int>() + 13 * sB; w.ws_phase = 0;
    w.sub_guard = h->tun.get("OMC_SUB_GUARD") ? atoi(h->tun.get("OMC_SUB_GUARD")) : 2;
    w.sub_qmax = h->tun.get("OMC_SUB_QMAX") ? atoi(h->tun.get("OMC_SUB_QMAX")) : 24;
    w.sub_chunk = h->tun.get("OMC_SUB_CHUNK") ? atoi(h->tun.get("OMC_SUB_CHUNK")) : 3;
    w.sub_lazy = h->tun.get("OMC_SUB_LAZY") ? atoi(h->tun.get("OMC_SUB_LAZY")) : 1;
    w.sub_tol = h->tun.get("OMC_SUB_TOL") ? atof(h->tun.get("OMC_SUB_TOL")) : 1e-10;
    w.sub_adapt = h->tun.get("OMC_SUB_ADAPT") ? atof(h->tun.get("OMC_SUB_ADAPT")) : 1e-3;
    w.sub_debug = h->tun.get("OMC_SUB_DEBUG") ? atoi(h->tun.get("OMC_SUB_DEBUG")) : 0;
    w.sub_enable = 0;     // decided below, once the cone kernel variant is known
    ENS(h->bXsC, sB * w.np16 * 16 * 8); ENS(h->bsubSC, sB * 18 * 8); ENS(h->bsubIC, sB * 3 * sizeof(int));
    HIPCHK(hipMemsetAsync(h->bXsC.p, 0, sB * w.np16 * 16 * 8, h->stream));
    HIPCHK(hipMemsetAsync(h->bsubSC.p, 0, sB * 18 * 8, h->stream));
    HIPCHK(hipMemsetAsync(h->bsubIC.p, 0, sB * 3 * sizeof(int), h->stream));
    w.XsC = h->bXsC.as<double>(); w.sub_thetaC = h->bsubSC.as<double>(); w.trMc = h->bsubSC.as<double>() + sB * 16; w.lb_est = h->bsubSC.as<double>() + sB * 17;
    w.sub_onC = h->bsubIC.as<int>(); w.confirm = h->bsubIC.as<int>() + sB; w.sep_done = nullptr;
    w.cert_enable = 0;
  }
  if (!h->tun.get("OMC_COLD_CHECK")) {   // warm-started eigenvalues for the certificate matrix
    ENS(h->bMbufC, sB * w.np16 * w.np16 * 8); ENS(h->bVrowC, sB * w.np16 * w.np16 * 8); ENS(h->bchkS, sB * 8); ENS(h->bchkI, sB * sizeof(int));
    HIPCHK(hipMemsetAsync(h->bMbufC.p, 0, sB * w.np16 * w.np16 * 8, h->stream));
    HIPCHK(hipMemsetAsync(h->bVrowC.p, 0, sB * w.np16 * w.np16 * 8, h->stream));   // its zero padding feeds the branch-free K loop of the MFMA GEMM: 0 x garbage could be NaN
    HIPCHK(hipMemsetAsync(h->bchkI.p, 0, sB * sizeof(int), h->stream));
    w.MbufC = h->bMbufC.as<double>(); w.VrowC = h->bVrowC.as<double>(); w.fro2c = h->bchkS.as<double>(); w.vvalidC = h->bchkI.as<int>();
  }
  w.Y = h->bY.as<double>(); w.Yp = h->bYp.as<double>(); w.U = h->bU.as<double>();
  w.Yx = nullptr;
  if (!shor && !P.accel && !h->tun.get("OMC_NO_YX")) { ENS(h->bYx, sB * n * n * 8); w.Yx = h->bYx.as<double>(); }      // 2 Y - Yp for the column gathers (k_aa rewrites Y and Yp behind k_global's back)
  w.D1 = h->bD1.as<double>(); w.D3 = h->bD3.as<double>(); w.W1 = h->bW1.as<double>(); w.E3 = h->bE3.as<double>();
  w.dS = h->bdS.as<double>();
  {
    double* sm = h->bsm.as<double>();
    const size_t vk = sB * rmax * k, tk = sB * k * k;
    w.Vt = sm; w.D3V = sm + vk; w.W3V = sm + 2 * vk; w.Q3V = sm + 3 * vk;
    w.D3T = sm + 4 * vk; w.W3T = sm + 4 * vk + tk; w.Q3T = sm + 4 * vk + 2 * tk;
  }
  w.alpha = h->balpha.as<double>(); w.alphaX = h->balphaX.as<double>(); w.sval = h->bsval.as<double>();
  ENS(h->bobjcol, sB * m * 2 * 8);
  w.objcol = h->bobjcol.as<double>(); w.c0col = w.objcol + sB * m;
  w.Mchk = h->bMchk.as<double>(); w.chk_scratch = h->bchk.as<double>(); w.stamps = h->bchk.as<double>() + sB * n * k; w.G = h->bG.as<double>(); w.lam = h->blam.as<double>();
  double* sc = h->bscal.as<double>();
  w.obj = sc; w.objout = sc + sB; w.lb = sc + 2 * sB; w.c0 = sc + 3 * sB; w.evsum = sc + 4 * sB; w.cpen = sc + 5 * sB;
  w.cst = sc + 6 * sB; w.rp = sc + 7 * sB; w.rd = sc + 8 * sB; w.lmin = sc + 9 * sB;  // lmin uses 2B (slots 9,10)
  w.objprev = sc + 11 * sB; w.lbprev = sc + 12 * sB; w.fro2 = sc + 13 * sB; w.bfac = sc + 14 * sB;
  w.bx = h->bbx.as<double>();
  if (w.accel) {
    w.aa_dim = 4 * n * n + 2 * rmax * k + k * k + h->nnz;
    const size_t ring = sB * (size_t)(w.aa_mem + 1) * w.aa_dim * 8;
    ENS(h->baaF, ring); ENS(h->baaG, ring); ENS(h->baaZ, sB * (size_t)w.aa_dim * 8); ENS(h->baaS, sB * 8); ENS(h->baaI, sB * 6 * sizeof(int));
    w.aa_F = h->baaF.as<double>(); w.aa_G = h->baaG.as<double>(); w.aa_zin = h->baaZ.as<double>(); w.aa_fn = h->baaS.as<double>();
    int* ai = h->baaI.as<int>();
    w.aa_hist = ai; w.aa_head = ai + sB; w.aa_pending = ai + 2 * sB; w.aa_valid = ai + 3 * sB; w.aa_nacc = ai + 4 * sB; w.aa_nrej = ai + 5 * sB;
    HIPCHK(hipMemsetAsync(ai, 0, sB * 6 * sizeof(int), h->stream));
  }
  int* ip = h->bint.as<int>();
  w.done = ip; w.status = ip + sB; w.iters = ip + 2 * sB; w.sweeps = ip + 3 * sB; w.stall = ip + 4 * sB; w.vvalid = ip + 5 * sB; w.nbump = ip + 6 * sB; w.lastbump = ip + 7 * sB; w.rowov = ip + 8 * sB;
  {
    ENS(h->bslotint, sB * 3 * sizeof(int));
    int* si = h->bslotint.as<int>();
    w.node_of = si; w.init = si + sB; w.fin = si + 2 * sB;
    std::vector<int> hs(sB * 3);
    for (size_t b = 0; b < sB; ++b) { hs[b] = (int)b; hs[sB + b] = 1; hs[2 * sB + b] = 1; }   // identity map, every slot initialises
    HIPCHK(hipMemcpyAsync(si, hs.data(), sizeof(int) * hs.size(), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));          // hs dies with this block
    ENS(h->boY, sN * n * n * 8); ENS(h->boU, sN * n * k * 8); ENS(h->boal, sN * h->nnz * 8); ENS(h->bobx, sN * n * 8);
    ENS(h->boscal, sN * 5 * 8); ENS(h->boint, sN * 2 * sizeof(int));
    w.oY = h->boY.as<double>(); w.oU = h->boU.as<double>(); w.oalphaX = h->boal.as<double>(); w.obx = h->bobx.as<double>();
    double* os = h->boscal.as<double>();
    w.oobj = os; w.olb = os + sN; w.olmin = os + 2 * sN; w.orho = os + 4 * sN;
    w.ostatus = h->boint.as<int>(); w.oiters = h->boint.as<int>() + sN;
  }
  // warm-start indices of this batch (consumed: they belong to this stage call only)
  w.load_from = nullptr; w.save_to = nullptr;
  if (h->pool_cap > 0 && !shor && extra_nodes > 0) {      // appended nodes may name pool entries: both index arrays exist, -1 where nothing was given
    if (h->warm_load.size() != (size_t)B) h->warm_load.assign(B, -1);
    if (h->warm_save.size() != (size_t)B) h->warm_save.assign(B, -1);
    h->warm_load.resize(sN, -1); h->warm_save.resize(sN, -1);
  } else if (extra_nodes == 0) {
    if (h->warm_load.size() != (size_t)B) h->warm_load.clear();
    if (h->warm_save.size() != (size_t)B) h->warm_save.clear();
  }
  if (h->pool_cap > 0 && !shor && (h->warm_load.size() == sN || h->warm_save.size() == sN)) {
    if (h->warm_load.size() == sN) { if ((rc_ = upload(h->bwarmL, h->warm_load.data(), sizeof(int) * sN, h->stream))) return rc_; w.load_from = h->bwarmL.as<int>(); }
    if (h->warm_save.size() == sN) { if ((rc_ = upload(h->bwarmS, h->warm_save.data(), sizeof(int) * sN, h->stream))) return rc_; w.save_to = h->bwarmS.as<int>(); }
    HIPCHK(hipStreamSynchronize(h->stream));
    w.pY = h->pY.as<double>(); w.pD1 = h->pD1.as<double>(); w.pD3 = h->pD3.as<double>(); w.pU = h->pU.as<double>(); w.palpha = h->palpha.as<double>();
    w.psval = h->psval.as<double>(); w.pXs = h->pXs.as<double>(); w.ptheta = h->ptheta.as<double>(); w.pscal = h->pscal.as<double>();
  }
  h->warm_load.clear(); h->warm_save.clear();
  // Q upload
  {
    std::vector<double> hQ(sN * n * rmax, 0.0);
    for (int b = 0; b < B; ++b) memcpy(&hQ[(size_t)b * n * rmax], Qn[b].data(), sizeof(double) * Qn[b].size());
    if ((rc_ = upload(h->bQb, hQ.data(), sizeof(double) * hQ.size(), h->stream))) return rc_;
    rrv.resize(sN, 0);
    if ((rc_ = upload(h->brr, rrv.data(), sizeof(int) * sN, h->stream))) return rc_;
    HIPCHK(hipStreamSynchronize(h->stream));          // hQ dies with this block
    w.Qb = h->bQb.as<double>(); w.rr = h->brr.as<int>();
  }
  // rows upload (padded to Rmax)
  std::vector<int> hR(sN, 0), hk(sN * Rmax, 0), hc(sN * Rmax, 0), hbi(sN * Rmax, 0), hbj(sN * Rmax, 0);
  std::vector<double> hcoef(sN * Rmax * k, 0.0), hrhs(sN * Rmax, 0.0), hx(sN * w.Lmax * n, 0.0);
  cutbase = 0;
  for (int b = 0; b < B; ++b) {
    hR[b] = (int)rk[b].size();
    for (int r = 0; r < hR[b]; ++r) {
      hk[(size_t)b * Rmax + r] = rk[b][r]; hc[(size_t)b * Rmax + r] = rc[b][r];
      hbi[(size_t)b * Rmax + r] = rbi[b][r]; hbj[(size_t)b * Rmax + r] = rbj[b][r];
      hrhs[(size_t)b * Rmax + r] = rrhs[b][r];
      for (int j = 0; j < k; ++j) hcoef[((size_t)b * Rmax + r) * k + j] = rcoef[b][(size_t)r * k + j];
    }
    const int Lb = L ? L[b] : 0;
    for (int l = 0; l < Lb; ++l)
      memcpy(&hx[((size_t)b * w.Lmax + l) * n], cut_x + (size_t)(cutbase + l) * n, sizeof(double) * n);
    cutbase += Lb;
  }
  if ((rc_ = upload(h->bR, hR.data(), sizeof(int) * sN, h->stream))) return rc_;
  if ((rc_ = upload(h->brkind, hk.data(), sizeof(int) * hk.size(), h->stream))) return rc_;
  if ((rc_ = upload(h->brcut, hc.data(), sizeof(int) * hc.size(), h->stream))) return rc_;
  if ((rc_ = upload(h->brbi, hbi.data(), sizeof(int) * hbi.size(), h->stream))) return rc_;
  if ((rc_ = upload(h->brbj, hbj.data(), sizeof(int) * hbj.size(), h->stream))) return rc_;
  if ((rc_ = upload(h->brcoef, hcoef.data(), sizeof(double) * hcoef.size(), h->stream))) return rc_;
  if ((rc_ = upload(h->brrhs, hrhs.data(), sizeof(double) * hrhs.size(), h->stream))) return rc_;
  if ((rc_ = upload(h->bcutx, hx.data(), sizeof(double) * hx.size(), h->stream))) return rc_;
  w.R = h->bR.as<int>(); w.rkind = h->brkind.as<int>(); w.rcut = h->brcut.as<int>(); w.rbi = h->brbi.as<int>();
  w.rbj = h->brbj.as<int>(); w.rcoef = h->brcoef.as<double>(); w.rrhs = h->brrhs.as<double>(); w.cutx = h->bcutx.as<double>();
  // LDS / scratch decisions
  {
    const int c_lds = std::min(h->cmax, 64);
    w.cp_lds_c = c_lds;
    w.cp_keepB = (c_lds <= 40 || h->tun.get("OMC_COLPROX_KEEPB")) ? 1 : 0;
    w.cp_lds_doubles = w.cp_keepB ? c_lds * c_lds + 5 * c_lds + 8 : c_lds * (c_lds + 1) / 2 + 4 * c_lds + 8;
    if ((size_t)4 * w.cp_lds_doubles * 8 > OMC_MAX_DYN_LDS) { w.cp_lds_c = 48; w.cp_lds_doubles = 48 * 48 + 5 * 48 + 8; w.cp_keepB = 1; }
    if (h->cmax > w.cp_lds_c) {
      w.cp_scratch_stride = (size_t)h->cmax * h->cmax + 5 * (size_t)h->cmax + 8;
      ENS(h->bcp, sB * m * w.cp_scratch_stride * 8);
      w.cp_scratch = h->bcp.as<double>();
    }
    auto cone_bytes = [](int Nn) { int Np = (Nn + 1) & ~1; int ld = Np | 1; return ((size_t)Np * ld + 2 * Np) * 8 + (size_t)Np * 4 + 16; };
    h->cone_lds = cone_bytes(n);
    h->cone_use_lds = h->cone_lds <= OMC_MAX_DYN_LDS;
    {
      // warm-started kernel: lanes per pair so that 512 threads cover the n/2 pairs, rows padded to lpp*rpl (<= 20 rows per lane).
      // G lives in LDS when it fits, else in the per-node global scratch (L2 resident) with 16 lanes per pair.
      const int Np2 = (n + 1) & ~1;
      int lpp = 16; while (lpp > 4 && lpp * (Np2 / 2) > 512) lpp >>= 1;
      int rpl = (((n + lpp - 1) / lpp) + 1) & ~1, Nrp = rpl * lpp;
      int ldw = Nrp + ((16 - (Nrp & 31)) & 31);                 // 16 (mod 32): neighbouring columns start 32 LDS banks apart
      if (((size_t)Np2 * ldw + 3 * Np2) * 8 + (size_t)(Np2 + 2) * 4 + 64 > OMC_MAX_DYN_LDS) ldw = Nrp + 2;   // does not fit: plain padding
      h->ws_lds = ((size_t)Np2 * ldw + 3 * Np2) * 8 + (size_t)(Np2 + 2) * 4 + 64;
      h->ws_use_lds = h->ws_lds <= OMC_MAX_DYN_LDS;
      w.ws_ld = ldw;
      if (!h->ws_use_lds) {
        lpp = 16; rpl = (((n + 15) / 16) + 1) & ~1; Nrp = rpl * 16; ldw = Nrp + 2; w.ws_ld = ldw;
        if (rpl > 32 && n <= 1024) { lpp = 64; rpl = (((n + 63) / 64) + 1) & ~1; Nrp = rpl * 64; ldw = Nrp + 2; w.ws_ld = ldw; }      // orders 513 .. 1024: a wave per pair
        const size_t need = ((size_t)Np2 * ldw + 3 * Np2) * 8 + (size_t)(Np2 + 2) * 4 + 64;
        if (need / 8 + 8 > w.cone_scratch_stride) {
          w.cone_scratch_stride = need / 8 + 8;
          ENS(h->bcone, sB * w.cone_scratch_stride * 8);
          w.cone_scratch = h->bcone.as<double>();
        }
      }
      h->ws_lpp = (rpl <= 32 && h->tun.get("OMC_NO_WARMSTART") == nullptr) ? lpp : 0;   // WS_JROWS
      // subspace tracking needs the warm-started kernel as its seed / fall-back and at least 3 x 16 rows
      w.sub_enable = (h->ws_lpp && n >= 48 && omc_cone_sub_lds(w.np16) <= OMC_MAX_DYN_LDS && !h->tun.get("OMC_NO_SUBSPACE")) ? 1 : 0;
      w.sub_zscratch = nullptr;
      if (w.sub_enable && w.np16 > 512) { ENS(h->bsubz, sB * 16 * (size_t)(w.np16 + 2) * 8); w.sub_zscratch = h->bsubz.as<double>(); }
      w.cert_enable = (w.sub_enable && w.MbufC && (h->tun.get("OMC_CERT_SUB") || w.np16 > 512)) ? 1 : 0;      // large orders: the certificate eigenvalues by the tracked block too (rigorous confirmation before anything is reported)
      w.sep_done = (w.sub_enable && !h->tun.get("OMC_NO_SEP_SUB")) ? h->bsubIC.as<int>() + 2 * sB : nullptr;   // opt-in: measured no gain (the eigenvalues are not what the check spends its time on) and fewer rigorous samples of the bound
    }
    h->glob_lds = ((size_t)n * (n + 1) / 2 + (size_t)n * k + (size_t)rmax * k + 3 * Rmax + 8 + 16 * (size_t)n) * 8 + 16;   // packed lower triangle of the target; 16 = GL_XS staged cut vectors
    h->glob_use_lds = (h->glob_lds + 20 * 1024 <= OMC_MAX_DYN_LDS) && !h->tun.get("OMC_GLOBAL_NOLDS");   // + the static LDS of k_global (NNQP scratch for NNQP_PMAX = 64 passive rows)
    if (!h->glob_use_lds) {
      w.glob_scratch_stride = h->glob_lds / 8 + 8;
      ENS(h->bglob, sB * w.glob_scratch_stride * 8);
      w.glob_scratch = h->bglob.as<double>();
    }
    {
      const int N3 = rmax + k; const int Npm = (N3 + 1) & ~1, ldm = Npm | 1;
      h->small_lds = ((size_t)n * rmax + (size_t)N3 * N3 + (size_t)Npm * ldm + 2 * Npm + (size_t)rmax * k + 8 + (size_t)n * 16 + 4) * 8 + (size_t)(Npm + 2) * 4 + 16;   // + Q' staging (n x 16)
      h->small_use_lds = h->small_lds + 1024 <= OMC_MAX_DYN_LDS;
      if (!h->small_use_lds) {
        w.small_scratch_stride = h->small_lds / 8 + 8;
        ENS(h->bsmall, sB * w.small_scratch_stride * 8);
        w.small_scratch = h->bsmall.as<double>();
      }
    }
  }
  HIPCHK(hipMemsetAsync(w.sweeps, 0, sizeof(int) * S, h->stream));
  HIPCHK(hipMemsetAsync(w.stamps, 0, (32 + 8 * (size_t)S) * 8, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->staged = true;
  return 0;
}

static int finish_events(omc_instance* h) {
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    float msv = 0.f;
    if (hipEventElapsedTime(&msv, h->ev_pool[i], h->ev_pool[i + 1]) == hipSuccess) h->ms[h->ev_class[i / 2]] += msv;
  }
  h->ev_used = 0; h->ev_class.clear();
  return 0;
}

int omc_set_node_rho_scales(omc_instance* h, int B, const double* rho_scale) {
  if (!h || B <= 0 || !rho_scale) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  for (int b = 0; b < B; ++b) if (!(rho_scale[b] > 0.0)) return fail(OMC_ERR_ARGUMENT, "rho_scale must be positive");
  h->rho_scale_per_node.assign(rho_scale, rho_scale + B);
  return 0;
}

// ---- warm start: a child is its parent plus one cut.  The caller keeps a pool of final states on the device and tells the
// next staged batch which entry each node starts from (its parent's) and where its own final state goes.  The reference rebuilds and cold-starts
// every model (OMC.jl:1482); this is an addition of the engine, off unless asked for.
int omc_state_pool_create(omc_instance* h, int capacity) {
  if (!h) return fail(OMC_ERR_ARGUMENT, "handle is NULL");
  if (capacity < 0) return fail(OMC_ERR_ARGUMENT, "capacity must be non-negative");
  HIPCHK(hipSetDevice(h->device));
  const size_t c = (size_t)capacity, n = h->n, k = h->k, np16 = (h->n + 15) & ~15;
  if (capacity > 0) {
    ENS(h->pY, c * n * n * 8); ENS(h->pD1, c * n * n * 8); ENS(h->pD3, c * n * n * 8); ENS(h->pU, c * n * k * 8); ENS(h->palpha, c * (size_t)h->nnz * 8);
    ENS(h->psval, c * (size_t)h->m * 8); ENS(h->pXs, c * np16 * 16 * 8); ENS(h->ptheta, c * 16 * 8); ENS(h->pscal, c * 4 * 8);
  }
  h->pool_cap = capacity;
  return 0;
}

int omc_relax_set_warm(omc_instance* h, int B, const int* load_from, const int* save_to) {
  if (!h || B <= 0) return fail(OMC_ERR_ARGUMENT, "omc_relax_set_warm: bad arguments");
  if (h->pool_cap <= 0) return fail(OMC_ERR_ARGUMENT, "omc_relax_set_warm: no state pool (omc_state_pool_create)");
  h->warm_load.clear(); h->warm_save.clear();
  for (int b = 0; b < B; ++b) {
    if (load_from && load_from[b] >= h->pool_cap) return fail(OMC_ERR_ARGUMENT, "load_from index beyond the pool");
    if (save_to && save_to[b] >= h->pool_cap) return fail(OMC_ERR_ARGUMENT, "save_to index beyond the pool");
  }
  if (load_from) h->warm_load.assign(load_from, load_from + B);
  if (save_to) h->warm_save.assign(save_to, save_to + B);
  return 0;
}

int omc_relax_solve(omc_instance* h) {
  if (!h || !h->staged) return fail(OMC_ERR_ARGUMENT, "omc_relax_solve: nothing staged");
  HIPCHK(hipSetDevice(h->device));
  const OmcWS& w = h->ws;
  const omc_relax_params& P = h->params;
  for (int c = 0; c < OMC_KERNEL_NCLASS; ++c) { h->launches[c] = 0; h->ms[c] = 0; h->units[c] = 0; }
  h->ev_used = 0; h->ev_class.clear();
  auto t0 = std::chrono::steady_clock::now();
  hipStream_t s = h->stream;
  const int S = w.B; int Btot = h->Btot_live.load();      // nodes staged so far: omc_relax_append may add more while this loop runs (re-read at every check)
  struct CloseGuard { omc_instance* h; ~CloseGuard() { std::lock_guard<std::mutex> lk(h->append_mu); h->append_closed = true; h->ws.Btot = h->Btot_live.load(); h->Btot = h->ws.Btot; } } close_guard{h};
  // slot bookkeeping on the host: node of each slot (-1 = idle), next pending node
  std::vector<int> node_of(S), flags(3 * (size_t)S), done(S, 0);
  for (int b = 0; b < S; ++b) node_of[b] = (b < Btot) ? b : -1;      // with omc_relax_reserve there may be more slots than nodes staged so far: the others start idle
  int next = std::min(S, Btot), harvested = 0, nactive = next;
  auto push_flags = [&](const std::vector<int>& init, const std::vector<int>& fin) -> int {
    for (int b = 0; b < S; ++b) { flags[b] = node_of[b] < 0 ? 0 : node_of[b]; flags[S + b] = init[b]; flags[2 * (size_t)S + b] = fin[b]; }
    HIPCHK(hipMemcpyAsync(w.node_of, flags.data(), sizeof(int) * flags.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));                  // `flags` is rewritten by the next push
    return 0;
  };
  {
    std::vector<int> init(S, 1), fin(S, 0);
    for (int b = 0; b < S; ++b) { init[b] = node_of[b] >= 0 ? 1 : 0; done[b] = node_of[b] >= 0 ? 0 : 1; }
    int rc = push_flags(init, fin); if (rc) return rc;
    if (nactive < S) { HIPCHK(hipMemcpyAsync(w.done, done.data(), sizeof(int) * S, hipMemcpyHostToDevice, s)); HIPCHK(hipStreamSynchronize(s)); }      // idle slots are skipped by every kernel
  }
  const bool shor = h->shor_on;
  const ShWS& sw = h->sh;
  if (shor) omc_shor_launch_setup(&sw, s);       // before the base setup, which clears the init flags
  TIMED(OMC_KERNEL_SETUP, S, omc_launch_setup(&w, s));
  int it = 0;
  const int check = std::max(1, P.check_every);
  // Streams.  Inside an iteration the three blocks are independent of each other (columns: Y, Yp, alpha -> alpha, Lambda;
  // cone: Y - D1 -> W1; small cone: Y, D3, V -> E3, W3*): a main stream (cone, then the global step) and two side streams
  // (columns, small cone) forked and joined by events, so that the latency-bound column waves and the small workgroups
  // share the CUs with the LDS-bound cone kernel (measured on config 2, 2048 slots: 212 -> 235 node-relaxations/s).
  // OMC_STREAMS=1 serialises everything on one stream (kernel-by-kernel measurements).  OMC_GROUPS=2 additionally splits
  // the slots in two groups with their own stream triple, meeting only at the certificate checks, so that one group's
  // global step overlaps the other's cone kernel: measured slower (219/s), kept for experiments only.
  const bool multi = !(h->tun.get("OMC_STREAMS") && atoi(h->tun.get("OMC_STREAMS")) <= 1);
  const int G = (multi && S >= 64 && h->tun.get("OMC_GROUPS") && atoi(h->tun.get("OMC_GROUPS")) == 2) ? 2 : 1;
  if (multi && !h->ev_main) {
    for (int g = 0; g < 2; ++g) {
      for (int q = 0; q < 4; ++q) HIPCHK(hipStreamCreateWithFlags(&h->gs[g][q], hipStreamNonBlocking));
      for (int q = 0; q < 5; ++q) { HIPCHK(hipEventCreateWithFlags(&h->gev[g][q], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&h->gevc[g][q], hipEventDisableTiming)); }
    }
    HIPCHK(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
  }
  // compact list of the slots that hold a running node: the per-iteration kernels launch over it (rebuilt when slots finish or are refilled)
  std::vector<char> parked(S, 0);      // finished, waiting for the next harvest
  int check_index = 0;
  const int refill_every = std::max(1, h->tun.get("OMC_REFILL_EVERY") ? atoi(h->tun.get("OMC_REFILL_EVERY")) : 3);
  std::vector<int> alist(S);
  for (int b = 0; b < S; ++b) alist[b] = b;
  int nlist = S;
  const bool use_list = !(h->tun.get("OMC_NO_SLOT_LIST"));
  auto push_list = [&]() -> int {
    nlist = 0;
    for (int b = 0; b < S; ++b) if (node_of[b] >= 0 && !parked[b]) alist[nlist++] = b;
    if (nlist) HIPCHK(hipMemcpyAsync(h->bslotlist.p, alist.data(), sizeof(int) * nlist, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
  };
  { int rc = push_list(); if (rc) return rc; }
  int gb0[2] = {0, 0}, gnB[2] = {S, 0}, gact[2] = {nactive, 0};
  if (G == 2) { gnB[0] = S / 2; gb0[1] = S / 2; gnB[1] = S - S / 2; gact[0] = gnB[0]; gact[1] = gnB[1]; }
  bool timed_out = false;
  h->total_sweeps = 0;
  bool wait_main = true;     // group streams must wait for the work queued on the main stream (setup, checks, refills)
  // Small batches are launch-bound (batch 1: ~190 us of launches, event records and waits around ~115 us of kernels per iteration):
  // the body of an iteration (fork, three concurrent blocks, join, global step) is captured once into a hipGraph and replayed; it is
  // captured again only when the number of live slots changes.  Per-kernel HIP-event timing is not available inside a graph, so the
  // large batches that the bench times keep the eager path.
  // hipGraph replay of the iteration body for small batches (measured at batch 1: 266 instead of 322 us per iteration).  Since the end of round 3
  // only for batches that are small FROM THE START (<= OMC_GRAPH_MAX = 16 nodes staged: one capture per solve): the draining tail of a large batch
  // re-captured the graph at every harvest, and a sporadic host crash inside omc_relax_solve was seen three times in the round, always in or after
  // solves on that path, never with replay off; not located.  OMC_GRAPH_TAILS=1 restores replay in the tails.
  const int graph_max_cfg = h->tun.get("OMC_GRAPH_MAX") ? atoi(h->tun.get("OMC_GRAPH_MAX")) : 16;
  const int graph_max = (Btot <= graph_max_cfg || h->tun.get("OMC_GRAPH_TAILS")) ? graph_max_cfg : 0;
  const bool shared_events = h->tun.get("OMC_GRAPH_SHARED_EVENTS") != nullptr;      // diagnostics: the pre-fix behaviour (captured and eager bodies use the same fork / join events)
  const bool no_graph = h->tun.get("OMC_NO_GRAPH") != nullptr;      // read once per solve, not per iteration
  const int timing_stride = h->tun.get("OMC_TIMING_STRIDE") ? atoi(h->tun.get("OMC_TIMING_STRIDE")) : 1;
  hipGraphExec_t gexec[2] = {nullptr, nullptr}; int gexec_n = -1;
  struct GraphGuard { hipGraphExec_t* e; ~GraphGuard() { for (int q = 0; q < 2; ++q) if (e[q]) (void)hipGraphExecDestroy(e[q]); } } gguard{gexec};
  auto body = [&](int g, const OmcWS& wg, bool timed, bool with_aa, bool capturing) -> int {
    hipEvent_t* const ev = (capturing && !shared_events) ? h->gevc[g] : h->gev[g];
    hipStream_t sm = multi ? h->gs[g][0] : s, sb = multi ? h->gs[g][1] : s, sc = multi ? h->gs[g][2] : s;
    if (multi) {
      HIPCHK(hipEventRecord(ev[0], sm));
      HIPCHK(hipStreamWaitEvent(sb, ev[0], 0)); HIPCHK(hipStreamWaitEvent(sc, ev[0], 0));
    }
#define MAYBE_TIMED(strm, cls, units_, call) do { if (timed) TIMED_ON(strm, cls, units_, call); else { call; } } while (0)
    // the cone workgroups are few (two per CU, long serial phases) and the column waves many: the cone kernel goes first so that its
    // workgroups are resident when the column kernel floods the wave slots (OMC_COLPROX_FIRST=1 restores the other order)
    const bool colprox_first = h->tun.get("OMC_COLPROX_FIRST") != nullptr;
    if (shor) {
      // Shor mode: clip on the main stream, the order-(n+m) cone on the second, small cone + order-5 blocks on the third; then the
      // global step: rows / Y (base kernel), columns (X, W, Theta, duals of the big cone), duals of the order-5 blocks, per-slot sums
      OmcWS wb = h->wbig; wb.b0 = wg.b0; wb.nB = wg.nB; wb.slot_list = wg.slot_list;
      if (w.sub_enable) MAYBE_TIMED(sm, OMC_KERNEL_CONESUB, gact[g], omc_launch_cone_sub(&wg, sm));
      if (wb.sub_enable) MAYBE_TIMED(sb, OMC_KERNEL_SHOR_BIGCONE, 0, omc_launch_cone_sub(&wb, sb));
      if (h->big_lpp) MAYBE_TIMED(sb, OMC_KERNEL_SHOR_BIGCONE, gact[g], omc_launch_cone_ws(&wb, h->big_lpp, h->big_use_lds, h->big_lds, sb));
      else MAYBE_TIMED(sb, OMC_KERNEL_SHOR_BIGCONE, gact[g], omc_launch_cone(&wb, CONE_BIG, h->big_cone_lds_ok, h->big_cone_lds, sb));
      if (h->ws_lpp) MAYBE_TIMED(sm, OMC_KERNEL_CONE, gact[g], omc_launch_cone_ws(&wg, h->ws_lpp, h->ws_use_lds, h->ws_lds, sm));
      else MAYBE_TIMED(sm, OMC_KERNEL_CONE, gact[g], omc_launch_cone(&wg, CONE_CLIP01, h->cone_use_lds, h->cone_lds, sm));
      MAYBE_TIMED(sc, OMC_KERNEL_SMALL, gact[g], omc_launch_small(&wg, SMALL_PROJ, h->small_use_lds, h->small_lds, sc));
      MAYBE_TIMED(sc, OMC_KERNEL_SHOR_MINORS, gact[g], { omc_shor_launch_minor_pre(&sw, sc); omc_shor_launch_vkeys(&sw, sc); });
      if (multi) {
        HIPCHK(hipEventRecord(ev[1], sb)); HIPCHK(hipEventRecord(ev[2], sc));
        HIPCHK(hipStreamWaitEvent(sm, ev[1], 0)); HIPCHK(hipStreamWaitEvent(sm, ev[2], 0));
      }
      MAYBE_TIMED(sm, OMC_KERNEL_GLOBAL, gact[g], omc_launch_global(&wg, h->glob_use_lds, h->glob_lds, sm));
      MAYBE_TIMED(sm, OMC_KERNEL_SHOR_COLS, gact[g], omc_shor_launch_cols(&sw, sm));
      MAYBE_TIMED(sm, OMC_KERNEL_SHOR_MINORS, gact[g], { omc_shor_launch_minor_post(&sw, sm); omc_shor_launch_reduce(&sw, sm); });
      return 0;
    }
    // The full eigen-kernel runs the slots that have no tracked block (or are backing off) -- a handful per launch, each a long single-workgroup
    // job, known before the iteration starts (ws_first) -- on a stream of its own beside k_cone_sub; what k_cone_sub then could not do (a failed
    // call, ~1 in 30 000) is a second, almost empty launch behind both.  One launch after k_cone_sub made every iteration wait for the sum.
    const bool ws_split_ok = h->tun.get("OMC_NO_WS_SPLIT") == nullptr;
    const bool split = multi && ws_split_ok && w.sub_enable && h->ws_lpp && w.ws_first;
    if (split) {      // on the small-cone stream, behind k_small (a fifth stream would share a hardware queue with one of the other four: measured, k_small then ran behind it)
      MAYBE_TIMED(sc, OMC_KERNEL_SMALL, gact[g], omc_launch_small(&wg, SMALL_PROJ, h->small_use_lds, h->small_lds, sc));
      OmcWS wA = wg; wA.ws_phase = 1;
      MAYBE_TIMED(sc, OMC_KERNEL_CONE, gact[g], omc_launch_cone_ws(&wA, h->ws_lpp, h->ws_use_lds, h->ws_lds, sc));
      HIPCHK(hipEventRecord(ev[4], sc));
    }
    if (colprox_first) MAYBE_TIMED(sb, OMC_KERNEL_COLPROX, (int64_t)gact[g] * w.m, omc_launch_colprox(&wg, 0, sb));
    if (w.sub_enable) MAYBE_TIMED(sm, OMC_KERNEL_CONESUB, gact[g], omc_launch_cone_sub(&wg, sm));
    if (!colprox_first) MAYBE_TIMED(sb, OMC_KERNEL_COLPROX, (int64_t)gact[g] * w.m, omc_launch_colprox(&wg, 0, sb));
    if (split) {
      HIPCHK(hipStreamWaitEvent(sm, ev[4], 0));
      OmcWS wB = wg; wB.ws_phase = 2;
      MAYBE_TIMED(sm, OMC_KERNEL_CONE, 0, omc_launch_cone_ws(&wB, h->ws_lpp, h->ws_use_lds, h->ws_lds, sm));
    }
    else if (h->ws_lpp) MAYBE_TIMED(sm, OMC_KERNEL_CONE, gact[g], omc_launch_cone_ws(&wg, h->ws_lpp, h->ws_use_lds, h->ws_lds, sm));
    else MAYBE_TIMED(sm, OMC_KERNEL_CONE, gact[g], omc_launch_cone(&wg, CONE_CLIP01, h->cone_use_lds, h->cone_lds, sm));
    if (!split) MAYBE_TIMED(sc, OMC_KERNEL_SMALL, gact[g], omc_launch_small(&wg, SMALL_PROJ, h->small_use_lds, h->small_lds, sc));
    if (multi) {
      HIPCHK(hipEventRecord(ev[1], sb)); HIPCHK(hipEventRecord(ev[2], sc));
      HIPCHK(hipStreamWaitEvent(sm, ev[1], 0)); HIPCHK(hipStreamWaitEvent(sm, ev[2], 0));
    }
    MAYBE_TIMED(sm, OMC_KERNEL_GLOBAL, gact[g], omc_launch_global(&wg, h->glob_use_lds, h->glob_lds, sm));
    if (with_aa) MAYBE_TIMED(sm, OMC_KERNEL_ACCEL, gact[g], omc_launch_aa(&wg, sm));
    return 0;
  };
  // nodes appended while every slot was idle (or while the loop was about to end): hand them to idle slots
  auto refill_idle = [&]() -> int {
    std::vector<int> init2(S, 0), fin2(S, 0);
    int ninit2 = 0;
    for (int b = 0; b < S && next < Btot; ++b) if (node_of[b] < 0) { node_of[b] = next++; init2[b] = 1; parked[b] = 0; ++ninit2; }
    if (!ninit2) return 0;
    int rc = push_flags(init2, fin2); if (rc) return rc;
    TIMED(OMC_KERNEL_SETUP, ninit2, omc_launch_setup(&w, s));
    nactive = 0; gact[0] = gact[1] = 0;
    for (int b = 0; b < S; ++b) if (node_of[b] >= 0) { ++nactive; if (!parked[b]) ++gact[(G == 2 && b >= gb0[1]) ? 1 : 0]; }
    rc = push_list(); if (rc) return rc;
    wait_main = true;
    return 0;
  };
  for (;;) {
    if (nactive == 0) {
      {   // the end of the batch is decided under the lock omc_relax_append takes: a node is either seen here or refused there
        std::lock_guard<std::mutex> lk(h->append_mu);
        Btot = h->Btot_live.load();
        if (timed_out && next < Btot) {      // appended after the time limit struck: reported as TIME_LIMIT without values, like the nodes that never got a slot
          std::vector<int> st(Btot - next, OMC_ST_TIME), itz(Btot - next, 0);
          std::vector<double> inf(Btot - next, 1e300), ninf(Btot - next, -1e300);
          HIPCHK(hipMemcpyAsync(w.ostatus + next, st.data(), sizeof(int) * st.size(), hipMemcpyHostToDevice, s));
          HIPCHK(hipMemcpyAsync(w.oiters + next, itz.data(), sizeof(int) * itz.size(), hipMemcpyHostToDevice, s));
          HIPCHK(hipMemcpyAsync(w.oobj + next, inf.data(), 8 * inf.size(), hipMemcpyHostToDevice, s));
          HIPCHK(hipMemcpyAsync(w.olb + next, ninf.data(), 8 * ninf.size(), hipMemcpyHostToDevice, s));
          HIPCHK(hipStreamSynchronize(s));
          next = Btot;
        }
        if (next >= Btot) {
          const double el_idle = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          if (!h->hold.load() || el_idle > P.time_limit) { h->append_closed = true; break; }
        }
      }
      if (next >= Btot) { std::this_thread::sleep_for(std::chrono::microseconds(100)); continue; }      // held open (omc_relax_hold): wait for the host's next push
      int rc = refill_idle(); if (rc) return rc;
      continue;
    }
    ++it;
    const bool is_check = (it % check == 0);
    if (multi && wait_main) HIPCHK(hipEventRecord(h->ev_main, s));
    // per-kernel HIP-event timing brackets every launch of a sampled iteration (two event records per kernel: ~25 us of queue bubbles per
    // iteration at small batches); OMC_TIMING_STRIDE=s samples every s-th iteration (averages per launch are over the sampled launches), 0 = none
    const bool sampled = timing_stride > 0 && (it % timing_stride) == 0;
    const bool use_graph = multi && G == 1 && use_list && nlist <= graph_max && !no_graph && !(sampled && timing_stride > 1);
    for (int g = 0; g < G; ++g) {
      if (gact[g] == 0) continue;
      OmcWS wg = w; wg.b0 = gb0[g]; wg.nB = gnB[g];
      if (use_list && G == 1) { wg.slot_list = h->bslotlist.as<int>(); wg.b0 = 0; wg.nB = nlist; }
      hipStream_t sm = multi ? h->gs[g][0] : s;
      if (multi && wait_main) HIPCHK(hipStreamWaitEvent(sm, h->ev_main, 0));
      if (use_graph) {
        if (gexec_n != nlist) {      // (re)capture: one graph without and one with the acceleration kernel at its end
          for (int q = 0; q < 2; ++q) { if (gexec[q]) { (void)hipGraphExecDestroy(gexec[q]); gexec[q] = nullptr; } }
          for (int q = 0; q < (w.accel ? 2 : 1); ++q) {
            hipGraph_t gr = nullptr;
            HIPCHK(hipStreamBeginCapture(sm, hipStreamCaptureModeThreadLocal));
            int rc = body(g, wg, false, q == 1, true);
            hipError_t ce = hipStreamEndCapture(sm, &gr);
            if (rc) { if (gr) (void)hipGraphDestroy(gr); return rc; }
            HIPCHK(ce);
            hipError_t ie = hipGraphInstantiate(&gexec[q], gr, nullptr, nullptr, 0);
            (void)hipGraphDestroy(gr);
            HIPCHK(ie);
          }
          gexec_n = nlist;
        }
        const int q = (!is_check && w.accel) ? 1 : 0;
        HIPCHK(hipGraphLaunch(gexec[q], sm));
        h->launches[OMC_KERNEL_GLOBAL] += 1; h->units[OMC_KERNEL_GLOBAL] += gact[g];
      } else {
        int rc = body(g, wg, sampled, !is_check && w.accel, false); if (rc) return rc;
      }
      if (multi && is_check) { HIPCHK(hipEventRecord(h->gev[g][3], sm)); HIPCHK(hipStreamWaitEvent(s, h->gev[g][3], 0)); }
    }
    wait_main = false;
    if (!is_check) continue;
    wait_main = true;
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    timed_out = el > P.time_limit;
    TIMED(OMC_KERNEL_CHECK_COL, nactive, {
      if (shor) omc_shor_launch_check(&sw, s);      // primal value, constants and the dense multiplier of the Shor program
      else { omc_launch_check_zero(&w, s); omc_launch_colprox(&w, 1, s); }
    });
    TIMED(OMC_KERNEL_CHECK_BUILD, nactive, omc_launch_check_build(&w, s));
    TIMED(OMC_KERNEL_CHECK, nactive, {
      if (w.cert_enable) {        // estimate by the tracked block, decisions, rigorous evaluation of the slots that are about to finish
        omc_launch_cert_sub(&w, s);
        omc_launch_check_final(&w, timed_out ? OMC_ST_TIME : 0, 0, s);
        OmcWS wc = w; wc.ws_mode = 1; omc_launch_cone_ws(&wc, h->ws_lpp, h->ws_use_lds, h->ws_lds, s);
        omc_launch_check_final(&w, timed_out ? OMC_ST_TIME : 0, 1, s);
      } else {
        if (h->ws_lpp && w.MbufC) { OmcWS wc = w; wc.ws_mode = 1; omc_launch_cone_ws(&wc, h->ws_lpp, h->ws_use_lds, h->ws_lds, s); }
        else omc_launch_cone(&w, CONE_EVALS, h->cone_use_lds, h->cone_lds, s);
        omc_launch_check_final(&w, timed_out ? OMC_ST_TIME : 0, 2, s);      // per-slot iteration cap is applied on the device
      }
      if (w.bump_max > 0) { if (shor) omc_shor_launch_rescale(&sw, s); omc_launch_rho_rescale(&w, s); }
      if (w.accel) omc_launch_aa(&w, s);      // after the certificate (computed on an image of the map), skips finished slots
    });
    HIPCHK(hipMemcpyAsync(done.data(), w.done, sizeof(int) * S, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    Btot = h->Btot_live.load();
    if (P.first_wins) {   // the first certified node ends the batch (penalty autotune): everything still running is harvested as it stands
      std::vector<int> stv(S);
      HIPCHK(hipMemcpyAsync(stv.data(), w.status, sizeof(int) * S, hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      bool won = false;
      for (int b = 0; b < S; ++b) if (node_of[b] >= 0 && done[b] && stv[b] == OMC_ST_OPTIMAL) won = true;
      if (won) {
        for (int b = 0; b < S; ++b) done[b] = 1;
        HIPCHK(hipMemcpyAsync(w.done, done.data(), sizeof(int) * S, hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        if (next < Btot) {   // nodes that never got a slot
          std::vector<int> st(Btot - next, OMC_ST_SLOW), itz(Btot - next, 0);
          std::vector<double> inf(Btot - next, 1e300), ninf(Btot - next, -1e300);
          HIPCHK(hipMemcpyAsync(w.ostatus + next, st.data(), sizeof(int) * st.size(), hipMemcpyHostToDevice, s));
          HIPCHK(hipMemcpyAsync(w.oiters + next, itz.data(), sizeof(int) * itz.size(), hipMemcpyHostToDevice, s));
          HIPCHK(hipMemcpyAsync(w.oobj + next, inf.data(), 8 * inf.size(), hipMemcpyHostToDevice, s));
          HIPCHK(hipMemcpyAsync(w.olb + next, ninf.data(), 8 * ninf.size(), hipMemcpyHostToDevice, s));
          HIPCHK(hipStreamSynchronize(s));
          next = Btot;
        }
      }
    }
    // harvest finished slots, hand them the next pending nodes
    // Harvest and refill every `refill_every`-th check only (or when nothing is left running): a refilled slot spends its first dozen
    // iterations in the full eigendecomposition, the straggler of every launch it is part of, and each harvest is 1 - 3 ms of few-workgroup
    // kernels on the main stream -- batching them halves the launches that carry young slots.  A finished slot waits (done = 1, skipped
    // by every kernel and left out of the slot list) for at most refill_every - 1 check intervals.
    std::vector<int> init(S, 0), fin(S, 0);
    int nfin = 0, nlive = 0, nnew = 0;
    for (int b = 0; b < S; ++b) {
      if (node_of[b] < 0) continue;
      if (done[b]) { ++nfin; if (!parked[b]) { parked[b] = 1; ++nnew; } } else ++nlive;
    }
    ++check_index;
    // with pending nodes: every refill_every-th check, at once when the live slots no longer fill the chip; without: the finished slots can
    // wait longer (nothing to hand them), until nothing runs any more
    const bool harvest_now = nfin > 0 && (nlive == 0 || (next < Btot ? (check_index % refill_every == 0 || nlive < 256) : (check_index % (4 * refill_every) == 0)));
    if (harvest_now) for (int b = 0; b < S; ++b) if (node_of[b] >= 0 && done[b]) { fin[b] = 1; parked[b] = 0; }
    if (!harvest_now) nfin = 0;
    if (nfin) {
      int rc = push_flags(init, fin); if (rc) return rc;
      TIMED(OMC_KERNEL_HARVEST, nfin, {
        if (w.save_to) omc_launch_state_save(&w, s);                              // warm-start pool: before the recovery overwrites the iterate U = Q Vt
        omc_launch_small(&w, SMALL_RECOVER, h->small_use_lds, h->small_lds, s);   // a U with U U' <= Y and the same Q'U
        if (w.sep_done) omc_launch_sep_sub(&w, s);                                // separation vector from the tracked block where there is one
        omc_launch_cone(&w, CONE_SEP, h->cone_use_lds, h->cone_lds, s);           // separation vector (OMC.jl:2466-2477)
        omc_launch_harvest(&w, s);
        if (shor) omc_shor_launch_harvest(&sw, s);
      });
      harvested += nfin; h->nodes_done.store(harvested);
      int ninit = 0;
      std::vector<int> harvested_ids;
      for (int b = 0; b < S; ++b) {
        if (!fin[b]) continue;
        fin[b] = 0;
        harvested_ids.push_back(node_of[b]);
        if (next < Btot && !timed_out) { node_of[b] = next++; init[b] = 1; ++ninit; }
        else node_of[b] = -1;
      }
      rc = push_flags(init, fin); if (rc) return rc;      // synchronises the stream: the harvest kernels have written the per-node outputs
      { std::lock_guard<std::mutex> lk(h->done_mu); h->done_q.insert(h->done_q.end(), harvested_ids.begin(), harvested_ids.end()); }
      if (ninit) { if (shor) omc_shor_launch_setup(&sw, s); TIMED(OMC_KERNEL_SETUP, ninit, omc_launch_setup(&w, s)); }
    }
    nactive = 0; gact[0] = gact[1] = 0;
    for (int b = 0; b < S; ++b) if (node_of[b] >= 0) { ++nactive; if (!parked[b]) ++gact[(G == 2 && b >= gb0[1]) ? 1 : 0]; }
    if (nfin || nnew) { int rc = push_list(); if (rc) return rc; }
    if (next < Btot && !timed_out && nactive < S) { int rc = refill_idle(); if (rc) return rc; }      // appended nodes for slots that had gone idle
    if (timed_out && next < Btot) {
      // nodes that never got a slot: report TIME_LIMIT without values
      std::vector<int> st(Btot - next, OMC_ST_TIME), itz(Btot - next, 0);
      std::vector<double> inf(Btot - next, 1e300), ninf(Btot - next, -1e300);
      HIPCHK(hipMemcpyAsync(w.ostatus + next, st.data(), sizeof(int) * st.size(), hipMemcpyHostToDevice, s));
      HIPCHK(hipMemcpyAsync(w.oiters + next, itz.data(), sizeof(int) * itz.size(), hipMemcpyHostToDevice, s));
      HIPCHK(hipMemcpyAsync(w.oobj + next, inf.data(), 8 * inf.size(), hipMemcpyHostToDevice, s));
      HIPCHK(hipMemcpyAsync(w.olb + next, ninf.data(), 8 * ninf.size(), hipMemcpyHostToDevice, s));
      HIPCHK(hipStreamSynchronize(s));
      next = Btot;
    }
  }
  {
    std::vector<int> sw(S);
    HIPCHK(hipMemcpyAsync(sw.data(), w.sweeps, sizeof(int) * S, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int v : sw) h->total_sweeps += v;
    std::vector<int> ss(8 * (size_t)S);
    HIPCHK(hipMemcpyAsync(ss.data(), w.sub_stat, sizeof(int) * ss.size(), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int q = 0; q < 8; ++q) h->sub_tot[q] = 0;
    for (int b = 0; b < S; ++b) for (int q = 0; q < 8; ++q) h->sub_tot[q] += ss[8 * b + q];
    HIPCHK(hipMemsetAsync(w.sub_stat, 0, sizeof(int) * ss.size(), s));
  }
  if (shor && h->wbig.sub_enable) {      // tracked-subspace accounting of the big cone (same layout as omc_last_subspace_stats)
    std::vector<int> ss(8 * (size_t)S);
    HIPCHK(hipMemcpyAsync(ss.data(), h->wbig.sub_stat, sizeof(int) * ss.size(), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int q = 0; q < 8; ++q) h->big_sub_tot[q] = 0;
    for (int b = 0; b < S; ++b) for (int q = 0; q < 8; ++q) h->big_sub_tot[q] += ss[8 * b + q];
    if (h->tun.get("OMC_SHOR_DEBUG")) for (int b = 0; b < S && b < 4; ++b) fprintf(stderr, "slot %d big-cone subspace: calls %d steps %d fallbacks %d seeds %d\n", b, ss[8 * b], ss[8 * b + 1], ss[8 * b + 2], ss[8 * b + 3]);
    HIPCHK(hipMemsetAsync(h->wbig.sub_stat, 0, sizeof(int) * ss.size(), s));
  }
  HIPCHK(hipGetLastError());
  finish_events(h);
  h->last_solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  h->last_iters_total = it;
  (void)harvested;
  return 0;
}

// ---- appending nodes to a staged / running batch ------------------------------------------------------------------------------------------
// The reference's loop pops one node at a time from a queue that children keep filling (OMC.jl:700-719, 2520-2542); a staged batch is closed.
// omc_relax_reserve(h, extra_nodes, max_cuts) makes the NEXT omc_relax_stage size its per-node arrays for extra_nodes more nodes with at most
// max_cuts cuts each (default U bounds); omc_relax_append then adds nodes -- before omc_relax_solve / omc_relax_submit or while the submitted
// solve is running (the loop hands them to slots at its next check) -- until the solve has ended, after which it is refused.
int omc_relax_reserve(omc_instance* h, int extra_nodes, int max_cuts) {
  if (!h) return fail(OMC_ERR_ARGUMENT, "handle is NULL");
  if (extra_nodes < 0 || max_cuts < 0) return fail(OMC_ERR_ARGUMENT, "omc_relax_reserve: negative argument");
  h->reserve_nodes = extra_nodes; h->reserve_cuts = max_cuts;
  return 0;
}

int omc_relax_append(omc_instance* h, int B2, const int* L, const double* cut_x, const double* cut_Uhat, const int8_t* cut_dir,
                     const int* load_from, const int* save_to) {
  if (!h || !h->staged) return fail(OMC_ERR_ARGUMENT, "omc_relax_append: nothing staged");
  if (B2 <= 0) return fail(OMC_ERR_ARGUMENT, "B must be positive");
  if (h->shor_on) return fail(OMC_ERR_UNSUPPORTED, "omc_relax_append: not available in Shor mode");
  HIPCHK(hipSetDevice(h->device));
  std::lock_guard<std::mutex> lk(h->append_mu);
  if (h->append_closed) return fail(OMC_ERR_ARGUMENT, "omc_relax_append: the solve of this batch has ended (stage a new batch)");
  const OmcWS& w = h->ws;
  const int first = h->Btot_live.load();
  if (first + B2 > h->node_cap) return fail(OMC_ERR_ARGUMENT, "omc_relax_append: beyond the capacity given to omc_relax_reserve");
  NodePack pk;
  { int rcp = pack_nodes(h, h->params, h->staged_cut_type, B2, L, cut_x, cut_Uhat, cut_dir, nullptr, nullptr, pk); if (rcp) return rcp; }
  if (pk.Rmax > w.Rmax || pk.rmax > w.rmax || pk.Lmax > w.Lmax) return fail(OMC_ERR_ARGUMENT, "omc_relax_append: a node has more cuts than omc_relax_reserve allowed for");
  const size_t n = h->n, k = h->k, Rm = w.Rmax, rm = w.rmax, Lm = w.Lmax;
  std::vector<int> hR(B2), hk((size_t)B2 * Rm, 0), hc((size_t)B2 * Rm, 0), hbi((size_t)B2 * Rm, 0), hbj((size_t)B2 * Rm, 0);
  std::vector<double> hcoef((size_t)B2 * Rm * k, 0.0), hrhs((size_t)B2 * Rm, 0.0), hx((size_t)B2 * Lm * n, 0.0), hQ((size_t)B2 * n * rm, 0.0), hrho(B2, w.rho);
  long cutbase = 0;
  for (int b = 0; b < B2; ++b) {
    hR[b] = (int)pk.rk[b].size();
    for (int r = 0; r < hR[b]; ++r) {
      hk[(size_t)b * Rm + r] = pk.rk[b][r]; hc[(size_t)b * Rm + r] = pk.rc[b][r];
      hbi[(size_t)b * Rm + r] = pk.rbi[b][r]; hbj[(size_t)b * Rm + r] = pk.rbj[b][r];
      hrhs[(size_t)b * Rm + r] = pk.rrhs[b][r];
      for (size_t j = 0; j < k; ++j) hcoef[((size_t)b * Rm + r) * k + j] = pk.rcoef[b][(size_t)r * k + j];
    }
    const int Lb = L ? L[b] : 0;
    for (int l = 0; l < Lb; ++l) memcpy(&hx[((size_t)b * Lm + l) * n], cut_x + (size_t)(cutbase + l) * n, sizeof(double) * n);
    cutbase += Lb;
    memcpy(&hQ[(size_t)b * n * rm], pk.Qn[b].data(), sizeof(double) * pk.Qn[b].size());
  }
  if ((load_from || save_to) && !(w.load_from && w.save_to)) return fail(OMC_ERR_ARGUMENT, "omc_relax_append: warm-start indices need a state pool created before the batch was staged");
  for (int b = 0; b < B2; ++b) {
    if (load_from && load_from[b] >= h->pool_cap) return fail(OMC_ERR_ARGUMENT, "load_from index beyond the pool");
    if (save_to && save_to[b] >= h->pool_cap) return fail(OMC_ERR_ARGUMENT, "save_to index beyond the pool");
  }
  if (!h->append_stream) HIPCHK(hipStreamCreateWithFlags(&h->append_stream, hipStreamNonBlocking));
  hipStream_t as = h->append_stream;
  const size_t f = (size_t)first;
  // the running kernels read the descriptors of nodes below Btot_live only: these rows are not visible to them until the counter moves
  HIPCHK(hipMemcpyAsync(w.R + f, hR.data(), sizeof(int) * B2, hipMemcpyHostToDevice, as));
  HIPCHK(hipMemcpyAsync(w.rkind + f * Rm, hk.data(), sizeof(int) * hk.size(), hipMemcpyHostToDevice, as));
  HIPCHK(hipMemcpyAsync(w.rcut + f * Rm, hc.data(), sizeof(int) * hc.size(), hipMemcpyHostToDevice, as));
  HIPCHK(hipMemcpyAsync(w.rbi + f * Rm, hbi.data(), sizeof(int) * hbi.size(), hipMemcpyHostToDevice, as));
  HIPCHK(hipMemcpyAsync(w.rbj + f * Rm, hbj.data(), sizeof(int) * hbj.size(), hipMemcpyHostToDevice, as));
  HIPCHK(hipMemcpyAsync(w.rcoef + f * Rm * k, hcoef.data(), 8 * hcoef.size(), hipMemcpyHostToDevice, as));
  HIPCHK(hipMemcpyAsync(w.rrhs + f * Rm, hrhs.data(), 8 * hrhs.size(), hipMemcpyHostToDevice, as));
  HIPCHK(hipMemcpyAsync(w.cutx + f * Lm * n, hx.data(), 8 * hx.size(), hipMemcpyHostToDevice, as));
  HIPCHK(hipMemcpyAsync(w.Qb + f * n * rm, hQ.data(), 8 * hQ.size(), hipMemcpyHostToDevice, as));
  HIPCHK(hipMemcpyAsync(w.rr + f, pk.rrv.data(), sizeof(int) * B2, hipMemcpyHostToDevice, as));
  HIPCHK(hipMemcpyAsync(const_cast<double*>(w.rho_node) + f, hrho.data(), 8 * (size_t)B2, hipMemcpyHostToDevice, as));
  if (load_from) HIPCHK(hipMemcpyAsync(const_cast<int*>(w.load_from) + f, load_from, sizeof(int) * B2, hipMemcpyHostToDevice, as));
  if (save_to) HIPCHK(hipMemcpyAsync(const_cast<int*>(w.save_to) + f, save_to, sizeof(int) * B2, hipMemcpyHostToDevice, as));
  HIPCHK(hipStreamSynchronize(as));
  h->Btot_live.store(first + B2); h->Btot = first + B2;
  if (!h->worker_running.load()) h->ws.Btot = first + B2;      // no solve in flight: the staged batch simply grew
  return 0;
}

// omc_relax_hold(h, 1) after staging: the submitted solve does not end when every staged node has finished but waits for omc_relax_append
// (a queue-driven host creates the next nodes from the results of the last ones); omc_relax_hold(h, 0) lets it end once it runs dry.
int omc_relax_hold(omc_instance* h, int on) {
  if (!h) return fail(OMC_ERR_ARGUMENT, "handle is NULL");
  h->hold.store(on ? 1 : 0);
  return 0;
}

// Results of the nodes harvested since the last call, in harvest order -- also while the submitted solve is still running: with omc_relax_append
// this is the other half of a queue-driven host loop (OMC.jl:700-719: pop, relax, push the children).  node_ids[i] indexes the staged + appended
// nodes; U (n*k), lambda_min (2), breakpoint_x (n) and Y (n*n) per node as in omc_relax_fetch (NULL: not copied).  *n_out = nodes returned (<= max_nodes).
int omc_relax_fetch_done(omc_instance* h, int max_nodes, int* node_ids, double* objective, double* dual_bound, int* status, int* iters,
                         double* U, double* lambda_min, double* breakpoint_x, double* Y, int* n_out) {
  if (!h || !h->staged || !node_ids || !n_out) return fail(OMC_ERR_ARGUMENT, "omc_relax_fetch_done: nothing staged or NULL argument");
  if (max_nodes < 0) return fail(OMC_ERR_ARGUMENT, "max_nodes is negative");
  HIPCHK(hipSetDevice(h->device));
  std::vector<int> ids;
  {
    std::lock_guard<std::mutex> lk(h->done_mu);
    const size_t avail = h->done_q.size() - h->done_read, take = std::min(avail, (size_t)max_nodes);
    ids.assign(h->done_q.begin() + h->done_read, h->done_q.begin() + h->done_read + take);
    h->done_read += take;
  }
  *n_out = (int)ids.size();
  if (ids.empty()) return 0;
  if (!h->fetch_stream) HIPCHK(hipStreamCreateWithFlags(&h->fetch_stream, hipStreamNonBlocking));
  hipStream_t fs = h->fetch_stream;
  const OmcWS& w = h->ws;
  const size_t n = h->n, k = h->k;
  for (size_t i = 0; i < ids.size(); ++i) {
    const size_t nb = (size_t)ids[i];
    node_ids[i] = ids[i];
    if (objective) HIPCHK(hipMemcpyAsync(objective + i, w.oobj + nb, 8, hipMemcpyDeviceToHost, fs));
    if (dual_bound) HIPCHK(hipMemcpyAsync(dual_bound + i, w.olb + nb, 8, hipMemcpyDeviceToHost, fs));
    if (status) HIPCHK(hipMemcpyAsync(status + i, w.ostatus + nb, 4, hipMemcpyDeviceToHost, fs));
    if (iters) HIPCHK(hipMemcpyAsync(iters + i, w.oiters + nb, 4, hipMemcpyDeviceToHost, fs));
    if (U) HIPCHK(hipMemcpyAsync(U + i * n * k, w.oU + nb * n * k, 8 * n * k, hipMemcpyDeviceToHost, fs));
    if (lambda_min) HIPCHK(hipMemcpyAsync(lambda_min + 2 * i, w.olmin + 2 * nb, 16, hipMemcpyDeviceToHost, fs));
    if (breakpoint_x) HIPCHK(hipMemcpyAsync(breakpoint_x + i * n, w.obx + nb * n, 8 * n, hipMemcpyDeviceToHost, fs));
    if (Y) HIPCHK(hipMemcpyAsync(Y + i * n * n, w.oY + nb * n * n, 8 * n * n, hipMemcpyDeviceToHost, fs));
  }
  HIPCHK(hipStreamSynchronize(fs));
  return 0;
}

// ---- asynchronous boundary: the solve runs on a worker thread of the library, the caller (the Julia task that owns the queue) keeps
// working -- popping, pruning, building the next batch -- and polls.  One solve in flight per handle; no other call on the handle
// except omc_relax_poll until omc_relax_wait has returned.
int omc_relax_submit(omc_instance* h) {
  if (!h || !h->staged) return fail(OMC_ERR_ARGUMENT, "omc_relax_submit: nothing staged");
  if (h->worker.joinable()) return fail(OMC_ERR_ARGUMENT, "omc_relax_submit: a solve is already in flight (call omc_relax_wait first)");
  h->nodes_done.store(0); h->worker_running.store(1); h->worker_rc = 0; h->worker_err.clear();
  h->worker = std::thread([h]() {
    const int rc = omc_relax_solve(h);
    h->worker_rc = rc;
    if (rc) h->worker_err = g_err;        // g_err is thread-local: hand the message to the thread that will call omc_relax_wait
    h->worker_running.store(0);
  });
  return 0;
}

int omc_relax_poll(omc_instance* h, int* running, int* nodes_done, int* nodes_total) {
  if (!h) return fail(OMC_ERR_ARGUMENT, "handle is NULL");
  if (running) *running = h->worker_running.load();
  if (nodes_done) *nodes_done = h->nodes_done.load();
  if (nodes_total) *nodes_total = h->Btot_live.load();
  return 0;
}

int omc_relax_wait(omc_instance* h) {
  if (!h) return fail(OMC_ERR_ARGUMENT, "handle is NULL");
  if (!h->worker.joinable()) return fail(OMC_ERR_ARGUMENT, "omc_relax_wait: no solve in flight");
  h->worker.join();
  if (h->worker_rc) g_err = h->worker_err;
  return h->worker_rc;
}

int omc_relax_fetch(omc_instance* h, double* objective, double* dual_bound, int* status, int* iters, double* Y,
                    double* U, double* X, double* Theta, double* lambda_min, double* breakpoint_x, double* solve_time) {
  if (!h || !h->staged) return fail(OMC_ERR_ARGUMENT, "omc_relax_fetch: nothing staged");
  HIPCHK(hipSetDevice(h->device));
  const OmcWS& w = h->ws;
  const size_t B = w.Btot, n = w.n, m = w.m, k = w.k;
  hipStream_t s = h->stream;
  if (objective) HIPCHK(hipMemcpyAsync(objective, w.oobj, 8 * B, hipMemcpyDeviceToHost, s));
  if (dual_bound) HIPCHK(hipMemcpyAsync(dual_bound, w.olb, 8 * B, hipMemcpyDeviceToHost, s));
  if (status) HIPCHK(hipMemcpyAsync(status, w.ostatus, 4 * B, hipMemcpyDeviceToHost, s));
  if (iters) HIPCHK(hipMemcpyAsync(iters, w.oiters, 4 * B, hipMemcpyDeviceToHost, s));
  if (Y) HIPCHK(hipMemcpyAsync(Y, w.oY, 8 * B * n * n, hipMemcpyDeviceToHost, s));
  if (U) HIPCHK(hipMemcpyAsync(U, w.oU, 8 * B * n * k, hipMemcpyDeviceToHost, s));
  if (lambda_min) HIPCHK(hipMemcpyAsync(lambda_min, w.olmin, 8 * 2 * B, hipMemcpyDeviceToHost, s));
  if (breakpoint_x) HIPCHK(hipMemcpyAsync(breakpoint_x, w.obx, 8 * B * n, hipMemcpyDeviceToHost, s));
  if (h->shor_on) {
    if (X) HIPCHK(hipMemcpyAsync(X, h->sh.oX, 8 * B * n * m, hipMemcpyDeviceToHost, s));
    if (Theta) HIPCHK(hipMemcpyAsync(Theta, h->sh.oTh, 8 * B * m * m, hipMemcpyDeviceToHost, s));
  } else if (X || Theta) {
    int r_ = h->bXout.ensure(8 * B * n * m); if (r_) return r_;
    omc_launch_make_X(&w, h->bXout.as<double>(), s);
    if (X) HIPCHK(hipMemcpyAsync(X, h->bXout.p, 8 * B * n * m, hipMemcpyDeviceToHost, s));
    if (Theta) {
      r_ = h->bThout.ensure(8 * B * m * m); if (r_) return r_;
      omc_launch_make_Theta(&w, h->bXout.as<double>(), h->bThout.as<double>(), s);
      HIPCHK(hipMemcpyAsync(Theta, h->bThout.p, 8 * B * m * m, hipMemcpyDeviceToHost, s));
    }
  }
  HIPCHK(hipStreamSynchronize(s));
  if (solve_time) for (size_t b = 0; b < B; ++b) solve_time[b] = h->last_solve_seconds;
  // a status still SLOW/TIME here has values (OMC.jl:1871-1877): feasible = true for every node
  return 0;
}

int omc_relax_batch(omc_instance* h, int B, const omc_relax_params* params, int cut_type, const int* L,
                    const double* cut_x, const double* cut_Uhat, const int8_t* cut_dir, const double* U_lower,
                    const double* U_upper, double* objective, double* dual_bound, int* status, int* iters, double* Y,
                    double* U, double* X, double* Theta, double* lambda_min, double* breakpoint_x, double* solve_time) {
  int rc = omc_relax_stage(h, B, params, cut_type, L, cut_x, cut_Uhat, cut_dir, U_lower, U_upper);
  if (rc) return rc;
  rc = omc_relax_solve(h);
  if (rc) return rc;
  return omc_relax_fetch(h, objective, dual_bound, status, iters, Y, U, X, Theta, lambda_min, breakpoint_x, solve_time);
}


// ---- Shor mode (rank 1): matrix_completion_SDP_relaxation with add_Shor_valid_inequalities = true ---------------------------------------
// OMC.jl:1503-1525, 1755-1779, 1838-1846; node.Shor_info lists OMC.jl:37-40.  Formulation: oracle/omc_oracle_shor.py, DESIGN.md 3.7.
int omc_set_shor_penalties(omc_instance* h, double rho, double r4, double r5) {
  if (!h) return fail(OMC_ERR_ARGUMENT, "handle is NULL");
  if (!(rho > 0.0) || !(r4 >= 0.0) || !(r5 > 0.0)) return fail(OMC_ERR_ARGUMENT, "rho and r5 must be positive, r4 non-negative (0 = automatic)");
  h->shor_rho = rho; h->shor_r4 = r4; h->shor_r5 = r5;
  return 0;
}

namespace {
struct ShorGroupHost {
  int nq = 0, nv1 = 0, nv2 = 0;
  std::vector<int> mi, kid, cptr, cent, v1ptr, v1ent, v2ptr, v2ent, slackrow;
  std::vector<uint8_t> eclass, ctype;
  size_t off_int = 0, off_byte = 0;       // offsets into the concatenated device buffers
};
// keys sorted, ids assigned in sorted order (as numpy.unique does in the oracle); members in increasing (minor, position) order
static void build_keys(const std::vector<uint64_t>& keyA, const std::vector<uint64_t>& keyB, int nq, int& nkeys, int* kidA, int* kidB,
                       std::vector<int>& ptr, std::vector<int>& ent) {
  std::vector<std::pair<uint64_t, int>> all((size_t)2 * nq);
  for (int q = 0; q < nq; ++q) { all[(size_t)2 * q] = {keyA[q], 2 * q}; all[(size_t)2 * q + 1] = {keyB[q], 2 * q + 1}; }
  std::sort(all.begin(), all.end());
  ptr.clear(); ent.resize((size_t)2 * nq);
  nkeys = 0;
  for (size_t e = 0; e < all.size(); ++e) {
    if (e == 0 || all[e].first != all[e - 1].first) { ptr.push_back((int)e); ++nkeys; }
    ent[e] = all[e].second;
    const int q = all[e].second >> 1;
    if (all[e].second & 1) kidB[q] = nkeys - 1; else kidA[q] = nkeys - 1;
  }
  ptr.push_back((int)all.size());
}
}  // namespace

int omc_relax_stage_shor(omc_instance* h, int B, const omc_relax_params* params, int cut_type, const int* L, const double* cut_x,
                         const double* cut_Uhat, const int8_t* cut_dir, const double* U_lower, const double* U_upper,
                         const int64_t* n_shor, const int64_t* shor_idx, const int64_t* n_soc, const int64_t* soc_idx) {
  if (!h) return fail(OMC_ERR_ARGUMENT, "handle is NULL");
  if (B <= 0) return fail(OMC_ERR_ARGUMENT, "B must be positive");
  if (!n_shor || !n_soc) return fail(OMC_ERR_ARGUMENT, "n_shor / n_soc is NULL");
  const int n = h->n, m = h->m;
  if ((long long)n * m >= (1ll << 30)) return fail(OMC_ERR_UNSUPPORTED, "Shor mode: n * m too large for the 32-bit index structures");
  // ---- group the nodes by identical lists ------------------------------------------------------------------------------------------
  std::vector<size_t> so(B + 1, 0), co(B + 1, 0);
  for (int b = 0; b < B; ++b) {
    if (n_shor[b] < 0 || n_soc[b] < -1) return fail(OMC_ERR_ARGUMENT, "negative list length");
    if (n_shor[b] >= (1ll << 28)) return fail(OMC_ERR_UNSUPPORTED, "Shor mode: more than 2^28 minors in one node");
    so[b + 1] = so[b] + (size_t)n_shor[b]; co[b + 1] = co[b] + (size_t)(n_soc[b] > 0 ? n_soc[b] : 0);
  }
  if (so[B] > 0 && !shor_idx) return fail(OMC_ERR_ARGUMENT, "shor_idx is NULL but n_shor > 0");
  if (co[B] > 0 && !soc_idx) return fail(OMC_ERR_ARGUMENT, "soc_idx is NULL but n_soc > 0");
  auto fnv = [](const void* p, size_t bytes, uint64_t hsh) { const uint8_t* c = (const uint8_t*)p; for (size_t e = 0; e < bytes; ++e) { hsh ^= c[e]; hsh *= 1099511628211ull; } return hsh; };
  std::unordered_map<uint64_t, std::vector<int>> byhash;      // hash -> group ids
  std::vector<int> rep;                                        // representative node of each group
  std::vector<int> node_group(B);
  for (int b = 0; b < B; ++b) {
    uint64_t hv = 1469598103934665603ull;
    hv = fnv(&n_shor[b], 8, hv); hv = fnv(&n_soc[b], 8, hv);
    hv = fnv(shor_idx + 4 * so[b], 32 * (size_t)n_shor[b], hv);
    if (n_soc[b] > 0) hv = fnv(soc_idx + 2 * co[b], 16 * (size_t)n_soc[b], hv);
    int gid = -1;
    for (int g : byhash[hv]) {
      const int r = rep[g];
      if (n_shor[r] == n_shor[b] && n_soc[r] == n_soc[b] && memcmp(shor_idx + 4 * so[r], shor_idx + 4 * so[b], 32 * (size_t)n_shor[b]) == 0 &&
          (n_soc[b] <= 0 || memcmp(soc_idx + 2 * co[r], soc_idx + 2 * co[b], 16 * (size_t)n_soc[b]) == 0)) { gid = g; break; }
    }
    if (gid < 0) { gid = (int)rep.size(); rep.push_back(b); byhash[hv].push_back(gid); }
    node_group[b] = gid;
  }
  const int NG = (int)rep.size();
  std::vector<ShorGroupHost> gh(NG);
  int nqmax = 0, nv1max = 0, nv2max = 0;
  size_t tot_int = 0, tot_byte = 0;
  for (int g = 0; g < NG; ++g) {
    ShorGroupHost& G = gh[g];
    const int b = rep[g];
    const int nq = (int)n_shor[b];
    G.nq = nq;
    G.mi.resize((size_t)4 * nq); G.kid.resize((size_t)4 * nq);
    std::vector<uint64_t> enc(nq), k1a(nq), k1b(nq), k2a(nq), k2b(nq);
    std::vector<int> cnt((size_t)n * m + 1, 0);
    for (int q = 0; q < nq; ++q) {
      const int64_t* t = shor_idx + 4 * (so[b] + q);
      const int64_t i1 = t[0] - 1, i2 = t[1] - 1, j1 = t[2] - 1, j2 = t[3] - 1;
      if (!(0 <= i1 && i1 < i2 && i2 < n && 0 <= j1 && j1 < j2 && j2 < m))
        return fail(OMC_ERR_ARGUMENT, "Shor minors must satisfy 1 <= i1 < i2 <= n, 1 <= j1 < j2 <= m (OMC.jl:2556-2603)");
      G.mi[q] = (int)i1; G.mi[(size_t)nq + q] = (int)i2; G.mi[(size_t)2 * nq + q] = (int)j1; G.mi[(size_t)3 * nq + q] = (int)j2;
      enc[q] = (((uint64_t)i1 * n + (uint64_t)i2) * m + (uint64_t)j1) * m + (uint64_t)j2;
      k1a[q] = ((uint64_t)i1 * m + j1) * m + j2; k1b[q] = ((uint64_t)i2 * m + j1) * m + j2;      // V1[i,(j1,j2)]
      k2a[q] = ((uint64_t)i1 * n + i2) * m + j1; k2b[q] = ((uint64_t)i1 * n + i2) * m + j2;      // V2[(i1,i2),j]
      ++cnt[(size_t)j1 * n + i1]; ++cnt[(size_t)j2 * n + i1]; ++cnt[(size_t)j1 * n + i2]; ++cnt[(size_t)j2 * n + i2];
    }
    { std::vector<uint64_t> se = enc; std::sort(se.begin(), se.end()); if (std::adjacent_find(se.begin(), se.end()) != se.end()) return fail(OMC_ERR_ARGUMENT, "duplicate Shor minor in a node's list"); }
    build_keys(k1a, k1b, nq, G.nv1, G.kid.data(), G.kid.data() + nq, G.v1ptr, G.v1ent);
    build_keys(k2a, k2b, nq, G.nv2, G.kid.data() + (size_t)2 * nq, G.kid.data() + (size_t)3 * nq, G.v2ptr, G.v2ent);
    // coordinate CSR: members in increasing (minor, position) order
    G.cptr.assign((size_t)n * m + 1, 0);
    for (size_t e = 0; e < (size_t)n * m; ++e) G.cptr[e + 1] = G.cptr[e] + cnt[e];
    G.cent.resize((size_t)4 * nq);
    { std::vector<int> fill(G.cptr.begin(), G.cptr.end() - 1);
      for (int q = 0; q < nq; ++q) {
        const int i1 = G.mi[q], i2 = G.mi[(size_t)nq + q], j1 = G.mi[(size_t)2 * nq + q], j2 = G.mi[(size_t)3 * nq + q];
        const size_t ce[4] = {(size_t)j1 * n + i1, (size_t)j2 * n + i1, (size_t)j1 * n + i2, (size_t)j2 * n + i2};
        for (int p = 0; p < 4; ++p) G.cent[fill[ce[p]]++] = 4 * q + p;
      } }
    // entry classes, column types, slack rows
    G.eclass.assign((size_t)n * m, 0);
    if (n_soc[b] < 0) { for (size_t e = 0; e < (size_t)n * m; ++e) G.eclass[e] = 1; }
    else for (int64_t c = 0; c < n_soc[b]; ++c) {
      const int64_t i = soc_idx[2 * (co[b] + c)] - 1, j = soc_idx[2 * (co[b] + c) + 1] - 1;
      if (!(0 <= i && i < n && 0 <= j && j < m)) return fail(OMC_ERR_ARGUMENT, "SOC coordinate out of range");
      G.eclass[(size_t)j * n + i] = 1;
    }
    for (size_t e = 0; e < (size_t)n * m; ++e) if (cnt[e] > 0) G.eclass[e] = 2;      // W >= X^2 is implied by the order-5 block there
    G.ctype.assign(m, 2); G.slackrow.assign(m, -1);
    for (int j = 0; j < m; ++j) {
      int first_nonC = -1, first_unobs = -1;
      for (int i = 0; i < n; ++i) {
        if (G.eclass[(size_t)j * n + i] == 2) continue;
        if (first_nonC < 0) first_nonC = i;
        if (first_unobs < 0 && !h->mask[(size_t)j * n + i]) first_unobs = i;
      }
      if (first_unobs >= 0) { G.ctype[j] = 0; G.slackrow[j] = first_unobs; }
      else if (first_nonC >= 0) { G.ctype[j] = 1; G.slackrow[j] = first_nonC; }
    }
    if (h->k > 1) {
      // Reference quirk Q5 (oracle/omc_oracle_shor.py header): in the rank k > 1 form (OMC.jl:1526-1551, 1780-1827) the slack that H
      // cancels in W = sum Wt + 2 sum H makes every per-layer order-5 block satisfiable, so the minors do not constrain (X, W); what remains of them
      // is W >= X^2 on their coordinates (implied by the order-(k+1) block).  The program solved is therefore the one without order-5 blocks and
      // with those coordinates on the SOC list; the lifted variables Xt, Wt, H, V of the result are an explicit extension (host side: api.py).
      for (size_t e = 0; e < (size_t)n * m; ++e) if (G.eclass[e] == 2) G.eclass[e] = 1;
      for (int j = 0; j < m; ++j) {
        int first_unobs = -1;
        for (int i = 0; i < n; ++i) if (!h->mask[(size_t)j * n + i]) { first_unobs = i; break; }
        if (first_unobs >= 0) { G.ctype[j] = 0; G.slackrow[j] = first_unobs; } else { G.ctype[j] = 1; G.slackrow[j] = 0; }
      }
      G.nq = 0; G.nv1 = 0; G.nv2 = 0;
      G.mi.clear(); G.kid.clear(); G.cent.clear(); G.v1ent.clear(); G.v2ent.clear();
      G.cptr.assign((size_t)n * m + 1, 0); G.v1ptr.assign(1, 0); G.v2ptr.assign(1, 0);
    }
    nqmax = std::max(nqmax, G.nq); nv1max = std::max(nv1max, G.nv1); nv2max = std::max(nv2max, G.nv2);
    G.off_int = tot_int;
    tot_int += G.mi.size() + G.kid.size() + G.cptr.size() + G.cent.size() + G.v1ptr.size() + G.v1ent.size() + G.v2ptr.size() + G.v2ent.size() + G.slackrow.size();
    G.off_byte = tot_byte;
    tot_byte += G.eclass.size() + G.ctype.size();
    tot_byte = (tot_byte + 15) & ~(size_t)15;
  }
  // ---- rank k > 1 with an unobserved entry in every column: the program without order-5 blocks IS the base relaxation (theta_j >= x_j' Y^+ x_j
  // implies theta_j >= ||x_j||^2 because Y <= I, and the slack of Theta_jj = sum_i W_ij sits on an unobserved entry at no cost): the base engine
  // solves it (no order-(n+m) cone), omc_relax_fetch_shor completes W = X^2 + slack from its (X, Theta).
  if (h->k > 1) {
    bool all_free = true;
    for (int g = 0; g < NG && all_free; ++g) for (int j = 0; j < m; ++j) if (gh[g].ctype[j] != 0) { all_free = false; break; }
    if (all_free && !h->tun.get("OMC_SHOR_EXPLICIT")) {
      int rc0 = omc_relax_stage(h, B, params, cut_type, L, cut_x, cut_Uhat, cut_dir, U_lower, U_upper);
      if (rc0) return rc0;
      h->shor_via_base = true;
      h->shor_slackrow = gh[0].slackrow;      // first unobserved row of every column (a property of the mask: the same for every list)
      return 0;
    }
  }
  // ---- base staging (rows, small cone, clip, certificate machinery) with the Shor flag ----------------------------------------------
  h->shor_req = true;
  int rc = omc_relax_stage(h, B, params, cut_type, L, cut_x, cut_Uhat, cut_dir, U_lower, U_upper);
  h->shor_req = false;
  if (rc) return rc;
  h->staged = false;
  OmcWS& w = h->ws;
  const int S = w.B, N = n + m;
  hipStream_t s = h->stream;
  // ---- upload the index structures ------------------------------------------------------------------------------------------------
  {
    std::vector<int> hi(tot_int ? tot_int : 1); std::vector<uint8_t> hb(tot_byte ? tot_byte : 16, 0);
    if ((rc = h->sgInts.ensure(hi.size() * sizeof(int)))) return rc;
    if ((rc = h->sgBytes.ensure(hb.size()))) return rc;
    std::vector<ShorGroupDev> gd(NG);
    for (int g = 0; g < NG; ++g) {
      ShorGroupHost& G = gh[g];
      size_t o = G.off_int;
      const int* base = h->sgInts.as<int>();
      auto put = [&](const std::vector<int>& v) { const int* dp = base + o; if (!v.empty()) memcpy(&hi[o], v.data(), v.size() * sizeof(int)); o += v.size(); return dp; };
      gd[g].nq = G.nq; gd[g].nv1 = G.nv1; gd[g].nv2 = G.nv2; gd[g].pad = 0;
      // the weight of the order-5 blocks on an entry of X / W is r4 x (blocks that hold it): r4 follows the mean multiplicity 4 nq / (n m)
      // (measured: 20 at 12 x 14 with 152 minors, 5 at 100 x 100 with 38 813, 1.2 at 200 x 200 with 632 732 certify fastest)
      gd[g].r4 = (h->shor_r4 > 0.0) ? h->shor_r4 : std::min(40.0, std::max(0.25, 75.0 * (double)n * (double)m / (4.0 * (double)std::max(G.nq, 1))));
      gd[g].mi = put(G.mi); gd[g].kid = put(G.kid); gd[g].cptr = put(G.cptr); gd[g].cent = put(G.cent);
      gd[g].v1ptr = put(G.v1ptr); gd[g].v1ent = put(G.v1ent); gd[g].v2ptr = put(G.v2ptr); gd[g].v2ent = put(G.v2ent); gd[g].slackrow = put(G.slackrow);
      memcpy(&hb[G.off_byte], G.eclass.data(), G.eclass.size());
      memcpy(&hb[G.off_byte + G.eclass.size()], G.ctype.data(), G.ctype.size());
      gd[g].eclass = h->sgBytes.as<uint8_t>() + G.off_byte; gd[g].ctype = gd[g].eclass + G.eclass.size();
    }
    HIPCHK(hipMemcpyAsync(h->sgInts.p, hi.data(), hi.size() * sizeof(int), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->sgBytes.p, hb.data(), hb.size(), hipMemcpyHostToDevice, s));
    if ((rc = upload(h->sgGroups, gd.data(), sizeof(ShorGroupDev) * NG, s))) return rc;
    if ((rc = upload(h->sgNodeGroup, node_group.data(), sizeof(int) * B, s))) return rc;
    HIPCHK(hipStreamSynchronize(s));          // the host vectors die with this block
  }
  // ---- scaling: the program is homogeneous of degree 2 in A (oracle: shor_scale) -------------------------------------------------------
  const double sc = sqrt(((double)h->nnz / ((double)n * m)) * (double)std::min(n, m) / std::max(h->sumA2, 1e-300));
  {
    std::vector<double> Ah((size_t)n * m);
    for (size_t e = 0; e < Ah.size(); ++e) Ah[e] = h->A[e] * sc;
    if ((rc = upload(h->sAh, Ah.data(), 8 * Ah.size(), s))) return rc;
    HIPCHK(hipStreamSynchronize(s));
  }
  // ---- state ---------------------------------------------------------------------------------------------------------------------------
  ShWS& sh = h->sh;
  memset(&sh, 0, sizeof(sh));
  const size_t sB = (size_t)S, sN = (size_t)B, nm = (size_t)n * m;
  const int NPb = (N + 15) & ~15;
  const int nmb = std::max(1, (nqmax + 255) / 256);
  const size_t nq1 = (size_t)std::max(nqmax, 1), nv11 = (size_t)std::max(nv1max, 1), nv21 = (size_t)std::max(nv2max, 1);
  ENS(h->sX, sB * nm * 8); ENS(h->sW, sB * nm * 8); ENS(h->sTh, sB * m * m * 8);
  ENS(h->sV1, sB * nv11 * 8); ENS(h->sV2, sB * nv21 * 8); ENS(h->sV3, sB * nq1 * 8);
  ENS(h->sD0, sB * N * N * 8); ENS(h->sP0, sB * N * N * 8); ENS(h->sMbufB, sB * NPb * NPb * 8); ENS(h->sVrowB, sB * NPb * NPb * 8);
  ENS(h->sTq, sB * 15 * nq1 * 8); ENS(h->sPq, sB * 15 * nq1 * 8); ENS(h->sNq, sB * 15 * nq1 * 8);
  ENS(h->sD5x, sB * nm * 8); ENS(h->sD5t, sB * m * 8); ENS(h->snu5, sB * m * 8); ENS(h->sP5x, sB * nm * 8);
  ENS(h->scolpart, sB * m * 4 * 8); ENS(h->sminpart, sB * nmb * 8); ENS(h->sminpart2, sB * nmb * 8);
  ENS(h->sfroB, sB * 2 * 8); ENS(h->svvB, sB * sizeof(int)); ENS(h->se1, sB * nv11 * 8); ENS(h->se2, sB * nv21 * 8);
  ENS(h->soX, sN * nm * 8); ENS(h->soW, sN * nm * 8); ENS(h->soTh, sN * m * m * 8);
  HIPCHK(hipMemsetAsync(h->sMbufB.p, 0, sB * NPb * NPb * 8, s));
  HIPCHK(hipMemsetAsync(h->sVrowB.p, 0, sB * NPb * NPb * 8, s));
  HIPCHK(hipMemsetAsync(h->sminpart.p, 0, sB * nmb * 8, s));
  HIPCHK(hipMemsetAsync(h->sminpart2.p, 0, sB * nmb * 8, s));
  HIPCHK(hipMemsetAsync(h->soX.p, 0, sN * nm * 8, s)); HIPCHK(hipMemsetAsync(h->soW.p, 0, sN * nm * 8, s)); HIPCHK(hipMemsetAsync(h->soTh.p, 0, sN * m * m * 8, s));
  sh.n = n; sh.m = m; sh.k = h->k; sh.N = N; sh.NPb = NPb; sh.S = S; sh.Btot = B; sh.nqmax = nqmax; sh.nv1max = nv1max; sh.nv2max = nv2max; sh.nmb = nmb;
  sh.rx = w.relax; sh.r4 = h->shor_r4; sh.r5 = h->shor_r5; sh.gamma = h->gamma; sh.sc = sc;
  sh.groups = h->sgGroups.as<ShorGroupDev>(); sh.node_group = h->sgNodeGroup.as<int>();
  sh.node_of = w.node_of; sh.done = w.done; sh.init = w.init; sh.fin = w.fin; sh.rho_b = w.rho_b; sh.bfac = w.bfac;
  sh.Ah = h->sAh.as<double>(); sh.mask = h->dmask.as<uint8_t>();
  sh.X = h->sX.as<double>(); sh.W = h->sW.as<double>(); sh.Th = h->sTh.as<double>();
  sh.V1 = h->sV1.as<double>(); sh.V2 = h->sV2.as<double>(); sh.V3 = h->sV3.as<double>();
  sh.D0 = h->sD0.as<double>(); sh.P0 = h->sP0.as<double>(); sh.MbufB = h->sMbufB.as<double>(); sh.VrowB = h->sVrowB.as<double>();
  sh.Tq = h->sTq.as<double>(); sh.Pq = h->sPq.as<double>(); sh.Nq = h->sNq.as<double>();
  sh.D5x = h->sD5x.as<double>(); sh.D5t = h->sD5t.as<double>(); sh.nu5 = h->snu5.as<double>(); sh.P5x = h->sP5x.as<double>();
  sh.colpart = h->scolpart.as<double>(); sh.minpart = h->sminpart.as<double>(); sh.minpart2 = h->sminpart2.as<double>();
  sh.fro2B = h->sfroB.as<double>(); sh.trB = h->sfroB.as<double>() + sB; sh.vvalidB = h->svvB.as<int>();
  sh.Y = w.Y; sh.Yp = w.Yp; sh.rp = w.rp; sh.rd = w.rd;
  sh.e1 = h->se1.as<double>(); sh.e2 = h->se2.as<double>();
  sh.objcol = w.objcol; sh.c0col = w.c0col; sh.lamDX = w.lamDX;
  sh.oX = h->soX.as<double>(); sh.oW = h->soW.as<double>(); sh.oTh = h->soTh.as<double>();
  sh.oV = nullptr;
  if (h->shor_keep_V) { ENS(h->soV, sN * 5 * nq1 * 8); HIPCHK(hipMemsetAsync(h->soV.p, 0, sN * 5 * nq1 * 8, s)); sh.oV = h->soV.as<double>(); }
  // ---- base workspace in Shor mode, and the view through which its eigen-kernels project the order-(n+m) cone -------------------------
  w.shor = 1; w.shN = N; w.shP0 = sh.P0; w.shD0 = sh.D0; w.inv_s2 = 1.0 / (sc * sc);
  OmcWS& wb = h->wbig;
  wb = w;
  wb.n = N; wb.np16 = NPb; wb.Mbuf = sh.MbufB; wb.Vrow = sh.VrowB; wb.vvalid = sh.vvalidB; wb.fro2 = sh.fro2B; wb.W1 = sh.P0;
  wb.sub_enable = 0; wb.cert_enable = 0; wb.ws_mode = 0; wb.clip_hi = 1e300; wb.sub_debug = 0; wb.ws_first = nullptr; wb.ws_phase = 0;
  {
    const int Np2 = (N + 1) & ~1;
    int lpp = 16; while (lpp > 4 && lpp * (Np2 / 2) > 512) lpp >>= 1;
    int rpl = (((N + lpp - 1) / lpp) + 1) & ~1, Nrp = rpl * lpp;
    int ldw = Nrp + ((16 - (Nrp & 31)) & 31);
    if (((size_t)Np2 * ldw + 3 * Np2) * 8 + (size_t)(Np2 + 2) * 4 + 64 > OMC_MAX_DYN_LDS) ldw = Nrp + 2;
    h->big_lds = ((size_t)Np2 * ldw + 3 * Np2) * 8 + (size_t)(Np2 + 2) * 4 + 64;
    h->big_use_lds = h->big_lds <= OMC_MAX_DYN_LDS;
    wb.ws_ld = ldw;
    size_t need = 0;
    if (!h->big_use_lds) {
      lpp = 16; rpl = (((N + 15) / 16) + 1) & ~1; Nrp = rpl * 16; ldw = Nrp + 2; wb.ws_ld = ldw;
      if (rpl > 32 && N <= 1024) { lpp = 64; rpl = (((N + 63) / 64) + 1) & ~1; Nrp = rpl * 64; ldw = Nrp + 2; wb.ws_ld = ldw; }
      need = ((size_t)Np2 * ldw + 3 * Np2) * 8 + (size_t)(Np2 + 2) * 4 + 64;
    }
    h->big_lpp = (rpl <= 32 && h->tun.get("OMC_NO_WARMSTART") == nullptr) ? lpp : 0;
    // generic cold kernel (orders beyond the warm-started one): Np x (Np | 1) matrix + 2 Np doubles + Np ints
    const int ldc = Np2 | 1;
    h->big_cone_lds = ((size_t)Np2 * ldc + 2 * Np2) * 8 + (size_t)Np2 * 4 + 16;
    h->big_cone_lds_ok = h->big_cone_lds <= OMC_MAX_DYN_LDS;
    if (!h->big_lpp && !h->big_cone_lds_ok) need = std::max(need, h->big_cone_lds);
    if (need) {
      wb.cone_scratch_stride = need / 8 + 8;
      ENS(h->sbigscr, sB * wb.cone_scratch_stride * 8);
      wb.cone_scratch = h->sbigscr.as<double>();
    }
  }
  // the big cone's input has a handful of positive eigenvalues once the iterate has settled (measured: 2 - 5 of n + m): the tracked-subspace
  // kernel of the base engine follows them; the warm-started full kernel seeds the block and is the fall-back
  wb.trM = sh.trB;
  if (h->big_lpp && N >= 48 && omc_cone_sub_lds(NPb) <= OMC_MAX_DYN_LDS && !h->tun.get("OMC_NO_SUBSPACE") && !h->tun.get("OMC_SHOR_NO_SUBSPACE")) {
    ENS(h->sXsB, sB * NPb * 16 * 8); ENS(h->ssubSB, sB * 16 * 8); ENS(h->ssubIB, sB * 12 * sizeof(int));
    HIPCHK(hipMemsetAsync(h->sXsB.p, 0, sB * NPb * 16 * 8, s));
    HIPCHK(hipMemsetAsync(h->ssubSB.p, 0, sB * 16 * 8, s));
    HIPCHK(hipMemsetAsync(h->ssubIB.p, 0, sB * 12 * sizeof(int), s));
    int* si = h->ssubIB.as<int>();
    sh.sub_onB = si; sh.cone_doneB = si + sB; sh.sub_waitB = si + 2 * sB; sh.sub_nfailB = si + 3 * sB;
    wb.sub_enable = 1; wb.Xs = h->sXsB.as<double>(); wb.sub_theta = h->ssubSB.as<double>();
    wb.sub_zscratch = nullptr;
    if (NPb > 512) { ENS(h->bsubz, sB * 16 * (size_t)(NPb + 2) * 8); wb.sub_zscratch = h->bsubz.as<double>(); }
    wb.ws_first = nullptr; wb.ws_phase = 0; wb.sub_on = sh.sub_onB; wb.cone_done = sh.cone_doneB; wb.sub_wait = sh.sub_waitB; wb.sub_nfail = sh.sub_nfailB; wb.sub_stat = si + 4 * sB;
    wb.sep_done = nullptr;
  }
  HIPCHK(hipStreamSynchronize(s));
  h->shor_on = true;
  h->staged = true;
  return 0;
}

int omc_set_shor_keep_V(omc_instance* h, int keep) {
  if (!h) return fail(OMC_ERR_ARGUMENT, "handle is NULL");
  h->shor_keep_V = keep != 0;
  return 0;
}

int omc_relax_fetch_shor_V(omc_instance* h, double* V) {
  if (!h || !h->staged || !h->shor_on) return fail(OMC_ERR_ARGUMENT, "omc_relax_fetch_shor_V: no Shor-mode batch staged");
  if (!h->sh.oV) return fail(OMC_ERR_ARGUMENT, "omc_relax_fetch_shor_V: V was not kept (call omc_set_shor_keep_V(h, 1) before staging)");
  if (!V) return fail(OMC_ERR_ARGUMENT, "V is NULL");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(V, h->sh.oV, 8 * (size_t)h->sh.Btot * 5 * (size_t)std::max(h->sh.nqmax, 1), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int omc_last_shor_subspace_stats(omc_instance* h, int64_t* out) {
  if (!h || !out) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  for (int q = 0; q < 8; ++q) out[q] = h->big_sub_tot[q];
  return 0;
}

int omc_relax_fetch_shor(omc_instance* h, double* W) {
  if (h && h->staged && h->shor_via_base) {      // rank k > 1 served by the base engine: W = X^2, the slack Theta_jj - ||x_j||^2 on the column's first unobserved row
    if (!W) return 0;
    const size_t B = h->ws.Btot, n = h->n, m = h->m;
    std::vector<double> X(B * n * m), Th(B * m * m);
    int rc = omc_relax_fetch(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, X.data(), Th.data(), nullptr, nullptr, nullptr);
    if (rc) return rc;
    for (size_t b = 0; b < B; ++b)
      for (size_t j = 0; j < m; ++j) {
        double s2 = 0.0;
        for (size_t i = 0; i < n; ++i) { const double x = X[(b * m + j) * n + i]; W[(b * m + j) * n + i] = x * x; s2 += x * x; }
        W[(b * m + j) * n + h->shor_slackrow[j]] += Th[(b * m + j) * m + j] - s2;
      }
    return 0;
  }
  if (!h || !h->staged || !h->shor_on) return fail(OMC_ERR_ARGUMENT, "omc_relax_fetch_shor: no Shor-mode batch staged");
  HIPCHK(hipSetDevice(h->device));
  if (W) HIPCHK(hipMemcpyAsync(W, h->sh.oW, 8 * (size_t)h->sh.Btot * h->n * h->m, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int omc_relax_batch_shor(omc_instance* h, int B, const omc_relax_params* params, int cut_type, const int* L, const double* cut_x,
                         const double* cut_Uhat, const int8_t* cut_dir, const double* U_lower, const double* U_upper,
                         const int64_t* n_shor, const int64_t* shor_idx, const int64_t* n_soc, const int64_t* soc_idx, double* objective,
                         double* dual_bound, int* status, int* iters, double* Y, double* U, double* X, double* Theta, double* W,
                         double* lambda_min, double* breakpoint_x, double* solve_time) {
  int rc = omc_relax_stage_shor(h, B, params, cut_type, L, cut_x, cut_Uhat, cut_dir, U_lower, U_upper, n_shor, shor_idx, n_soc, soc_idx);
  if (rc) return rc;
  rc = omc_relax_solve(h);
  if (rc) return rc;
  rc = omc_relax_fetch(h, objective, dual_bound, status, iters, Y, U, X, Theta, lambda_min, breakpoint_x, solve_time);
  if (rc) return rc;
  return omc_relax_fetch_shor(h, W);
}

int omc_evaluate_objective(omc_instance* h, int B, const double* X, double* objective) {
  if (!h || !X || !objective) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  if (B <= 0) return fail(OMC_ERR_ARGUMENT, "B must be positive");
  HIPCHK(hipSetDevice(h->device));
  const size_t n = h->n, m = h->m;
  int r_ = h->bXin.ensure(8 * (size_t)B * n * m + 8 * (size_t)B + 64); if (r_) return r_;
  double* dX = h->bXin.as<double>();
  double* dout = dX + (size_t)B * n * m;
  HIPCHK(hipMemcpyAsync(dX, X, 8 * (size_t)B * n * m, hipMemcpyHostToDevice, h->stream));
  omc_launch_eval_objective(B, (int)n, (int)m, h->gamma, h->dA.as<double>(), h->dmask.as<uint8_t>(), dX, dout, h->stream);
  HIPCHK(hipMemcpyAsync(objective, dout, 8 * (size_t)B, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int omc_separation_batch(omc_instance* h, int B, int breakpoints, const double* Y, const double* U, double* eigvals,
                         double* x, int* feasible) {
  if (!h || !Y || !U) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  if (breakpoints != OMC_SMALLEST_1_EIGVEC && breakpoints != OMC_SMALLEST_2_EIGVEC)
    return fail(OMC_ERR_INVALID_ENUM, "Invalid input for disjunctive cuts breakpoints (OMC.jl:2440-2446)");
  // stage an empty batch to get a workspace of the right size, then overwrite (Y, U)
  std::vector<int> L(B, 0);
  omc_relax_params P = h->params; P.breakpoints = breakpoints; P.slots = B;   // identity slot map: the state arrays are addressed by node
  int rc = omc_relax_stage(h, B, &P, OMC_CUT_LINEAR, L.data(), nullptr, nullptr, nullptr, nullptr, nullptr);
  if (rc) return rc;
  const OmcWS& w = h->ws;
  const size_t n = h->n, k = h->k;
  HIPCHK(hipMemcpyAsync(w.Y, Y, 8 * (size_t)B * n * n, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(w.U, U, 8 * (size_t)B * n * k, hipMemcpyHostToDevice, h->stream));
  omc_launch_cone(&w, CONE_SEP, h->cone_use_lds, h->cone_lds, h->stream);
  std::vector<double> ev(2 * (size_t)B);
  HIPCHK(hipMemcpyAsync(ev.data(), w.lmin, 16 * (size_t)B, hipMemcpyDeviceToHost, h->stream));
  if (x) HIPCHK(hipMemcpyAsync(x, w.bx, 8 * (size_t)B * n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  for (int b = 0; b < B; ++b) {
    if (eigvals) { eigvals[2 * b] = ev[2 * b]; eigvals[2 * b + 1] = ev[2 * b + 1]; }
    if (feasible) feasible[b] = ev[2 * b] >= -1e-6 ? 1 : 0;   // OMC.jl:1274-1276 projection_tolerance
  }
  h->staged = false;
  return 0;
}

int omc_round_Y_batch(omc_instance* h, int B, const double* Y, double* U_rounded) {
  if (!h || !Y || !U_rounded) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  std::vector<int> L(B, 0);
  omc_relax_params P = h->params; P.slots = B;
  int rc = omc_relax_stage(h, B, &P, OMC_CUT_LINEAR, L.data(), nullptr, nullptr, nullptr, nullptr, nullptr);
  if (rc) return rc;
  const OmcWS& w = h->ws;
  const size_t n = h->n, k = h->k;
  HIPCHK(hipMemcpyAsync(w.Y, Y, 8 * (size_t)B * n * n, hipMemcpyHostToDevice, h->stream));
  omc_launch_cone(&w, CONE_TOPK, h->cone_use_lds, h->cone_lds, h->stream);
  HIPCHK(hipMemcpyAsync(U_rounded, w.U, 8 * (size_t)B * n * k, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  h->staged = false;
  return 0;
}

// svd(X).U[:, 1:k] of B matrices X (n x m): the k dominant eigenvectors of X X' (Gram product on the matrix cores, then the eigen-kernel of
// omc_round_Y_batch); same canonical sign.  The singular values are well separated from zero for the matrices the driver rounds (products U V of
// rank k, OMC.jl:564, 921; the zero-filled A of the root, OMC.jl:524), so squaring the condition number costs nothing that matters.
int omc_left_singular_batch(omc_instance* h, int B, const double* X, double* U_out) {
  if (!h || !X || !U_out) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  if (B <= 0) return fail(OMC_ERR_ARGUMENT, "B must be positive");
  std::vector<int> L(B, 0);
  omc_relax_params P = h->params; P.slots = B;
  int rc = omc_relax_stage(h, B, &P, OMC_CUT_LINEAR, L.data(), nullptr, nullptr, nullptr, nullptr, nullptr);
  if (rc) return rc;
  const OmcWS& w = h->ws;
  const size_t n = h->n, m = h->m, k = h->k;
  ENS(h->bXin, (size_t)B * n * m * 8);
  HIPCHK(hipMemcpyAsync(h->bXin.p, X, 8 * (size_t)B * n * m, hipMemcpyHostToDevice, h->stream));
  omc_launch_gram_XXt(&w, h->bXin.as<double>(), B, h->stream);
  omc_launch_cone(&w, CONE_TOPK, h->cone_use_lds, h->cone_lds, h->stream);
  HIPCHK(hipMemcpyAsync(U_out, w.U, 8 * (size_t)B * n * k, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipGetLastError());
  h->staged = false;
  return 0;
}

int omc_altmin_batch(omc_instance* h, int B, int cut_type, int reference_quirk_q1, const int* L, const double* cut_x,
                     const double* cut_Uhat, const int8_t* cut_dir, const double* U_initial, double eps, int max_iters,
                     double time_limit, double* U, double* V, int* converged, int* n_iters, double* objectives,
                     double* solve_time) {
  if (!h || !U_initial || !U || !V) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  if (B <= 0 || max_iters <= 0) return fail(OMC_ERR_ARGUMENT, "B and max_iters must be positive");
  if (cut_type != OMC_CUT_LINEAR && cut_type != OMC_CUT_LINEAR2 && cut_type != OMC_CUT_LINEAR3)
    return fail(OMC_ERR_INVALID_ENUM, "Invalid input for disjunctive cuts type (OMC.jl:1456-1462)");
  if (!(time_limit > 0.0)) {      // OMC.jl:2186-2189: the loop condition fails at once -- nothing is solved (a launch runs <= max_iters iterations in milliseconds, so this is the only case in which the limit can bind)
    const size_t nk = (size_t)h->n * h->k, mk = (size_t)h->m * h->k;
    for (int b = 0; b < B; ++b) { if (converged) converged[b] = 0; if (n_iters) n_iters[b] = 0; if (solve_time) solve_time[b] = 0.0; }
    memset(U, 0, sizeof(double) * nk * B); memset(V, 0, sizeof(double) * mk * B);
    if (objectives) for (size_t e = 0; e < (size_t)B * max_iters; ++e) objectives[e] = NAN;
    h->amobj_B = 0;
    return 0;
  }
  if (B <= 0 || max_iters <= 0) return fail(OMC_ERR_ARGUMENT, "B and max_iters must be positive");
  if (cut_type != OMC_CUT_LINEAR && cut_type != OMC_CUT_LINEAR2 && cut_type != OMC_CUT_LINEAR3)
    return fail(OMC_ERR_INVALID_ENUM, "Invalid input for disjunctive cuts type (OMC.jl:1456-1462)");
  if (h->k > 4) return fail(OMC_ERR_UNSUPPORTED, "omc_altmin_batch: rank k > 4 is not supported");
  HIPCHK(hipSetDevice(h->device));
  const int n = h->n, m = h->m, k = h->k;
  auto t0 = std::chrono::steady_clock::now();
  int Lmax = 1; long Ltot = 0;
  for (int b = 0; b < B; ++b) { int Lb = L ? L[b] : 0; if (Lb < 0) return fail(OMC_ERR_ARGUMENT, "negative cut count"); Lmax = std::max(Lmax, Lb); Ltot += Lb; }
  if (Ltot > 0 && (!cut_x || !cut_Uhat || !cut_dir)) return fail(OMC_ERR_ARGUMENT, "cut arrays are NULL but L > 0");
  // rows of model_U: box entries not implied by ||U_j|| <= 1 (defaults OMC.jl:1989-1996: U[n-k+j.., j] >= 0), per-cut bounds on
  // every column (2047-2093)
  const int Rmax = k * (k + 1) / 2 + 2 * k * Lmax;
  const size_t sB = (size_t)B;
  std::vector<int> hR(B, 0), hk(sB * Rmax, 0), hc(sB * Rmax, 0), hbi(sB * Rmax, 0), hbj(sB * Rmax, 0);
  std::vector<double> hcoef(sB * Rmax, 0.0), hrhs(sB * Rmax, 0.0), hx(sB * Lmax * n, 0.0);
  long cutbase = 0;
  for (int b = 0; b < B; ++b) {
    int r = 0;
    auto add = [&](int kind, int cut, int bi, int bj, double cf, double rhs) {
      hk[(size_t)b * Rmax + r] = kind; hc[(size_t)b * Rmax + r] = cut; hbi[(size_t)b * Rmax + r] = bi; hbj[(size_t)b * Rmax + r] = bj;
      hcoef[(size_t)b * Rmax + r] = cf; hrhs[(size_t)b * Rmax + r] = rhs; ++r;
    };
    for (int j = 0; j < k; ++j)
      for (int i = n - k + j; i < n; ++i) add(ROW_BOX, -1, i, j, -1.0, 0.0);     // -U[i,j] <= 0  (symmetry breaking, OMC.jl:1991-1993)
    const int Lb = L ? L[b] : 0;
    for (int l = 0; l < Lb; ++l) {
      const double* x = cut_x + (size_t)(cutbase + l) * n;
      for (int j = 0; j < k; ++j) {
        const double* Uh = cut_Uhat + ((size_t)(cutbase + l) * k + j) * n;
        double vhat = 0.0;
        for (int i = 0; i < n; ++i) vhat += Uh[i] * x[i];      // OMC.jl:2053
        double lo, hi, sl, ic;
        if (cut_piece(cut_type, cut_dir[(size_t)(cutbase + l) * k + j], vhat, reference_quirk_q1, &lo, &hi, &sl, &ic))
          return fail(OMC_ERR_INVALID_ENUM, "direction code invalid for this cut type (OMC.jl:2056-2091)");
        add(ROW_BOUND, l, -1, j, 1.0, hi);
        add(ROW_BOUND, l, -1, j, -1.0, -lo);
      }
      memcpy(&hx[((size_t)b * Lmax + l) * n], x, sizeof(double) * n);
    }
    hR[b] = r; cutbase += Lb;
  }
  int rc_ = 0;
  hipStream_t s = h->stream;
  if ((rc_ = upload(h->aR, hR.data(), sizeof(int) * B, s))) return rc_;
  if ((rc_ = upload(h->arkind, hk.data(), sizeof(int) * hk.size(), s))) return rc_;
  if ((rc_ = upload(h->arcut, hc.data(), sizeof(int) * hc.size(), s))) return rc_;
  if ((rc_ = upload(h->arbi, hbi.data(), sizeof(int) * hbi.size(), s))) return rc_;
  if ((rc_ = upload(h->arbj, hbj.data(), sizeof(int) * hbj.size(), s))) return rc_;
  if ((rc_ = upload(h->arcoef, hcoef.data(), sizeof(double) * hcoef.size(), s))) return rc_;
  if ((rc_ = upload(h->arrhs, hrhs.data(), sizeof(double) * hrhs.size(), s))) return rc_;
  if ((rc_ = upload(h->acutx, hx.data(), sizeof(double) * hx.size(), s))) return rc_;
  if ((rc_ = upload(h->aU0, U_initial, sizeof(double) * sB * n * k, s))) return rc_;
  if ((rc_ = h->aU.ensure(8 * sB * n * k))) return rc_;
  if ((rc_ = h->aV.ensure(8 * sB * m * k))) return rc_;
  if ((rc_ = h->aobj.ensure(8 * sB * max_iters))) return rc_;
  if ((rc_ = h->aint.ensure(4 * sB * 2))) return rc_;
  if ((rc_ = h->amobj.ensure(8 * sB))) return rc_;
  h->amobj_B = B;
  if ((rc_ = h->aG.ensure(8 * sB * Rmax * Rmax))) return rc_;
  AltminWS w{};
  w.B = B; w.n = n; w.m = m; w.k = k; w.Rmax = Rmax; w.Lmax = Lmax; w.max_iters = max_iters;
  w.gamma = h->gamma; w.eps = eps; w.sumA2 = h->sumA2;
  w.col_ptr = h->dcol_ptr.as<int>(); w.col_idx = h->dcol_idx.as<int>(); w.col_val = h->dcol_val.as<double>();
  w.row_ptr = h->drow_ptr.as<int>(); w.row_idx = h->drow_idx.as<int>(); w.row_val = h->drow_val.as<double>();
  w.R = h->aR.as<int>(); w.rkind = h->arkind.as<int>(); w.rcut = h->arcut.as<int>(); w.rbi = h->arbi.as<int>(); w.rbj = h->arbj.as<int>();
  w.rcoef = h->arcoef.as<double>(); w.rrhs = h->arrhs.as<double>(); w.cutx = h->acutx.as<double>();
  w.U0 = h->aU0.as<double>(); w.U = h->aU.as<double>(); w.V = h->aV.as<double>(); w.objectives = h->aobj.as<double>();
  w.converged = h->aint.as<int>(); w.n_iters = h->aint.as<int>() + B; w.G = h->aG.as<double>(); w.mobj = h->amobj.as<double>();
  const size_t lds = (k == 1) ? ((size_t)4 * n + m + 2 * Rmax + 8) * 8
                              : ((size_t)4 * n * k + (size_t)2 * n * k * k + (size_t)k * m + 2 * Rmax + 8) * 8;
  w.scratch = nullptr; w.scratch_stride = 0;
  size_t lds_launch = lds;
  if (lds + 8 * 1024 > ((k == 1) ? (size_t)OMC_MAX_DYN_LDS - 8 * 1024 : (size_t)128 * 1024) || h->tun.get("OMC_ALTMIN_NOLDS")) {
    // the problem does not fit the LDS (config 5: 1000 x 1000, k = 2 needs 144 KB): the same kernel runs on a per-problem global slab
    w.scratch_stride = lds / 8 + 8;
    ENS(h->aG2, (size_t)B * w.scratch_stride * 8);
    w.scratch = h->aG2.as<double>();
    lds_launch = 0;
  }
  if (k == 1) omc_launch_altmin(&w, lds_launch, s); else omc_launch_altmin_k(&w, lds_launch, s);
  HIPCHK(hipMemcpyAsync(U, w.U, 8 * sB * n * k, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(V, w.V, 8 * sB * m * k, hipMemcpyDeviceToHost, s));
  if (objectives) HIPCHK(hipMemcpyAsync(objectives, w.objectives, 8 * sB * max_iters, hipMemcpyDeviceToHost, s));
  if (converged) HIPCHK(hipMemcpyAsync(converged, w.converged, 4 * sB, hipMemcpyDeviceToHost, s));
  if (n_iters) HIPCHK(hipMemcpyAsync(n_iters, w.n_iters, 4 * sB, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (solve_time) for (int b = 0; b < B; ++b) solve_time[b] = el;
  return 0;
}

int omc_altmin_master_objectives(omc_instance* h, int B, double* objective) {
  if (!h || !objective) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  if (B <= 0 || B != h->amobj_B) return fail(OMC_ERR_ARGUMENT, "B does not match the last omc_altmin_batch call");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(objective, h->amobj.p, 8 * (size_t)B, hipMemcpyDeviceToHost, h->stream));      // never the legacy stream: another handle may be capturing a graph
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Shor minors: generate_rank1_matrix_completion_Shor_constraints_indexes (OMC.jl:2545-2612) and
// generate_violated_Shor_minors (OMC.jl:2614-2640).  Kernels in omc_shor.hip; the host only sequences the segments.
// ---------------------------------------------------------------------------------------------------------------------
struct ShorSeg { int kind, la, lb; };
// two timing events that are destroyed on every exit path
struct EventPair {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int create() { if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1; return 0; }
  ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};
static void shor_segments(int n_classes, const int* classes, bool dedup, std::vector<ShorSeg>& segs) {
  bool seen[5] = {false, false, false, false, false};
  for (int c = 0; c < n_classes; ++c) {
    const int p = classes[c];
    if (p < 0 || p > 4) continue;                 // the reference's if/elseif chain has no branch for other values
    if (dedup) { if (seen[p]) continue; seen[p] = true; }
    if (p == 4) segs.push_back({SH_COMBO, SH_BOTH, 0});                                       // OMC.jl:2556-2561
    else if (p == 3) segs.push_back({SH_PRODUCT, SH_BOTH, SH_XOR});                           // 2562-2569
    else if (p == 2) { segs.push_back({SH_PRODUCT, SH_BOTH, SH_NONE}); segs.push_back({SH_COMBO, SH_XOR, 0}); }  // 2570-2584
    else if (p == 1) segs.push_back({SH_PRODUCT, SH_XOR, SH_NONE});                           // 2585-2596
    else segs.push_back({SH_COMBO, SH_NONE, 0});                                              // 2597-2608
  }
}
static int shor_prepare(omc_instance* h) {
  if (h->shor_ready) return 0;
  const int n = h->n, m = h->m;
  if (m > 8192) return fail(OMC_ERR_UNSUPPORTED, "Shor enumeration supports m <= 8192 in this build");
  HIPCHK(hipSetDevice(h->device));
  const int W = (m + 63) / 64;
  std::vector<uint64_t> bits((size_t)n * W, 0);
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < n; ++i)
      if (h->mask[(size_t)j * n + i]) bits[(size_t)i * W + (j >> 6)] |= (1ULL << (j & 63));
  const long long np = (long long)n * (n - 1) / 2;
  int rc;
  if ((rc = upload(h->sbits, bits.data(), bits.size() * 8, h->stream))) return rc;
  if ((rc = h->scb.ensure((size_t)std::max(np, 1LL) * 4)) || (rc = h->scx.ensure((size_t)std::max(np, 1LL) * 4)) ||
      (rc = h->scz.ensure((size_t)std::max(np, 1LL) * 4)) || (rc = h->soff.ensure((size_t)std::max(np, 1LL) * 8)) ||
      (rc = h->stot.ensure(64)) || (rc = h->shist.ensure(256 * 8)) || (rc = h->scnt.ensure(64)))
    return rc;
  omc_shor_launch_pair_counts(n, m, W, h->sbits.as<uint64_t>(), np, h->scb.as<int>(), h->scx.as<int>(), h->scz.as<int>(), h->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));   // `bits` is a host temporary
  h->shor_W = W; h->shor_pairs = np; h->shor_ready = true;
  return 0;
}
// scan segment `sg` with offsets starting at `base`; returns its total
static int shor_scan(omc_instance* h, const ShorSeg& sg, long long base, long long* total) {
  omc_shor_launch_seg_scan(sg.kind, sg.la, sg.lb, h->scb.as<int>(), h->scx.as<int>(), h->scz.as<int>(), h->shor_pairs, base,
                           h->soff.as<long long>(), h->stot.as<long long>(), h->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(total, h->stot.p, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int omc_shor_count(omc_instance* h, int n_classes, const int* num_entries_present, int64_t* count_per_class) {
  if (!h || (n_classes > 0 && !num_entries_present) || !count_per_class) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  int rc = shor_prepare(h);
  if (rc) return rc;
  for (int c = 0; c < n_classes; ++c) {
    std::vector<ShorSeg> segs;
    shor_segments(1, num_entries_present + c, false, segs);
    long long tot = 0;
    for (const ShorSeg& sg : segs) { long long t = 0; if ((rc = shor_scan(h, sg, 0, &t))) return rc; tot += t; }
    count_per_class[c] = tot;
  }
  return 0;
}

int omc_shor_indexes(omc_instance* h, int n_classes, const int* num_entries_present, int64_t capacity, int64_t* out, int64_t* count) {
  if (!h || (n_classes > 0 && !num_entries_present) || !count) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  int rc = shor_prepare(h);
  if (rc) return rc;
  std::vector<ShorSeg> segs;
  shor_segments(n_classes, num_entries_present, false, segs);
  std::vector<long long> tot(segs.size(), 0);
  long long total = 0;
  for (size_t q = 0; q < segs.size(); ++q) { if ((rc = shor_scan(h, segs[q], 0, &tot[q]))) return rc; total += tot[q]; }
  *count = total;
  if (!out || capacity < total || total == 0) return 0;      // size query (or nothing to write)
  if ((rc = h->sout.ensure((size_t)total * 32))) return rc;
  EventPair ev;
  if (ev.create()) return fail(999, "hipEventCreate failed");
  hipEvent_t e0 = ev.e0, e1 = ev.e1;
  HIPCHK(hipEventRecord(e0, h->stream));
  long long base = 0;
  for (size_t q = 0; q < segs.size(); ++q) {
    long long t = 0;
    if ((rc = shor_scan(h, segs[q], base, &t))) return rc;
    omc_shor_launch_enum_tuples(h->n, h->m, h->shor_W, h->sbits.as<uint64_t>(), segs[q].kind, segs[q].la, segs[q].lb,
                                h->soff.as<long long>(), h->shor_pairs, h->sout.as<long long>(), h->stream);
    HIPCHK(hipGetLastError());
    base += t;
  }
  HIPCHK(hipEventRecord(e1, h->stream));
  HIPCHK(hipMemcpyAsync(out, h->sout.p, (size_t)total * 32, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  h->shor_last_ms = ms; h->shor_last_candidates = total;
  return 0;
}

int omc_violated_shor_minors(omc_instance* h, const double* X, int n_classes, const int* num_entries_present, int64_t n_existing,
                             const int64_t* existing, int n_minors, double* scores, int64_t* minors, int* n_out) {
  if (!h || !X || (n_classes > 0 && !num_entries_present) || (n_existing > 0 && !existing) || !n_out) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  if (n_minors < 0 || n_existing < 0) return fail(OMC_ERR_ARGUMENT, "negative count");
  if (n_minors > 0 && (!scores || !minors)) return fail(OMC_ERR_ARGUMENT, "NULL output");
  *n_out = 0;
  int rc = shor_prepare(h);
  if (rc) return rc;
  const int n = h->n, m = h->m, k = h->k;
  std::vector<ShorSeg> segs;
  shor_segments(n_classes, num_entries_present, true, segs);   // setdiff! also removes duplicates (OMC.jl:2626)
  std::vector<long long> tot(segs.size(), 0);
  long long N = 0;
  for (size_t q = 0; q < segs.size(); ++q) { if ((rc = shor_scan(h, segs[q], 0, &tot[q]))) return rc; N += tot[q]; }
  if (N == 0 || n_minors == 0) return 0;
  // keys of the existing constraints, sorted and unique
  std::vector<uint64_t> ex; ex.reserve((size_t)n_existing);
  for (int64_t e = 0; e < n_existing; ++e) {
    const int64_t i1 = existing[4 * e] - 1, i2 = existing[4 * e + 1] - 1, j1 = existing[4 * e + 2] - 1, j2 = existing[4 * e + 3] - 1;
    if (i1 < 0 || i1 >= n || i2 < 0 || i2 >= n || j1 < 0 || j1 >= m || j2 < 0 || j2 >= m) continue;   // cannot equal any candidate
    ex.push_back((((uint64_t)i1 * n + i2) * m + j1) * m + j2 + 1);
  }
  std::sort(ex.begin(), ex.end()); ex.erase(std::unique(ex.begin(), ex.end()), ex.end());
  if ((rc = upload(h->sexist, ex.data(), ex.size() * 8, h->stream))) return rc;
  if ((rc = upload(h->bXin, X, sizeof(double) * (size_t)k * n * m, h->stream))) return rc;
  if ((rc = h->shi.ensure((size_t)N * 8)) || (rc = h->slo.ensure((size_t)N * 8))) return rc;
  HIPCHK(hipMemsetAsync(h->scnt.p, 0, 16, h->stream));
  EventPair ev;
  if (ev.create()) return fail(999, "hipEventCreate failed");
  hipEvent_t e0 = ev.e0, e1 = ev.e1;
  HIPCHK(hipEventRecord(e0, h->stream));
  long long base = 0;
  for (size_t q = 0; q < segs.size(); ++q) {
    long long t = 0;
    if ((rc = shor_scan(h, segs[q], base, &t))) return rc;
    omc_shor_launch_enum_keys(n, m, h->shor_W, h->sbits.as<uint64_t>(), segs[q].kind, segs[q].la, segs[q].lb, h->soff.as<long long>(),
                              h->shor_pairs, h->bXin.as<double>(), k, h->sexist.as<uint64_t>(), (long long)ex.size(),
                              h->shi.as<uint64_t>(), h->slo.as<uint64_t>(), h->scnt.as<unsigned long long>(), h->stream);
    HIPCHK(hipGetLastError());
    base += t;
  }
  unsigned long long excluded = 0;
  HIPCHK(hipMemcpyAsync(&excluded, h->scnt.p, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  const long long V = N - (long long)excluded;
  const long long K = std::min<long long>(n_minors, V);      // fewer candidates than n_minors: all of them, sorted (OMC.jl:2634-2635)
  if (K > 0) {
    // MSD radix select of the K largest 128-bit keys
    const long long CAP = 8192;
    uint64_t phi = 0, plo = 0; long long need = K; unsigned long long bin = 0;
    for (int level = 0; level < 16; ++level) {
      HIPCHK(hipMemsetAsync(h->shist.p, 0, 256 * 8, h->stream));
      omc_shor_launch_hist(N, h->shi.as<uint64_t>(), h->slo.as<uint64_t>(), phi, plo, level, h->shist.as<unsigned long long>(), h->stream);
      HIPCHK(hipGetLastError());
      unsigned long long hist[256];
      HIPCHK(hipMemcpyAsync(hist, h->shist.p, sizeof(hist), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      long long cum = 0; int d = 255;
      for (; d > 0; --d) { if (cum + (long long)hist[d] >= need) break; cum += (long long)hist[d]; }
      need -= cum; bin = hist[d];
      if (level < 8) phi |= (uint64_t)d << (56 - 8 * level); else plo |= (uint64_t)d << (56 - 8 * (level - 8));
      if ((long long)bin - need <= CAP) break;
    }
    const unsigned long long cap = (unsigned long long)(K + CAP + 16);
    if ((rc = h->sohi.ensure(cap * 8)) || (rc = h->solo.ensure(cap * 8))) return rc;
    HIPCHK(hipMemsetAsync((char*)h->scnt.p + 8, 0, 8, h->stream));
    omc_shor_launch_emit(N, h->shi.as<uint64_t>(), h->slo.as<uint64_t>(), phi, plo, h->sohi.as<uint64_t>(), h->solo.as<uint64_t>(),
                         h->scnt.as<unsigned long long>() + 1, cap, h->stream);
    HIPCHK(hipGetLastError());
    unsigned long long got = 0;
    HIPCHK(hipMemcpyAsync(&got, (char*)h->scnt.p + 8, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipEventRecord(e1, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (got > cap || (long long)got < K) return fail(OMC_ERR_ARGUMENT, "internal: radix select emitted an unexpected number of keys");
    std::vector<uint64_t> ohi(got), olo(got);
    HIPCHK(hipMemcpyAsync(ohi.data(), h->sohi.p, got * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(olo.data(), h->solo.p, got * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<size_t> ord(got);
    for (size_t i = 0; i < got; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return ohi[a] != ohi[b] ? ohi[a] > ohi[b] : olo[a] > olo[b]; });
    for (long long r = 0; r < K; ++r) {
      const size_t i = ord[(size_t)r];
      double sc; memcpy(&sc, &ohi[i], 8);
      scores[r] = sc;
      uint64_t key = olo[i] - 1;
      const uint64_t j2 = key % m; key /= m;
      const uint64_t j1 = key % m; key /= m;
      const uint64_t i2 = key % n; const uint64_t i1 = key / n;
      minors[4 * r] = (int64_t)i1 + 1; minors[4 * r + 1] = (int64_t)i2 + 1; minors[4 * r + 2] = (int64_t)j1 + 1; minors[4 * r + 3] = (int64_t)j2 + 1;
    }
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    h->shor_last_ms = ms; h->shor_last_candidates = N;
  }
  *n_out = (int)K;
  return 0;
}

int omc_shor_last_stats(omc_instance* h, double* ms, int64_t* candidates) {
  if (!h) return fail(OMC_ERR_ARGUMENT, "handle is NULL");
  if (ms) *ms = h->shor_last_ms;
  if (candidates) *candidates = h->shor_last_candidates;
  return 0;
}

int omc_debug_residuals(omc_instance* h, double* rp, double* rd) {
  if (!h || !rp || !rd || !h->ws.rp) return fail(OMC_ERR_ARGUMENT, "nothing staged");
  HIPCHK(hipMemcpy(rp, h->ws.rp, 8 * (size_t)h->ws.B, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(rd, h->ws.rd, 8 * (size_t)h->ws.B, hipMemcpyDeviceToHost));
  return 0;
}

int omc_tuning_set(omc_instance* h, const char* name, const char* value) {
  if (!h || !name) return fail(OMC_ERR_ARGUMENT, "omc_tuning_set: handle or name is NULL");
  bool known = false;
  for (const char* key : OMC_TUNING_KEYS) if (!strcmp(key, name)) known = true;
  if (!known) return fail(OMC_ERR_ARGUMENT, "omc_tuning_set: unknown knob");
  if (value) h->tun.kv[name] = value; else h->tun.kv.erase(name);
  return 0;
}

int omc_tuning_reload_env(omc_instance* h) {
  if (!h) return fail(OMC_ERR_ARGUMENT, "handle is NULL");
  tuning_from_env(h->tun);
  return 0;
}

int omc_debug_stamps(omc_instance* h, double* out32) {
  if (!h || !out32 || !h->ws.stamps) return fail(OMC_ERR_ARGUMENT, "no stamps");
  if (h->tun.get("OMC_SUB_DEBUG") && atoi(h->tun.get("OMC_SUB_DEBUG")) == 3 && h->ws.B >= 8) {      // diagnostics: three histograms (tools/diagnostics/gpu_nkeep_hist.py passes 96 doubles)
    HIPCHK(hipMemcpy(out32, h->ws.stamps, 96 * 8, hipMemcpyDeviceToHost));
    return 0;
  }
  HIPCHK(hipMemcpy(out32, h->ws.stamps, 32 * 8, hipMemcpyDeviceToHost));
  return 0;
}

int omc_debug_aa(omc_instance* h, int* accepted, int* rejected) {
  if (!h || !accepted || !rejected || !h->ws.accel || !h->ws.aa_nacc) return fail(OMC_ERR_ARGUMENT, "acceleration is off or nothing staged");
  HIPCHK(hipMemcpy(accepted, h->ws.aa_nacc, 4 * (size_t)h->ws.B, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(rejected, h->ws.aa_nrej, 4 * (size_t)h->ws.B, hipMemcpyDeviceToHost));
  return 0;
}

int omc_debug_diag(omc_instance* h, double* out /* 8 * slots */) {
  if (!h || !out || !h->ws.stamps) return fail(OMC_ERR_ARGUMENT, "no diagnostics");
  HIPCHK(hipMemcpy(out, h->ws.stamps + 32, 8 * 8 * (size_t)h->ws.B, hipMemcpyDeviceToHost));
  return 0;
}

int omc_last_solver_info(omc_instance* h, double* info) {
  if (!h || !info) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  info[0] = h->last_solve_seconds; info[1] = (double)h->total_sweeps; info[2] = h->ws.rho; info[3] = (double)h->ws.rmax;
  info[4] = (double)h->cone_use_lds; info[5] = (double)h->glob_use_lds; info[6] = (double)h->small_use_lds; info[7] = (double)h->ws.Rmax;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// RCCL communicator behind the C ABI
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
int rccl_load() {
  if (g_rccl.lib) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* nm : names) { lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
  if (!lib) return fail(OMC_ERR_COMM, std::string("librccl not loadable: ") + (dlerror() ? dlerror() : "?"));
  Rccl r; r.lib = lib;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(lib, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(lib, "ncclAllReduce");
  r.Broadcast = (decltype(r.Broadcast))dlsym(lib, "ncclBroadcast");
  r.AllGather = (decltype(r.AllGather))dlsym(lib, "ncclAllGather");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.Broadcast || !r.AllGather || !r.CommDestroy) return fail(OMC_ERR_COMM, "librccl lacks an expected symbol");
  g_rccl = r;
  return 0;
}
int rccl_fail(ncclResult_t e, const char* what) {
  return fail(OMC_ERR_COMM, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "RCCL error"));
}
}  // namespace

int omc_comm_unique_id(void* id_out) {
  if (!id_out) return fail(OMC_ERR_ARGUMENT, "id_out is NULL");
  int rc = rccl_load(); if (rc) return rc;
  ncclUniqueId id;
  ncclResult_t e = g_rccl.GetUniqueId(&id);
  if (e != ncclSuccess) return rccl_fail(e, "ncclGetUniqueId");
  static_assert(sizeof(ncclUniqueId) == OMC_COMM_ID_BYTES, "unique id size");
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

int omc_comm_init(omc_instance* h, int rank, int world_size, const void* id) {
  if (!h || !id) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  if (world_size < 1 || rank < 0 || rank >= world_size) return fail(OMC_ERR_ARGUMENT, "rank / world_size out of range");
  if (h->comm) return fail(OMC_ERR_ARGUMENT, "communicator already initialised");
  int rc = rccl_load(); if (rc) return rc;
  HIPCHK(hipSetDevice(h->device));
  ncclUniqueId uid; memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  ncclResult_t e = g_rccl.CommInitRank(&c, world_size, uid, rank);
  if (e != ncclSuccess) return rccl_fail(e, "ncclCommInitRank");
  h->comm = (void*)c; h->comm_rank = rank; h->comm_world = world_size;
  if ((rc = h->bcomm.ensure(64))) return rc;
  return 0;
}

int omc_allreduce_bounds(omc_instance* h, double* ub, double* lb, int* owner) {
  if (!h || !ub || !lb) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  if (!h->comm) return fail(OMC_ERR_COMM, "omc_comm_init has not been called on this handle");
  HIPCHK(hipSetDevice(h->device));
  double* d = h->bcomm.as<double>();
  const double mine = *ub;
  double v[2] = {*ub, *lb};
  HIPCHK(hipMemcpyAsync(d, v, 16, hipMemcpyHostToDevice, h->stream));
  ncclResult_t e = g_rccl.AllReduce(d, d, 2, ncclDouble, ncclMin, (ncclComm_t)h->comm, h->stream);
  if (e != ncclSuccess) return rccl_fail(e, "ncclAllReduce(min, fp64[2])");
  HIPCHK(hipMemcpyAsync(v, d, 16, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *ub = v[0]; *lb = v[1];
  if (owner) {   // second 8-byte MIN: the smallest rank that holds the minimum (ties are common: every rank starts from the same root bound)
    double r = (mine == v[0]) ? (double)h->comm_rank : (double)h->comm_world;
    HIPCHK(hipMemcpyAsync(d + 2, &r, 8, hipMemcpyHostToDevice, h->stream));
    e = g_rccl.AllReduce(d + 2, d + 2, 1, ncclDouble, ncclMin, (ncclComm_t)h->comm, h->stream);
    if (e != ncclSuccess) return rccl_fail(e, "ncclAllReduce(min, owner)");
    HIPCHK(hipMemcpyAsync(&r, d + 2, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *owner = (int)r;
  }
  return 0;
}

int omc_bcast_incumbent(omc_instance* h, int root, double* X) {
  if (!h || !X) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  if (!h->comm) return fail(OMC_ERR_COMM, "omc_comm_init has not been called on this handle");
  if (root < 0 || root >= h->comm_world) return fail(OMC_ERR_ARGUMENT, "root out of range");
  HIPCHK(hipSetDevice(h->device));
  const size_t cnt = (size_t)h->n * h->m;
  int rc = h->bXin.ensure(8 * cnt + 64); if (rc) return rc;
  double* d = h->bXin.as<double>();
  if (h->comm_rank == root) HIPCHK(hipMemcpyAsync(d, X, 8 * cnt, hipMemcpyHostToDevice, h->stream));
  ncclResult_t e = g_rccl.Broadcast(d, d, cnt, ncclDouble, root, (ncclComm_t)h->comm, h->stream);
  if (e != ncclSuccess) return rccl_fail(e, "ncclBroadcast(X)");
  if (h->comm_rank != root) HIPCHK(hipMemcpyAsync(X, d, 8 * cnt, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// The per-node records every rank needs to grow the same tree (the serial loop OMC.jl:700-719 sees every relaxed node: status, objective,
// bound, eigenvalues, breakpoint vector, U -- 8 (6 + n + n k) bytes per node, never X or Y): all-gather of `cnt` rows of `width` doubles per rank.
// Ranks may hold different counts: the counts are gathered first (8 bytes per rank), the rows are padded to the largest count, and
// `out` receives the rows of rank 0, then rank 1, ... (sum of counts x width doubles; capacity = world x max_cnt_capacity rows).
int omc_allgather_records(omc_instance* h, const double* rows, int cnt, int width, double* out, int out_capacity_rows, int* counts /* world */) {
  if (!h || !out || !counts || (cnt > 0 && !rows)) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  if (cnt < 0 || width <= 0) return fail(OMC_ERR_ARGUMENT, "cnt / width out of range");
  if (!h->comm) return fail(OMC_ERR_COMM, "omc_comm_init has not been called on this handle");
  HIPCHK(hipSetDevice(h->device));
  const int W = h->comm_world;
  int rc = h->bcomm.ensure(64 + 8 * (size_t)W); if (rc) return rc;
  double* d = h->bcomm.as<double>();
  const double mine = (double)cnt;
  HIPCHK(hipMemcpyAsync(d + 4, &mine, 8, hipMemcpyHostToDevice, h->stream));
  ncclResult_t e = g_rccl.AllGather(d + 4, d + 8, 1, ncclDouble, (ncclComm_t)h->comm, h->stream);
  if (e != ncclSuccess) return rccl_fail(e, "ncclAllGather(counts)");
  std::vector<double> hc(W);
  HIPCHK(hipMemcpyAsync(hc.data(), d + 8, 8 * (size_t)W, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  int cmax = 0; long tot = 0;
  for (int r = 0; r < W; ++r) { counts[r] = (int)hc[r]; cmax = std::max(cmax, counts[r]); tot += counts[r]; }
  if (tot > out_capacity_rows) return fail(OMC_ERR_ARGUMENT, "omc_allgather_records: out is too small for the gathered rows");
  if (cmax == 0) return 0;
  const size_t blk = (size_t)cmax * width;
  if ((rc = h->bXin.ensure(8 * blk * (size_t)(W + 1) + 64))) return rc;
  double* send = h->bXin.as<double>(); double* recv = send + blk;
  HIPCHK(hipMemsetAsync(send, 0, 8 * blk, h->stream));
  if (cnt) HIPCHK(hipMemcpyAsync(send, rows, 8 * (size_t)cnt * width, hipMemcpyHostToDevice, h->stream));
  e = g_rccl.AllGather(send, recv, blk, ncclDouble, (ncclComm_t)h->comm, h->stream);
  if (e != ncclSuccess) return rccl_fail(e, "ncclAllGather(records)");
  size_t o = 0;
  for (int r = 0; r < W; ++r) {
    if (counts[r]) HIPCHK(hipMemcpyAsync(out + o, recv + (size_t)r * blk, 8 * (size_t)counts[r] * width, hipMemcpyDeviceToHost, h->stream));
    o += (size_t)counts[r] * width;
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int omc_comm_destroy(omc_instance* h) {
  if (!h) return fail(OMC_ERR_ARGUMENT, "handle is NULL");
  if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)h->comm);
  h->comm = nullptr; h->comm_rank = 0; h->comm_world = 1;
  return 0;
}

int omc_last_subspace_stats(omc_instance* h, int64_t* out) {
  if (!h || !out) return fail(OMC_ERR_ARGUMENT, "NULL argument");
  for (int q = 0; q < 8; ++q) out[q] = h->sub_tot[q];
  return 0;
}

int omc_last_kernel_stats(omc_instance* h, int64_t* launches, double* ms, int64_t* units) {
  if (!h) return fail(OMC_ERR_ARGUMENT, "handle is NULL");
  for (int c = 0; c < OMC_KERNEL_NCLASS; ++c) {
    if (launches) launches[c] = h->launches[c];
    if (ms) ms[c] = h->ms[c];
    if (units) units[c] = h->units[c];
  }
  return 0;
}

}  // extern "C"
