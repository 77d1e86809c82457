// ctx.hpp — internal definitions shared by the host-side translation units of libsvgp_mi355x (api.hip, comm.hip):
// the opaque handles of include/svgp_mi355x.h and the error macros.  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/svgp_mi355x.h"
#include "knobs.hpp"

// ------------------------------------------------------------------------------------------------
struct svgp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int num_cus = 256;
  svgp::Knobs kn;   // environment settings, read once at context creation (knobs.hpp)
  std::string err;
  svgp_timing timing{};
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // start, prep done, strip done, all done
  hipEvent_t ev_chol[2] = {nullptr, nullptr};               // around the blocked Cholesky inside the prep
  // growable scratch
  void* work = nullptr;       size_t work_bytes = 0;
  double* partial = nullptr;  unsigned* negcnt = nullptr;  // [1024] per-block sums of the expectation kernel
  double* mom = nullptr;      size_t mom_cap = 0;           // [2][mom_cap] per-point mean / variance
  double* d_res = nullptr;    // [8] device results
  unsigned* counter = nullptr; // strip queue head of the running strip launch
  // concurrent narrow-strip launch for the last partial round of a batch (enqueue_strips): its own stream, queue and scratch
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  unsigned* counter2 = nullptr;
  void* work2 = nullptr;      size_t work2_bytes = 0;
  // strips beside the factorisation (api.hip: SegRun): one event per block row of T, the segmented strips' saved sums
  hipEvent_t ev_row[16] = {};
  bool ev_row_ready = false, overlapped = false;
  bool timing_on = true;   // SVGP_TIMING=0 at context creation: no timing events on the stream (each record costs the stream ~5 us); svgp_last_timing then reports zeros
  hipEvent_t ev_S = nullptr;                  // the chain-independent part of the gradient's prep (S = B B' - I, cleared accumulators), second stream
  hipEvent_t ev_R = nullptr;                  // the gradient's M-sized prep (Linv, alpha, R) is final: phase 3 of the segmented strips
  hipEvent_t ev_ov[2] = {nullptr, nullptr};   // timed: fork point, first strip launch done (svgp_timing.ms_overlap)
  double* seg_state = nullptr; size_t seg_state_doubles = 0;
  void* work_seg = nullptr;    size_t work_seg_bytes = 0;   // per-strip scratch of the segmented strips
  void* hstage = nullptr;      size_t hstage_bytes = 0;     // pinned host staging of the gradient read-back (api.hip: grad_finish)
  hipEvent_t ev_piece[8] = {};                               // one behind each piece of that read-back
  // chunk pipeline of a value-and-gradient evaluation (api.hip: grad_enqueue_impl): up to two streams that carry the chunks' strips
  // beside the main stream's SYRK / kernel-gradient launches of the chunks before; the second one has its own scratch, queue head, moments
  hipStream_t pst[2] = {nullptr, nullptr};
  hipEvent_t ev_pipe_prep = nullptr;
  void* pwork = nullptr;       size_t pwork_bytes = 0;
  unsigned* pcounter = nullptr;
  double* pmom = nullptr;      size_t pmom_cap = 0;
  bool pipelined = false;      // the last value-and-gradient call enqueued work on pst[]
  void* kuf_buf = nullptr;    size_t kuf_bytes = 0;
  double* ext_g = nullptr;    size_t ext_cap = 0;           // [2][ext_cap] point gradients of a host-evaluated likelihood
  struct GradWs* gws = nullptr;  // gradient workspace, cached by problem shape
  // data-parallel communicator (comm.hip): one RCCL rank per context; world == 1 without one
  void* comm = nullptr;        // ncclComm_t
  int world = 1, rank = 0;
  bool comm_owned_by_group = false;
  double* h_open = nullptr;    // pinned host word: the reduced failure flag of svgp_elbo_grad's opening all-reduce (api.hip: grad_handshake)
  hipEvent_t ev_open = nullptr;   // ... recorded behind its copy
  double* d_coll = nullptr;    // [8] the all-reduced vector {sum E, n_points, n_neg_var, chol flag, failure flag, ...}
};

// device buffers of svgp_elbo_grad, sized by (dtype, Mp, d, nc)
struct GradWs {
  int dtype = -1, d = 0, nslices = 1, ns_uf = 1, ns_uu = 1, rb = 1;
  int64_t Mp = 0, nc = 0;
  std::vector<void*> all;
  void *At = nullptr, *Pt = nullptr, *gmu = nullptr, *gv = nullptr;   // per chunk: A and P = Kuf_bar point-major [nc][Mp], g_mu, g_v
  void *Lqp = nullptr, *G1 = nullptr, *G2 = nullptr, *LkRM = nullptr, *LbarRM = nullptr, *Phi = nullptr,
       *tmp = nullptr, *H = nullptr, *BbarRM = nullptr, *rbar = nullptr;
  void *LinvRM = nullptr, *LinvCM = nullptr;   // Lk^-1 in both storage orders (launch_linv): every Lk^-T . of the tail is a GEMM with it
  void* Gmm = nullptr;   // split-K scratch of the M x M products that run BESIDE an early SYRK (api.hip: syrk_early), allocated on first use
  void *W2 = nullptr, *Rcm = nullptr, *G1p = nullptr, *alpha = nullptr;   // W = A diag(2 g_v) A', R = Lk^-T (Lq Lq' - I), 2 W Lq, Lk^-T m
  // the user-layout gradient blocks {z_bar (M d) | m_bar (M) | Lq_bar (M^2)}: ONE allocation, contiguous for the model's M, so
  // that the data-parallel sum is one ncclAllReduce and the read-back one copy; zbar / mbar / Lqbar point into it (set per call)
  void *gblk = nullptr, *zbar = nullptr, *mbar = nullptr, *Lqbar = nullptr;
  void* cblk = nullptr;   // the same block with Lq_bar packed to its lower triangle: what the collective all-reduces (M d + M + M (M + 1) / 2)
  // accumulators zeroed by ONE memset per evaluation: [rp_uf | sp_uf | rp_uu | sp_uu | sums (8) | scal_out (1 + dreg) | prep (5)]
  void* zero_blk = nullptr;
  size_t zero_b = 0;
  double *rp_uf = nullptr, *sp_uf = nullptr, *rp_uu = nullptr, *sp_uu = nullptr, *partial5 = nullptr, *sums = nullptr,
         *invl_d = nullptr, *scal_out = nullptr, *avec = nullptr, *kred = nullptr, *gemv_part = nullptr;
  int64_t part5_strips = 0;
  size_t rp_uf_b = 0, sp_uf_b = 0, rp_uu_b = 0, sp_uu_b = 0, g_b = 0;
  // chunk pipeline: buffer sets ("lanes") of the per-chunk arrays; lane 0 is {At, Pt, gmu, gv, partial5} above, the others are
  // allocated the first time a call has more than one chunk
  struct Lane {
    void *At = nullptr, *Pt = nullptr, *gmu = nullptr, *gv = nullptr;
    double* partial5 = nullptr;
    hipEvent_t ev_strips = nullptr, ev_done = nullptr;   // the chunk's A / P / g are complete; its SYRK and reductions have read them
  };
  std::vector<Lane> lanes;
  void release() {
    for (void* p : all)
      if (p) (void)hipFree(p);
    all.clear();
    for (Lane& l : lanes) {
      if (l.ev_strips) (void)hipEventDestroy(l.ev_strips);
      if (l.ev_done) (void)hipEventDestroy(l.ev_done);
    }
    lanes.clear();
  }
};

struct svgp_data {
  int dtype = 0, d = 0;
  int64_t n = 0, ldx = 0;
  void* x = nullptr;  // feature-major [d][ldx]
  void* y = nullptr;
  bool own = true;
};

struct svgp_model {
  svgp_model_desc desc{};
  std::vector<double> invl_host;
  int64_t M = 0, Mp = 0;
  int dtype = 0, d = 0;
  size_t es = 8;
  void *z_raw = nullptr, *m_raw = nullptr, *Lq_raw = nullptr;  // user layout
  void *invl = nullptr, *zs = nullptr, *L = nullptr, *T = nullptr, *U = nullptr, *mp = nullptr, *B = nullptr;
  double* scal = nullptr;  // [8 + Mp]
  int* info = nullptr;
  double *gh_x = nullptr, *gh_w = nullptr;
  int gh_n = 0;
  bool prepared = false;
  // host copies of the last prep's scalars
  double kl = 0, logdet_kuu = 0;
  int chol_info = 0;
};

// one process driving several GPUs: member contexts share one RCCL communicator (ncclCommInitAll), rank i = member i
struct svgp_group {
  std::vector<svgp_ctx*> ctxs;
  std::string err;
};

#define HIPC(ctx, call)                                                                            \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_) + svgp::noted();              \
      return (e_ == hipErrorOutOfMemory) ? SVGP_OOM : SVGP_HIP_ERROR;                              \
    }                                                                                              \
  } while (0)

namespace svgp {
std::string take_note_text();   // prep.hip: the pending host-side diagnosis (device_common.hpp: leave_note), emptied
inline std::string noted() {
  const std::string n = take_note_text();
  return n.empty() ? n : " [" + n + "]";
}
}  // namespace svgp

// SVGP_DEBUG_SYNC=1: synchronise and check after every kernel launch, naming the offender.
inline bool debug_sync() {
  static const bool on = [] { const char* e = getenv("SVGP_DEBUG_SYNC"); return e && e[0] == '1'; }();
  return on;
}
#define KCHECK(ctx, name)                                                                          \
  do {                                                                                             \
    hipError_t e_ = hipGetLastError();                                                             \
    if (e_ == hipSuccess && debug_sync()) e_ = hipStreamSynchronize((ctx)->stream);                \
    if (e_ != hipSuccess) {                                                                        \
      (ctx)->err = std::string("kernel ") + name + ": " + hipGetErrorString(e_) + svgp::noted();   \
      return SVGP_HIP_ERROR;                                                                       \
    }                                                                                              \
  } while (0)

namespace svgp {
// device memory released on every path out of a function (the HIPC macro returns early)
struct DevBuf {
  void* p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
};
// a svgp_data under construction: freed unless release()d to the caller
struct DataGuard {
  svgp_data* D = nullptr;
  ~DataGuard() {
    if (!D) return;
    if (D->own) {
      if (D->x) (void)hipFree(D->x);
      if (D->y) (void)hipFree(D->y);
    }
    delete D;
  }
  svgp_data* release() { svgp_data* d = D; D = nullptr; return d; }
};

// ---- comm.hip: RCCL, loaded lazily with dlopen (the library has no link-time dependency on it) ----
// in-place sum all-reduce of `count` elements (f64: dtype 0, f32: dtype 1) on the context's stream; no host sync
int comm_allreduce(svgp_ctx* ctx, void* buf, size_t count, int dtype);
int comm_group_start(svgp_ctx* ctx);
int comm_group_end(svgp_ctx* ctx);
void comm_abort(svgp_ctx* ctx);   // after a local failure that left peers inside a collective: abort instead of hanging

inline int fail(svgp_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}
}  // namespace svgp
