// Host-visible launch interface of the HIP kernels (implemented in prep.hip / strip.hip).
// dtype: 0 = f64, 1 = f32 (SVGP_F64 / SVGP_F32).  All pointers are device pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svgp {

struct KernelParams {
  int family;        // SVGP_KERNEL_*
  int d;
  double variance;
  const void* invl;  // [d] inverse lengthscales, compute dtype
};

constexpr int kLikExternal = -1;
struct LikParams {
  int lik;            // SVGP_LIK_*, or kLikExternal: a likelihood the host evaluated (svgp_elbo_grad_ext) - gh_x / gh_w then hold
                      // the chunk's point gradients dE_i/dmu_i / dE_i/dv_i (unscaled), indexed by the point's position in the chunk
  int gh_n;           // 0 = closed form
  double sigma2;      // likelihood parameter: Gaussian sigma^2, Gamma shape alpha (1 otherwise)
  double digamma_alpha;  // digamma(alpha) for the Gamma likelihood's parameter gradient
  const double* gh_x; // [gh_n] nodes (device)
  const double* gh_w; // [gh_n] weights / sqrt(pi) (device)
  int clamp_neg_var;
  double mean_const;
};

// ---- prep.hip --------------------------------------------------------------------------------
// zs[k][i] = z_k,i * invl[k] for i < M, 0 for M <= i < Mp.  z given per `layout` (SVGP_COLVECS...).
void launch_scale_inputs(int dtype, hipStream_t s, const void* z, int layout, int d, int64_t M, int64_t Mp,
                         const void* invl, void* zs);
// host ColVecs (d x n, point-contiguous) -> feature-major [d][ldx]
void launch_transpose_colvecs(int dtype, hipStream_t s, const void* x_dn, int d, int64_t n, int64_t ldx, void* x_fm);
// Kuu (Mp x Mp col-major): k(z_i, z_j) + jitter*[i==j] on the M x M block, identity on the padding.
void launch_kuu(int dtype, hipStream_t s, const KernelParams& kp, const void* zs, int64_t M, int64_t Mp,
                double jitter, void* Kuu);
// Blocked Cholesky in place (lower).  T receives the inverted 128x128 diagonal blocks AND, since round 3, its panels
// T[I, <I] = -inv(L_II) * L[I, <I] (computed inside the factorisation's own launches); info = 0 or the 1-based order of the
// first non-positive pivot; sync: Mp / 128 zeroed counters (device) for the in-kernel hand-over of the next diagonal block.
// row_events (nullable; Mp / 128 <= potrf_max_row_events() events): event p is recorded on `s` once block row p of T is final
// row hook: called on the host right after row_events[p] is recorded (p = block row of T that has just become final), so that the
// caller can enqueue the work that waits for that event BEHIND the record and still AHEAD of the device (api.hip: SegRun)
struct RowHook { void (*fn)(void* user, int row) = nullptr; void* user = nullptr; };
void launch_potrf(int dtype, hipStream_t s, void* A, void* T, int64_t Mp, int* info, unsigned* sync, int num_cus, hipEvent_t* row_events = nullptr,
                  const RowHook* hook = nullptr);
int potrf_max_row_events();
// U = Lq' (upper triangular, Mp x Mp col-major, zero padding); also mp[i] = m[i] padded with zeros.
void launch_pack_q(int dtype, hipStream_t s, const void* Lq, const void* m, int64_t M, int64_t Mp, void* U, void* mp);
// scal[0] = sum(Lq.^2 lower), scal[1] = m'm, scal[2] = sum log diag Lq, scal[3] = sum log diag Lk (first M)
void launch_kl_terms(int dtype, hipStream_t s, const void* Lq, const void* m, const void* Lk, int64_t M, int64_t Mp,
                     double* scal);
void launch_pack_q_ld(int dtype, hipStream_t s, const void* Lq, int64_t ldq, const void* m, int64_t M, int64_t Mp, void* U,
                       void* mp);
void launch_kl_terms_ld(int dtype, hipStream_t s, const void* Lq, int64_t ldq, const void* m, const void* Lk, int64_t M,
                        int64_t Mp, double* scal);
// x := L \ x (trans = 0) or L' \ x (trans = 1) for one vector of length Mp, using the inverted diagonal blocks in T
void launch_trsv2(int dtype, hipStream_t s, const void* L, const void* Tm, int64_t Mp, int trans, void* x);
// X := Lk \ X in place for a lower-triangular Mp x Mp column-major X (Centered B = Lk \ Lq, SVA:133)
void launch_trsm_mat(int dtype, hipStream_t s, const void* Tm, int64_t Mp, void* X);
// out (Mp x Mp, ld Mp) = lower triangle of Lq (M x M, ld M), zero elsewhere
void launch_pad_lower(int dtype, hipStream_t s, const void* Lq, int64_t M, int64_t Mp, void* out);
// out[i] = m[i] + shift for i < M, 0 for M <= i < Mp
void launch_shift_vec(int dtype, hipStream_t s, const void* m, double shift, int64_t M, int64_t Mp, void* out);
// copy the lower triangle of the leading M x M block of A (ld Mp) into out (ld M), zero the upper part
void launch_extract_lower(int dtype, hipStream_t s, const void* A, int64_t Mp, int64_t M, void* out);

// ---- strip.hip -------------------------------------------------------------------------------
enum : int { kSegPregen = 1, kSegPhase2 = 2, kSegLoad = 4, kSegStore = 8 };   // StripArgs::seg_flags
struct StripArgs {
  const void* T;     // Mp x Mp col-major: block rows of inv(L_II) * [-L_I,<I | I]
  const void* U;     // Mp x Mp col-major: B' (upper triangular)
  const void* zs;    // [d][Mp] scaled inducing inputs
  const void* mp;    // [Mp] padded mean of q
  const void* x;     // [d][ldx] feature-major inputs
  unsigned* counter; // zeroed before the launch: dynamic strip queue (strips beyond the first gridDim.x)
  void* work;        // grid * Mp * NT elements: per-workgroup A strip
  double* mom_mu;    // [len] posterior mean of every point of the batch      (SVA:250)
  double* mom_var;   // [len] posterior variance, before the 1e-18 of f_post(x) (SVA:251)
  void* A_out;       // nullable: A  = Lk \ Kuf as a k-major [Mp][lda] matrix (predict-cov path)
  void* C_out;       // nullable: B'A
  void* At_out;      // nullable: the same A, point-major [n][Mp] (column-major Julia A): gradient path
  void* Ct_out;      // nullable: B'A point-major
  int64_t lda;
  int64_t ldx, off, len;
  int64_t Mp, M;
  KernelParams kp;
  double mean_const;
  // ---- value-and-gradient strips (launch_strip_grad): phase 3 of the same kernel, P = Kuf_bar for the strip's points ----
  const void* R;        // Mp x Mp col-major: Lk^-T (Lq Lq' - I)
  const void* alpha;    // [Mp] Lk^-T m
  void* Pt_out;         // point-major [n][Mp]: the UNSCALED product R A (P = alpha g_mu' + 2 (R A) diag(g_v) is formed by kgrad)
  // (the likelihood gradients g_mu, g_v and the per-block sums come from launch_point_grads, which reads mom_mu / mom_var)
  // ---- segmented forward strips (launch_strip_seg): one launch covers the phase-1 panels [seg_lo, seg_hi) of every strip ----
  int seg_flags;              // kSegPregen 1: generate the Kuf block first; kSegPhase2 2: phase 2 + moments after the panels;
                              // kSegLoad 4 / kSegStore 8: restore / save the threads' fp64 column sums in seg_state
  int seg_lo, seg_hi;
  double* seg_state;          // [nstrips][256][2 NJ]; `work` then holds ONE scratch strip PER STRIP (nstrips x Mp x NT)
  // Split closing launch (segmented strips of a SMALL batch: fewer strips than workgroup slots).  The output panels of phase 3 (value and
  // gradient: dense) and of phase 2 (forward: C_J = sum_{I >= J} U[J, I] A_I) are independent, so the closing launch runs seg_split
  // workgroups per strip - workgroup (strip, part) = blockIdx (part * nstrips + strip) takes the panels [part nP / S, (part + 1) nP / S)
  // of phase 3, or J = part, part + S, .. of phase 2 - and the parts' column sums meet in seg_part ([S][nstrips][3][NT] doubles: sum A^2,
  // mean, variance term); the last part to arrive (seg_cnt[strip], zeroed by the host) adds them in part order and writes the moments.
  // Run-to-run bitwise reproducible; against the unsplit kernel the variance differs by summation order.
  int seg_split;
  double* seg_part;
  unsigned* seg_cnt;
};
int strip_nt(int dtype, int64_t Mp, int64_t len);                 // column-strip width chosen for a problem
size_t strip_work_bytes(int dtype, int64_t Mp, int nt, int grid);  // workspace bytes
int strip_grid(int dtype, int nt, int64_t nstrips, int num_cus);
struct StripPlan {       // regular-width launch over the first `points` points, then an optional half-width tail launch
  int nt, grid;
  int64_t nstrips, points;
  int nt_tail, grid_tail;
  int64_t nstrips_tail;
  bool concurrent_tail;   // the tail launch runs on a second stream BESIDE the main one (last partial round of a large batch)
};
StripPlan strip_plan(int dtype, int64_t Mp, int64_t len, int num_cus);
StripPlan strip_plan_single(int dtype, int64_t Mp, int64_t len, int num_cus);   // never a concurrent tail (paths that write A / C)
void launch_strip(int dtype, hipStream_t s, const StripArgs& a, int nt, int grid, int64_t nstrips);
void launch_strip_seg(int dtype, hipStream_t s, const StripArgs& a, int nt, int grid, int64_t nstrips, bool grad = false);
size_t strip_seg_state_doubles(int dtype, int nt);
// the value-and-gradient form: phase 1 as launch_strip, then phase 3 (a dense Mp x Mp GEMM R A on the strip's A, still in its
// scratch strip, whose epilogue also gives the variance) and the per-point likelihood gradients; writes At_out, Pt_out (R A,
// unscaled), gmu_out, gv_out, part5.  `work` must hold TWO scratch strips per workgroup (2 x strip_work_bytes).
// post = true (round 4): the strips leave their moments in mom_mu / mom_var (like launch_strip) and write no gmu / gv / part5 -
// launch_point_grads, next on the stream, produces those from the moments.  post = false: the round-3 in-kernel forms.
void launch_strip_grad(int dtype, hipStream_t s, const StripArgs& a, int nt, int grid, int64_t nstrips);
// marginals, expected log-likelihood and d E / d (mu, v) (x scale) of the points [off, off + len) of y from their moments: gmu_out /
// gv_out (compute dtype, [len]) and part5[point_grad_blocks(len)][5] = per-block {E, sum g_mu, sum g_v, dE/dsigma2, n_neg}
int point_grad_blocks(int64_t len);
void launch_point_grads(int dtype, hipStream_t s, const LikParams& lp, const double* mom_mu, const double* mom_var, const void* y,
                        int64_t off, int64_t len, double scale, const double* n_global_dev, double num_data, void* gmu_out,
                        void* gv_out, double* part5, unsigned* strip_queue = nullptr, int64_t pad_to = 0);
// marginals + expected log-likelihood of every point (SVA:354-355): per-block sums into partial/negcnt
int expect_blocks(int64_t len);
void launch_expect(int dtype, hipStream_t s, const LikParams& lp, const double* mom_mu, const double* mom_var,
                   const void* y, int64_t off, int64_t len, double* partial, unsigned* negcnt, void* mu_out,
                   void* var_out);
// out[0..8) = {sum(partial[0..n)), n_points, sum(negcnt), *chol_info != 0, 0, 0, 0, 0} in fixed order (deterministic):
// the vector a data-parallel evaluation all-reduces
// out[8..13) (not all-reduced) = this rank's prep scalars prep_scal[0..4) and *chol_info: one read-back per evaluation
void launch_final_reduce(hipStream_t s, const double* partial, const unsigned* negcnt, int64_t n, const int* chol_info,
                         double n_points, double* out, const double* prep_scal);
// standalone Kuf (M x len col-major, ld = M)
void launch_kuf(int dtype, hipStream_t s, const KernelParams& kp, const void* zs, int64_t M, int64_t Mp,
                const void* x, int64_t ldx, int64_t off, int64_t len, void* Kuf);
// out(n x n col-major) = kxx - A'A + C'C  from k-major A, C ([Mp][lda]); prior block computed from x.
void launch_cov_assemble(int dtype, hipStream_t s, const KernelParams& kp, const void* xa, int64_t ldxa, int64_t na,
                         const void* xb, int64_t ldxb, int64_t nb, const void* Aa, const void* Ca, int64_t lda,
                         const void* Ab, const void* Cb, int64_t ldb, int64_t Mp, void* out);

// ---- grad.hip (reverse-mode gradient of the NonCentered ELBO) -------------------------------------------
constexpr double kDefaultSigma2 = 1e-18;  // AbstractGPs.default_σ² added by f_post(x) (SVA:354)
int grad_dreg(int d);
int grad_rowblocks(int dtype, int d, int64_t Mp);
void launch_set_f64(hipStream_t s, double* dst, double value);
void launch_setvec_f64(hipStream_t s, double* dst, const double* vals, int n);   // dst[0 .. n) = vals, n <= 64, as kernel arguments (never blocks the host)
void launch_set2_f64(hipStream_t s, double* dst, double a, double b);   // dst[0] = a, dst[1] = b (values travel as kernel arguments: no host buffer to outlive)
// sums[5] = n_points, sums[6] = (*chol_info != 0), sums[7] = 0: the status slots of the all-reduced gradient scalars
// ... and, behind them, a copy of the prep scalars (4) and chol_info (as a double): prep_out[0..5)
void launch_grad_status(hipStream_t s, double* sums, const int* chol_info, double n_points, const double* prep_scal, double* prep_out);
// flags of the M x M x M products (gemm_pm_kernel): all tiles instead of the lower ones; triangular operands (the contraction
// starts / ends at the diagonal tile of the output row (X) or column (Y))
enum : int { kMmFull = 1, kMmXLow = 2, kMmYLow = 4, kMmXUp = 8, kMmYUp = 16 };
void launch_gemm_pm(int dtype, hipStream_t s, const void* Xt, const void* Yt, const void* w, double wscale, int64_t Mp,
                    int64_t n, int64_t slice_len, int nslices, void* out, int overwrite = 0, int flags = 0);
// the data-sized SYRK with UNIFORM weights (Gaussian likelihood): out (+)= wscale * w * scale * At' At over the first n points (slices of
// slice_len); scale = num_data / *n_global_dev when n_global_dev is given.  Rows [n, ceil16(n)) of At must be zero.
void launch_syrk_uniform(int dtype, hipStream_t s, const void* At, double w, double scale, const double* n_global_dev, double num_data,
                         double wscale, int64_t Mp, int64_t n, int64_t slice_len, int nslices, void* out, int overwrite);
// out (lower 128-tiles, or all with full) (+)= the sum of `ns` slice partials written by launch_gemm_pm(..., overwrite = 1)
void launch_sum_slices_lower(int dtype, hipStream_t s, const void* part, int ns, int64_t Mp, void* out, int full = 0, int overwrite = 0);
// Linv = Lk^-1 by recursive doubling from the inverted diagonal blocks in T: LinvRM row-major, LinvCM column-major (only
// the lower block triangle is written / ever read); Ytmp: Mp x Mp scratch
void launch_linv(int dtype, hipStream_t s, const void* L, const void* Tm, int64_t Mp, void* LinvRM, void* LinvCM, void* Ytmp);
// out = Lk^-T v = LinvRM' v; part: (Mp / 128) x Mp doubles of scratch
// vec_f64: v and out are fp64 whatever the matrix dtype (the kernel-gradient reductions keep their sums in fp64)
void launch_linv_t_gemv(int dtype, hipStream_t s, const void* LinvRM, const void* v, int64_t Mp, void* out, double* part, int notrans = 0,
                        int vec_f64 = 0);
// data part: Pt = the strips' unscaled R A, P_ij = alpha_i gmu_j + 2 gv_j Pt_ji formed inside (alpha != nullptr); Kuu part: Pt is
// the matrix itself (alpha = gmu = gv = nullptr); kmb: also the row sums (Kuf g_mu)_i into slot 1 (the caller applies Lk^-1: A g_mu)
void launch_kgrad(int dtype, hipStream_t s, const KernelParams& kp, const void* zs, int64_t Mp, int64_t M, const void* x, int64_t ldx,
                  int64_t xoff, int prescaled, int64_t n, int64_t nvalid, const void* Pt, const void* gmu,
                  const void* gv, const void* alpha, int64_t slice_len, int nslices, double* rowpart, double* scalpart, int kmb = 0);
// fused gradient path: sums of the per-strip partials, W = A diag(2 g_v) A' from its split-K lower tiles, (A g_mu), and the
// assembly of Lq_bar / Lk_bar from G1 = 2 W Lq, G2 = 2 R W and the rank-one term alpha (A g_mu)'
void launch_sum5(hipStream_t s, const double* partial, int nblocks, double* sums);
void launch_sym_from_lower(int dtype, hipStream_t s, const void* G, int nslices, int64_t Mp, double eye, void* out);
void launch_avec(hipStream_t s, const double* rp_uf, int ns, int64_t stride, int64_t Mp, double* avec);
void launch_finish_mm2(int dtype, hipStream_t s, const void* G1, const void* G2, const void* alpha, const double* avec, int64_t Mp,
                       int64_t M, const void* Lq, int64_t ldq, double klw, void* Lq_bar, void* BbarRM, void* LkbarRM);
void launch_lower_to_rowmajor(int dtype, hipStream_t s, const void* L, int64_t Mp, void* out);
void launch_symmetrize(int dtype, hipStream_t s, const void* St, int64_t Mp, void* H);
void launch_phi(int dtype, hipStream_t s, void* X, int64_t Mp);
void launch_mbar(int dtype, hipStream_t s, const double* avec, const void* mt, double klw, int64_t M, int64_t Mp, void* vec);
void launch_lbar_adjust(int dtype, hipStream_t s, void* LkbarRM, const void* RBt, const void* rbar, const void* mt, int64_t Mp);
void launch_cm_tril_to_user(int dtype, hipStream_t s, const void* R, int64_t Mp, int64_t M, void* out);   // R column-major
void launch_add_f64(hipStream_t s, double* p, double v);   // *p += v
// {head entries | M x M column-major lower-triangular block} <-> {head entries | its M (M + 1) / 2 packed entries}; dir 0 packs
void launch_pack_tril(int dtype, hipStream_t s, void* gblk, void* packed, int64_t head, int64_t M, int dir);
void launch_finish_kgrad(int dtype, hipStream_t s, int d, int64_t M, int64_t Mp, const void* zs, const double* invl,
                         const double* rp_uf, int ns_uf, const double* rp_uu, int ns_uu, const double* sp_uf, int nsp_uf,
                         const double* sp_uu, int nsp_uu, const void* m, double klw, int layout_z, double variance,
                         void* z_bar, void* m_bar, double* scal_out, double* red, const double* avec);

}  // namespace svgp
