/* temx.h -- C ABI of libtemx.so, the MI355X (gfx950) TEM-diagnostics engine.
 *
 * The reference (jhollowed/PyTEMDiags) is pure Python and has no FFI: its boundary for the
 * hot path is the public Python API.  These entry points are what a ctypes binding inside
 * the reference would call in place of its numpy/scipy call sites; each one cites the
 * reference code it replaces (paths relative to PyTEMDiags/).  INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the plan's device unless the name ends in _host;
 *   - fields are row-major [ncol][D] with D = nlev*nt (time fastest), exactly the
 *     (N, DD) reshape of sph_zonal_mean.py:243-246 after _config_dims' transpose
 *     (tem_diagnostics.py:343-357);
 *   - dtype: TEMX_F64 or TEMX_F32 input fields; all arithmetic and all outputs are fp64
 *     (the reference's matrices are fp64, so its matmul accumulates in fp64,
 *     sph_zonal_mean.py:278-282; the Python front end applies the final astype);
 *   - stream: a hipStream_t passed as void* (NULL = default stream); all compute entry
 *     points are asynchronous and stream ordered, they allocate nothing once the plan's
 *     workspace has been sized (first call for a given D);
 *   - return value: 0 (TEMX_OK) or a negative TEMX_E* code; temx_last_error() gives the
 *     thread-local message;
 *   - a plan is bound to one device and is not thread safe; different plans are independent.
 */
#ifndef TEMX_H
#define TEMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct temx_plan temx_plan;

enum { TEMX_F64 = 0, TEMX_F32 = 1 };

enum {
  TEMX_OK = 0,
  TEMX_EINVAL = -1,   /* bad argument */
  TEMX_EHIP = -2,     /* HIP runtime error (no device, launch failure, ...) */
  TEMX_ENOMEM = -3,   /* device (or host) allocation failed */
  TEMX_ERANK = -4,    /* Gram matrix has no positive eigenvalue (a rank-deficient Y0 is handled by a
                         pseudo-inverse, like the reference's lstsq) */
  TEMX_ESTATE = -5,   /* call order violated (plan not finalised, TEM levels not set, ...) */
  TEMX_EUNSUPPORTED = -6,
  TEMX_EINTERNAL = -7 /* a C++ exception inside the library (reported, never propagated across this ABI);
                         host allocation failures come back as TEMX_ENOMEM */
};

enum {
  TEMX_DEFER_FINALIZE = 1,
  TEMX_NO_SYMMETRY = 2,  /* generic sweeps only: neither latitude classes nor mirror pairing */
  TEMX_NO_CLASSES = 4,   /* do not use the latitude-class sweeps (mirror pairing is still tried) */
  TEMX_NO_QR = 8,        /* keep the Y0 basis and the explicit inverse of the normal equations (A/B runs; see
                            temx_plan_finalize).  TEMX_NO_QR=0/1 in the environment overrides */
  TEMX_LAT_TOL_F32 = 16  /* the fields will be fp32 (results compared to ~1e-5): latitudes are matched to 1e-8 degrees
                            instead of 1e-11 when latitude classes / mirror pairs are looked for, so that grids
                            whose latitude coordinate carries more noise keep the class sweeps (error ~ L x tol) */
};

/* Path selection (temx_plan_configure; takes effect at the next temx_plan_set_tem, which must follow).  The
 * defaults choose by grid and problem size; the TEMX_* environment variables named here override both. */
enum {
  TEMX_OPT_FORM = 1,               /* form of the latitude-class sweeps, TEMX_FORM_*
                                      (env: TEMX_TWO_PASS=1, TEMX_ONE_PASS=1, TEMX_SINGLE_SWEEP=0/1) */
  TEMX_OPT_OS_MAP = 2,             /* single sweep: 0 loads of 1 row x 64 columns (default), 1 the MFMA tile of
                                      4 rows x 16 columns (env: TEMX_OS_MAP=tile) */
  TEMX_OPT_OP_MAP = 3,             /* the same for sweep 1 of the class-sum form (env: TEMX_OP_MAP=tile) */
  TEMX_OPT_OS_SUBSAMPLE = 4,       /* class-groups of this plan's rows that enter the reference fit of the single
                                      sweep (default 32; an ncol-sharded job wants about 32 / ranks per rank, at least 8; set
                                      before temx_plan_set_tem builds the tables; env: TEMX_OS_SUBSAMPLE) */
  TEMX_OPT_TRACER_ONE_PASS = 5,    /* temx_tracer_run on the class-sum form: 1 = one-pass tracer stages
                                      (env: TEMX_TRACER_ONE_PASS=1) */
  TEMX_OPT_SINGLE_SWEEP_MIN_GROUPS = 6, /* smallest number of class-groups for which the automatic choice takes the
                                      single sweep (default 640) */
  TEMX_OPT_OS_CONTRACT = 7         /* single sweep, the contraction after it: 0 on the matrix cores (default), 1 the
                                      round-3 form with the matrices staged in LDS (A/B; env: TEMX_OS_CONTRACT=lds) */
};
enum {
  TEMX_FORM_AUTO = -1,
  TEMX_FORM_TWO_PASS = 0,          /* fields read twice (project sweep, eddy sweep) */
  TEMX_FORM_CLASS_SUMS = 1,        /* one pass with per-class sums + flux kernel wherever possible, never the single sweep */
  TEMX_FORM_SINGLE_SWEEP = 2,      /* the single sweep wherever its instantiations exist */
  TEMX_FORM_NO_SINGLE_SWEEP = 3    /* automatic choice between the two forms above by problem size, never the single sweep */
};

/* which matrix temx_get_matrix copies */
enum {
  TEMX_MAT_Y0 = 0,    /* [N][K]   sph_zonal_mean.py:360-363 */
  TEMX_MAT_Y0P = 1,   /* [M][K]   sph_zonal_mean.py:367-370 */
  TEMX_MAT_GRAM = 2,  /* [K][K]   Y0^T Y0 (this rank's rows only until finalised with a global G) */
  TEMX_MAT_GINV = 3,  /* [K][K]   inverse Gram; Y0inv = GINV . Y0^T */
  TEMX_MAT_Y0INV = 4, /* [K][N]   pinv(Y0) as the reference stores it, sph_zonal_mean.py:389 */
  TEMX_MAT_GRAM2 = 5, /* [K][K]   Q^T Q over this rank's rows, Q = Y0 R^-1 the basis the finalised plan projects
                                  on (see temx_plan_finalize); the identity if the plan keeps the Y0 basis */
  TEMX_MAT_GX = 6,    /* [K][2L+1] Y0^T Y0ext over this rank's rows, Y0ext = Y_l^0 up to degree 2L (single sweep;
                                  after temx_plan_set_tem on a plan with temx_plan_single_sweep() == 1) */
  TEMX_MAT_GSUB = 7   /* [KR][KR] Gram matrix of the first KR = min(16, K) harmonics over this rank's share of the
                                  reference subsample (single sweep) */
};

/* order of the ten GM16 Table-A1 results in the results buffer (tem_diagnostics.py:1018-1022) */
enum {
  TEMX_R_VTEM = 0, TEMX_R_OMEGATEM, TEMX_R_WTEM, TEMX_R_PSITEM, TEMX_R_EPFY, TEMX_R_EPFZ,
  TEMX_R_EPDIV, TEMX_R_UTENDEPFD, TEMX_R_UTENDVTEM, TEMX_R_UTENDWTEM, TEMX_NRESULTS
};

/* order of the zonal-grid intermediates (tem_diagnostics.py:1009-1017, zonal ones) */
enum {
  TEMX_Z_UB = 0, TEMX_Z_VB, TEMX_Z_THETAB, TEMX_Z_WAPB, TEMX_Z_UPVPB, TEMX_Z_UPWAPPB,
  TEMX_Z_VPTPB, TEMX_Z_DUB_DP, TEMX_Z_DTHETAB_DP, TEMX_Z_UBCOSLAT, TEMX_Z_DUBCOSLAT_DLAT,
  TEMX_Z_PSI, TEMX_Z_PSICOSLAT, TEMX_Z_DPSICOSLAT_DLAT, TEMX_Z_DPSI_DP, TEMX_Z_INT_VBDP,
  TEMX_NZONAL
};

/* order of the native-grid eddy fields (tem_diagnostics.py:517-529, 547-555) */
enum {
  TEMX_E_UP = 0, TEMX_E_VP, TEMX_E_THETAP, TEMX_E_WAPP, TEMX_E_UPVP, TEMX_E_UPWAPP, TEMX_E_VPTP,
  TEMX_NEDDY
};

int temx_version(void);
const char* temx_last_error(void);
int temx_device_count(void);

/* ---- plan: replaces sph_zonal_averager.__init__ + sph_compute_matrices ----------------------
 * (sph_zonal_mean.py:36-181, 302-422).  Builds Y0 (N x K) and Y0p (M x K), K = L+1, on the
 * device by the normalised Legendre recurrence, the Gram matrix G = Y0^T Y0 with the MFMA
 * projection kernel, and (unless TEMX_DEFER_FINALIZE) factorises G on the host (Cholesky),
 * which replaces lstsq(Y0, I_N) (sph_zonal_mean.py:389): pinv(Y0) = G^-1 Y0^T.
 * lat_deg_host[ncol], lat_out_deg_host[M] in degrees.  L <= 511.  Up to L = 63 (K <= 64) the TEM
 * sweeps are the fused kernels; larger L runs in 64-harmonic slices: on grids with latitude classes
 * and K <= 256 the fields are swept once for per-class sums and the slices work on those sums,
 * otherwise sliced projections, accumulating native reconstructions and an elementwise pass.
 * The sweeps come in three forms, chosen here from the latitudes alone (same operator, results equal
 * up to rounding; temx_plan_sweep_mode tells which):
 *   latitude-class  columns that share |lat| share a basis row (cubed-sphere: 16 per class, lat-lon:
 *                   2 NLON): member rows are added, the matrix work is done once per class and the
 *                   sweeps are HBM streams.  Used when the grid has >= 3 columns per class;
 *                   TEMX_NO_CLASSES / TEMX_NO_CLS=1 disables.
 *   mirror-paired   every column has a mirror column at the opposite latitude: 54 % of the matrix work.
 *   generic         any grid.  TEMX_NO_SYMMETRY / TEMX_NO_SYM=1 forces it.
 * Latitudes are matched to within 1e-11 degrees (1e-8 with TEMX_LAT_TOL_F32; TEMX_SYM_TOL_DEG in the environment
 * sets the tolerance outright). */
int temx_plan_create(temx_plan** out, int device, int64_t ncol, int L, int M,
                     const double* lat_deg_host, const double* lat_out_deg_host, int flags);

/* Replaces lstsq(Y0, I_N) (sph_zonal_mean.py:389).  The Gram matrix G is factorised on the host (Cholesky,
 * long double) and the plan is RE-ORTHOGONALISED (Cholesky-QR2): all basis blocks of the sweeps are rebuilt for
 * Q = Y0 R^-1 (a row of Q still depends on latitude only), the sweeps project on Q, the K x K solve becomes
 * (Q^T Q)^-1 ~ I and Y0p R^-1 takes the coefficients to the output latitudes.  Errors are of order
 * cond(Y0) eps instead of the cond(Y0)^4 eps of an explicit inverse of the normal equations.  Consequences:
 *   - the raw sums of this API (temx_project, B4, B3, Bq, Bq2) are Q^T A, not Y0^T A -- still linear in the
 *     rows, so ncol-sharded callers all-reduce them exactly as before; every rank must finalise with the
 *     SAME G (the all-reduced one);
 *   - the attributes are unchanged: TEMX_MAT_Y0, TEMX_MAT_Y0P, TEMX_MAT_GINV = G^-1, TEMX_MAT_Y0INV = G^-1 Y0^T.
 * Rank-deficient G (fewer distinct latitudes than harmonics): pseudo-inverse of G in the Y0 basis, as before.
 * TEMX_NO_QR=1 in the environment keeps the Y0 basis and the explicit G^-1.
 * ncol-sharded use: each rank creates its plan with TEMX_DEFER_FINALIZE over its own columns, copies its
 * local Gram out (temx_get_matrix(TEMX_MAT_GRAM)), all-reduces it, and hands the global G back here.
 * G_host == NULL finalises with the local Gram (and runs temx_plan_refine(plan, NULL) itself). */
int temx_plan_finalize(temx_plan* plan, const double* G_host /* [K][K] or NULL */);

/* Second pass of the re-orthogonalisation: (Q^T Q)^-1 from G2, the Gram matrix of Q.  G2_host == NULL:
 * computed from this plan's rows (single-process use; temx_plan_finalize(plan, NULL) does it).  ncol-sharded
 * use: all-reduce temx_get_matrix(TEMX_MAT_GRAM2) and hand it over; without this call a sharded plan uses
 * the identity, which leaves errors of order cond(G) eps (fine for cond(G) up to ~1e5).  A no-op for plans
 * that keep the Y0 basis. */
int temx_plan_refine(temx_plan* plan, const double* G2_host /* [K][K] or NULL */);

/* weights mode of the operator (sph_zonal_mean.py:180-181, 383-386): Y0inv = Y0^T diag(4 pi w).
 * Replaces the Gram solve by a per-column scale; call instead of temx_plan_finalize.
 * The operator API (temx_project / temx_zonal_mean) costs what it costs unweighted.  The TEM and tracer
 * stages on a weighted plan run the UNFUSED second sweep (native means materialised, elementwise eddies,
 * weighted projection -- the projection rows differ from the reconstruction rows, which the fused sweeps
 * cannot express): workspace of 8 x ncol x nlev x nt doubles and about 8 x the HBM traffic of the fused
 * path; TEMX_ENOMEM with the byte count when that does not fit -- use smaller blocks of snapshots.
 * (TEMDiagnostics never builds a weighted averager, tem_diagnostics.py:241-247; this is C-ABI only.) */
int temx_plan_set_weights(temx_plan* plan, const double* weights_host /* [ncol], sums to 1 */);

void temx_plan_destroy(temx_plan* plan);

/* 1 if the plan runs the mirror-paired sweeps (equatorially symmetric grid detected), else 0. */
int temx_plan_is_paired(const temx_plan* plan);
/* 0 generic sweeps, 1 mirror-paired sweeps, 2 latitude-class sweeps (columns that share |lat| share a
 * basis row: cubed-sphere, lat-lon and Gaussian grids; TEMX_NO_CLS=1 in the environment disables) */
int temx_plan_sweep_mode(const temx_plan* plan);
/* 1 when (after temx_plan_set_tem) the latitude-class path runs in its one-pass form: sweep 1
 * (temx_tem_stage1) also accumulates, per latitude class and hemisphere, the sums of u v, u omega and
 * v theta, projects them next to the four fields, and stores the four field sums of every class side
 * in the plan; temx_tem_stage2_from_sums then gets the eddy-product sums algebraically (the zonal mean
 * is constant inside a class side) and the fields are read once.  Used from nlev*nt >= 49 and
 * ncol*nlev*nt >= 1.2e7 up (below that the two-pass sweeps are as fast); needs workspace of
 * 8 x 512 B per class-group and d-tile; TEMX_TWO_PASS=1 in the environment disables, TEMX_ONE_PASS=1
 * lifts the size threshold. */
int temx_plan_one_pass(const temx_plan* plan);
/* 1 when temx_tem_run takes the SINGLE-SWEEP form (one-pass class path on a grid with >= 640 class-groups,
 * L <= 51; TEMX_SINGLE_SWEEP=0 / =1 in the environment disables / forces where possible): no per-class sums
 * are stored at all.  One sweep projects the four fields -- minus a band-limited reference of degree <= 15
 * fitted to a subsample of the latitude classes in a short pre-pass -- up to degree 2L, and their three
 * products up to degree L; the eddy-product sums then follow from the Legendre product linearisation
 * (sum_i Y_l abar b = a bilinear form of the coefficients of abar and of the degree-2L projection of b,
 * evaluated on Gauss-Legendre nodes).  Same results to ~1e-12; the staged, all-reducible entry points
 * (temx_tem_stage1 / stage2_from_sums, temx_tracer_stage*) keep the class-sum form.  temx_tracer_run after
 * such a temx_tem_run takes the tracer's own single sweep over (q, v, omega): it reuses the projections and
 * references of v and omega the TEM run left in the plan -- the SAME va, wap must be handed over, as for the
 * one-pass tracer stages -- and any temx_tem_stage1 / temx_plan_set_tem in between sends it back to the other
 * forms.  Both input types take it (fp32 inputs with a two-waves-per-SIMD variant of the sweep). */
int temx_plan_single_sweep(const temx_plan* plan);

int temx_get_matrix(temx_plan* plan, int which, double* dst, void* stream);

/* Path selection, see TEMX_OPT_*.  temx_plan_option returns the value in effect (for TEMX_OPT_FORM: the form
 * the configured plan runs, a TEMX_FORM_* value), -1 for an unknown option. */
int temx_plan_configure(temx_plan* plan, int option, int value);
int temx_plan_option(const temx_plan* plan, int option);

/* ncol-sharded single sweep: every rank hands back the all-reduced TEMX_MAT_GX and TEMX_MAT_GSUB (host
 * pointers) after temx_plan_set_tem; until then a plan that was finalised with an external Gram matrix and whose
 * own subsample cannot be fitted refuses to run (TEMX_ESTATE). */
int temx_plan_set_os_matrices(temx_plan* plan, const double* Gx_host /* [K][2L+1] */, const double* Gsub_host /* [KR][KR] */);

/* ---- operator API: replaces _sph_zonal_mean_generic (sph_zonal_mean.py:187-283) ------------ */

/* B[K][D] = Q^T A  (this rank's columns; raw sums in the plan's projection basis, Q = Y0 R^-1 after
 * temx_plan_finalize, Y0 itself with TEMX_NO_QR=1, weights or a rank-deficient grid).  A is [ncol][D]. */
int temx_project(temx_plan* plan, const void* A, int dtype, int64_t D, double* B, void* stream);

/* out = (Y . Y0inv) . A   (sph_zonal_mean.py:251);  native == 0: Y = Y0p, out is [M][D]
 * (sph_zonal_mean, :291);  native != 0: Y = Y0, out is [ncol][D] (sph_zonal_mean_native, :285).
 * out is fp64. */
int temx_zonal_mean(temx_plan* plan, const void* A, int dtype, int64_t D, double* out,
                    int native, void* stream);

/* Second half of temx_zonal_mean, for ncol-sharded use: B[K][D] (raw sums, all-reduced over the
 * ranks) -> out, [M][D] (native == 0) or this rank's own [ncol][D] block (native != 0). */
int temx_zonal_mean_from_sums(temx_plan* plan, const double* B, int64_t D, double* out,
                              int native, void* stream);

/* ---- TEM pipeline: replaces TEMDiagnostics.__init__ numerics + the ten diagnostics ----------
 * (tem_diagnostics.py:491-611 and :615-797).  plev ascending (the front end flips,
 * tem_diagnostics.py:372-382).  p_pa_host[nlev] = plev*100 (tem_diagnostics.py:385). */
int temx_plan_set_tem(temx_plan* plan, int nlev, int64_t nt, const double* p_pa_host, double p0);

/* stage 1: theta = T (p0/p)^kappa fused into the load (tem_diagnostics.py:498); raw sums
 * B4[4][K][D] = Y0^T {u, v, theta, omega}   (first half of the 4+4 zonal means, :515-529). */
int temx_tem_stage1(temx_plan* plan, const void* ua, const void* va, const void* ta,
                    const void* wap, int dtype, double* B4, void* stream);

/* stage 2: coefficients C = G^-1 B4, zonal means ub vb thetab wapb (Y0p C), then one sweep that
 * reconstructs the native-grid means (Y0 C), forms the eddies x' = x - xbar (:517-529), the
 * products u'v', u'w', v'theta' (:547-555) and projects them: B3[3][K][D] raw sums.
 * Always reads the four fields it is given (they need not be the arrays stage 1 saw, nor unchanged). */
int temx_tem_stage2(temx_plan* plan, const void* ua, const void* va, const void* ta,
                    const void* wap, int dtype, const double* B4, double* B3, void* stream);

/* stage 2 of the one-pass class path (temx_plan_one_pass(plan) == 1): the same B3, from the per-class
 * sums the LATEST temx_tem_stage1 on this plan left in the plan's workspace -- no field is read.  The
 * contract is explicit: the result describes the fields that stage 1 call was given, as they were
 * then.  TEMX_ESTATE when the plan is not in one-pass form or no stage 1 has run since
 * temx_plan_set_tem.  B4 is that stage 1's output (all-reduced over the ranks when ncol-sharded). */
int temx_tem_stage2_from_sums(temx_plan* plan, const double* B4, double* B3, void* stream);

/* stage 3: flux zonal means (:549-557), the derivatives / psi / integral of
 * _compute_derivatives (:574-599) and the ten diagnostics (:615-797) in one fused epilogue.
 * results: [TEMX_NRESULTS][M][D] fp64.  zonal: NULL or [TEMX_NZONAL][M][D] fp64. */
int temx_tem_stage3(temx_plan* plan, const double* B3, double* results, double* zonal,
                    void* stream);

/* all three stages with plan-owned B4/B3 (single-GPU / time-sharded use); stage 2 in its from-sums
 * form when the plan runs the one-pass class path. */
int temx_tem_run(temx_plan* plan, const void* ua, const void* va, const void* ta,
                 const void* wap, int dtype, double* results, double* zonal, void* stream);

/* ---- the single sweep in three steps: the ncol-sharded form of temx_tem_run -------------------------------
 * (temx_plan_single_sweep(plan) == 1).  Every step of the pipeline after the zonal sums -- solve, product
 * linearisation, vertical stencils, epilogue (tem_diagnostics.py:574-797) -- acts along latitude and pressure only,
 * i.e. independently per time snapshot, and the sums are linear in the rows (sph_zonal_mean.py:251).  So an ncol-sharded job
 * exchanges the sums by a REDUCE-SCATTER OVER TIME and every rank finishes the snapshots it receives: nothing of the
 * tail is replicated, the results stay time-sharded.  Per step and rank:
 *   temx_tem_os_prepass  As[4][KR][D] = raw sums of the four fields over the rank's share of the reference
 *                        subsample, first KR = min(16, L+1) harmonics         -> all-reduce (4 KR D doubles)
 *   temx_tem_os_sweep    reference coefficients from As, the ONE sweep over the rank's columns, and its reduction:
 *                        proj = (4 (2L+1) + 3 (L+1)) rows of projections, written as nslices time slices
 *                        [nslices][rows][nlev][ceil(nt / nslices)] (slice w holds the snapshots of
 *                        shard_bounds(nt, nslices, w), rows of nlev x ntw(w) columns packed at its start;
 *                        nslices == 1: [rows][nlev][nt])                      -> reduce-scatter (one slice per rank)
 *   temx_tem_os_tail     proj_slice [rows][nlev][nts] for the snapshots [t0, t0 + nts): the ten results
 *                        [TEMX_NRESULTS][M][nlev][nts] (and the zonal intermediates) for those snapshots.
 * temx_tem_run on such a plan is the three calls with nslices = 1.  After a tail on a proper slice the plan's
 * coefficients describe that slice only: the staged entry points for the whole run (temx_tem_stage3,
 * temx_tracer_stage2 ..., temx_tem_eddy) return TEMX_ESTATE until a whole-run stage 1 / 2 has run again.
 * Tracers (after temx_tem_os_tail on the same snapshots, same va / wap), nq = 1 or 2 per sweep -- two tracers share
 * one read of v and omega: temx_tracers_os_prepass (Asq[nq][KR][D], all-reduce), temx_tracers_os_sweep (projq:
 * nq (2L+1) + 2 nq (L+1) rows -- the q's, then q1 v, q1 omega, q2 v, q2 omega --, sliced like proj),
 * temx_tracers_os_tail (tres_host / tzon_host: host arrays of nq device pointers, [TEMX_NTRES][M][nlev][nts] /
 * [TEMX_NTZON][M][nlev][nts] each; tzon_host may be NULL).  q_host: host array of nq device pointers. */
int temx_tem_os_prepass(temx_plan* plan, const void* ua, const void* va, const void* ta, const void* wap, int dtype,
                        double* As, void* stream);
int temx_tem_os_sweep(temx_plan* plan, const void* ua, const void* va, const void* ta, const void* wap, int dtype,
                      const double* As, int nslices, double* proj, void* stream);
int temx_tem_os_tail(temx_plan* plan, const double* proj_slice, int64_t t0, int64_t nts, double* results,
                     double* zonal, void* stream);
int temx_tracers_os_prepass(temx_plan* plan, int nq, const void* const* q_host, const void* va, const void* wap,
                            int dtype, double* Asq, void* stream);
int temx_tracers_os_sweep(temx_plan* plan, int nq, const void* const* q_host, const void* va, const void* wap,
                          int dtype, const double* Asq, int nslices, double* projq, void* stream);
int temx_tracers_os_tail(temx_plan* plan, int nq, const double* projq_slice, double* const* tres_host,
                         double* const* tzon_host, void* stream);

/* The same time-sliced tail for raw sums of ANY form of the sweeps (class-sum form, paired, generic):
 * B4s [4][K][nlev][nts] and B3s [3][K][nlev][nts] (summed over the ranks) -> stages 2b + 3 for the snapshots
 * [t0, t0 + nts).  temx_time_slices cuts whole arrays of `rows` rows x [nlev][nt] into the reduce-scatter layout
 * described above. */
int temx_tem_tail_from_sums(temx_plan* plan, const double* B4s, const double* B3s, int64_t t0, int64_t nts,
                            double* results, double* zonal, void* stream);
int temx_time_slices(temx_plan* plan, const double* B, int64_t rows, int nslices, double* out, void* stream);

/* lazily materialise the native-grid eddy fields (properties up vp thetap wapp upvp upwapp vptp,
 * tem_diagnostics.py:420-433).  Needs the coefficients of a previous stage2/run on this plan.
 * eddy_host_ptrs: TEMX_NEDDY device pointers ([ncol][D] fp64 each), NULL entries are skipped. */
int temx_tem_eddy(temx_plan* plan, const void* ua, const void* va, const void* ta,
                  const void* wap, int dtype, double* const* eddy_ptrs_host, void* stream);

/* The same seven arrays for the native rows [row0, row0 + nrows) only (row0 a multiple of 16), written
 * compactly as [nrows][nlev][nt]: lets a caller stream the native-grid attributes of a large run to
 * the host or to a file chunk by chunk with bounded device memory. */
int temx_tem_eddy_rows(temx_plan* plan, const void* ua, const void* va, const void* ta, const void* wap,
                       int dtype, int64_t row0, int64_t nrows, double* const* eddy_ptrs_host /* [7], host array */,
                       void* stream);

/* ---- tracer TEM (Abalos+ 2017): replaces the tracer parts of _decompose_zm_eddy, _compute_fluxes,
 * _compute_derivatives (tem_diagnostics.py:532-538, 560-570, 602-611) and etfy ... qtendwtem
 * (:801-991) for ONE tracer q [ncol][D].  Needs the plan state of a preceding temx_tem_run /
 * temx_tem_stage3 on the same fields (coefficients of v and omega, psi, vtem, omegatem).
 * Staged like the TEM pipeline so that ncol-sharded callers can all-reduce the raw sums:
 *   stage1: Bq[K][D]    = Y0^T q
 *   stage2: Bq2[2][K][D] = Y0^T {q'v', q'w'}   (one sweep: reconstruct qbar, vbar, wbar)
 *   stage3: tres[TEMX_NTRES][M][D] fp64, tzon NULL or [TEMX_NTZON][M][D] fp64. */
enum { TEMX_T_ETFY = 0, TEMX_T_ETFZ, TEMX_T_ETDIV, TEMX_T_QTENDETFD, TEMX_T_QTENDVTEM, TEMX_T_QTENDWTEM,
       TEMX_NTRES };
enum { TEMX_TZ_QB = 0, TEMX_TZ_QPVPB, TEMX_TZ_QPWAPPB, TEMX_TZ_DQB_DP, TEMX_TZ_QBCOSLAT,
       TEMX_TZ_DQBCOSLAT_DLAT, TEMX_NTZON };
int temx_tracer_stage1(temx_plan* plan, const void* q, int dtype, double* Bq, void* stream);
int temx_tracer_stage2(temx_plan* plan, const void* q, const void* va, const void* wap, int dtype,
                       const double* Bq, double* Bq2, void* stream);
int temx_tracer_stage3(temx_plan* plan, const double* Bq2, double* tres, double* tzon, void* stream);
/* One-pass form of stages 1 and 2 (temx_plan_one_pass(plan) == 1 and the plan holds the class sums of a
 * temx_tem_stage1 / temx_tem_run on the SAME va, wap): stage1_sums reads (q, v, omega) once -- raw sums
 * Bq, and per latitude class the sum of q and the centred co-moments of q v, q omega; stage2_from_sums
 * forms Bq2 from those and the TEM run's class sums of v and omega without reading a field.
 * TEMX_ESTATE when the plan is not in that state.  temx_tracer_run uses the two-pass stages unless
 * TEMX_TRACER_ONE_PASS=1 is set in the environment (measured on ne120 x 72 x 30: 10.2 ms one-pass,
 * 9.6 ms two-pass -- the one-pass sweep runs fewer waves per SIMD). */
int temx_tracer_stage1_sums(temx_plan* plan, const void* q, const void* va, const void* wap, int dtype,
                            double* Bq, void* stream);
int temx_tracer_stage2_from_sums(temx_plan* plan, const double* Bq, double* Bq2, void* stream);
int temx_tracer_run(temx_plan* plan, const void* q, const void* va, const void* wap, int dtype,
                    double* tres, double* tzon, void* stream);
/* nq tracers of one TEM run (the reference takes a LIST of tracers, tem_diagnostics.py:281-301, 532-538, 560-570).
 * After a single-sweep temx_tem_run they are swept in PAIRS: (q1, q2, v, omega) read once -- the traffic of the TEM
 * sweep for two tracers instead of three quarters of it for each; a last odd one, and every tracer on any other path,
 * goes through temx_tracer_run.  q_host, tres_host, tzon_host (may be NULL): host arrays of nq device pointers. */
int temx_tracers_run(temx_plan* plan, int nq, const void* const* q_host, const void* va, const void* wap, int dtype,
                     double* const* tres_host, double* const* tzon_host, void* stream);
/* TEM and ONE tracer that is known up front (TEMDiagnostics(q=...), tem_diagnostics.py:241-259 runs both in
 * its constructor): temx_tem_tracer_stage1 is temx_tem_stage1 + temx_tracer_stage1_sums in ONE sweep over
 * (u, v, T, omega, q) -- 40 bytes per grid point instead of 32 + 24 -- with the four waves of a workgroup
 * sharing a d-tile (ten projections).  Needs the one-pass class path (TEMX_ESTATE otherwise); q has the
 * dtype of the fields.  Follow with temx_tem_stage2_from_sums, temx_tem_stage3, temx_tracer_stage2_from_sums,
 * temx_tracer_stage3 (B4, Bq are the all-reduce payloads when ncol-sharded).  temx_tem_tracer_run does all of
 * that with plan-owned sums, and on any other path runs temx_tem_run and temx_tracer_run one after the other. */
int temx_tem_tracer_stage1(temx_plan* plan, const void* ua, const void* va, const void* ta, const void* wap,
                           const void* q, int dtype, double* B4, double* Bq, void* stream);
int temx_tem_tracer_run(temx_plan* plan, const void* ua, const void* va, const void* ta, const void* wap,
                        const void* q, int dtype, double* results, double* zonal, double* tres, double* tzon,
                        void* stream);
/* lazily materialise qp, qpvp, qpwapp ([ncol][D] fp64 each; NULL entries skipped) of the tracer
 * whose temx_tracer_stage2 / run was the last one on this plan (tem_diagnostics.py:537, 563-567). */
int temx_tracer_eddy(temx_plan* plan, const void* q, const void* va, const void* wap, int dtype,
                     double* const* ptrs3_host, void* stream);

/* Synchronises the stream and reports whether any non-finite value reached the zonal sums
 * since the last call (the reference raises on NaN input, sph_zonal_mean.py:219-221). */
int temx_status(temx_plan* plan, int* nonfinite, void* stream);

/* ---- measurement helpers (bench / tests only; not on the product path) ---------------------- */

/* Deterministic synthetic fields of SURVEY section 8(d) written in place on the device:
 * analytic part + 0.1 * N(0,1) noise from a counter hash (splitmix64 -> Box-Muller).
 * lat/lon in degrees (device), plev_hpa (device). */
int temx_synth_fields(int device, int64_t ncol, int nlev, int64_t nt, int64_t t0,
                      const double* lat_deg, const double* lon_deg, const double* plev_hpa,
                      int dtype, uint64_t seed, void* ua, void* va, void* ta, void* wap,
                      void* stream);

/* Bare v_mfma_f64_16x16x4_f64 issue-rate micro-benchmark: returns achieved TFLOP/s. */
int temx_mfma_f64_peak(int device, int iters, double* tflops_out);

/* Kernel-level timing hooks used by bench.py's roofline block: average duration (ms) of the
 * dominant kernels over the launches since the last reset, measured with HIP events recorded
 * on the launch stream. which: 0 = project sweep, 1 = eddy/flux sweep. */
int temx_kernel_timing(temx_plan* plan, int enable);
int temx_kernel_timing_read(temx_plan* plan, int which, double* avg_ms, int* launches);

/* Self-test of the exception barrier of this ABI (needs no device): raises inside the library std::bad_alloc
 * (kind 0), std::runtime_error (kind 1) or a non-standard exception (kind 2) and must RETURN TEMX_ENOMEM /
 * TEMX_EINTERNAL / TEMX_EINTERNAL with temx_last_error() set; any other kind returns TEMX_EINVAL. */
int temx_selftest_exception(int kind);

#ifdef __cplusplus
}
#endif
#endif /* TEMX_H */
