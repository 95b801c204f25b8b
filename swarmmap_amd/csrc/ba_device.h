// ba_device.h — device-side layout of the bundle-adjustment solver (FP64 throughout, like g2o).
//
// HBM layout (one so_ba = one agent's local-mapping solver context; buffers grow on demand, stay resident):
//   poses     2 buffers (current / trial) of Pose{q[4] xyzw, t[3], pad}      n_poses x 64 B
//   points    2 buffers (current / trial) of double[3]                        n_points x 24 B
//   edges     sorted by point (stable): pose idx, point idx, obs[2], inv_sigma2, active flag
//             + stored error[2] (EdgeSE3ProjectXYZ::_error) that only active edges refresh
//   CSR       point -> its edges (contiguous after the sort); free pose -> its edges (ascending, i.e. by landmark);
//             edge_tab[hessian index][landmark] -> edge, filled on the device: block (i1,i2) of the reduced camera
//             system walks pose i1's edges and looks the partner edge of pose i2 up.  Built once per problem;
//             the second stage keeps everything and skips edges whose active flag was cleared.
//   system    Hpp[n_free][36], bp[n_free][6], Hll[n_pt][9], bl[n_pt][3], W[edge][18] = J_pose^T w J_point,
//             Dinv[n_pt][9], db[n_pt][3], BDinv[edge][18], S[(6 n_free)^2] dense, bs / x_p[6 n_free], x_l[n_pt][3]
// Summation orders are fixed (CSR order, fixed reduction trees, no floating-point atomics) so results are
// deterministic run to run.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <vector>
#include <stdint.h>

namespace so {

struct BaPose {
    double q[4];  // x y z w
    double t[3];
    double pad;
};

// Levenberg-Marquardt state of one optimize() call (OptimizationAlgorithmLevenberg::solve +
// SparseOptimizer::optimize), resident in HBM: every kernel of a trial reads it, ba_trial_decide_kernel
// advances it, so the host enqueues trials back to back without reading anything in between.
struct BaLm {
    double lambda, ni;          // _currentLambda, _ni
    double currentChi, iniChi;  // chi2 of the current estimate; chi2 at the start of the iteration
    double tempChi, rho;        // last trial
    double chi_out;             // activeRobustChi2 after the last completed iteration
    double chi_begin;           // chi2 of the estimate optimize() started from
    int cur;                    // which of the two estimate buffers is the current one (the other is the trial)
    int it, iterations;         // iteration counter of this optimize() and its limit
    int qmax, nBad;
    int done;                   // iterations completed in this optimize()
    int trials;                 // trials since the problem was set up (all stages)
    int active;                 // the running stage's tag (BaDev::stage) while optimize() wants another trial, 0 otherwise
    int need_build;             // the current estimate changed: re-linearise before the next trial
    int stages_begun;           // optimize() calls started on this problem (ba_stage_begin_kernel)
    int prev_done;              // `done` of the stage before the current one, saved when the current one began
    int pad;
    double prev_chi_out;        // `chi_out` of the stage before the current one
    double prev_chi_begin;      // `chi_begin` of the stage before the current one
};

// Block-Jacobi PCG on the reduced camera system (ba_pcg.hip): workspace pointers (device) ...
struct BaPcgDev {
    const int* indptr;   // n_free + 1: block rows of S - its nonzero 6 x 6 block columns, ascending (diagonal included)
    const int* indices;
    double* Sc;          // the nonzero blocks of S, 36 doubles each in the order of `indices` (copied per solve; null: read S itself)
    double* Minv;        // n_free x 36: inverse of the diagonal blocks
    double *x, *r, *z, *p, *Sp;  // 6 n_free each
    double* partA;       // ceil(n_free / 4): partial sums of p.Sp
    double* partB;       // 2 x ceil(n_free / 256): partial sums of r.z | r.r
    double* scal;        // rz by iteration parity [0..1], b.b, tol^2 b.b, p.Sp, r.r
    unsigned long long* status;       // device word: solve sequence number << 32 | iterations << 2 | failed << 1 | converged
    unsigned long long* status_host;  // the same in host-mapped memory (device-side address)
};
// ... and the host side of a solver context's PCG (not read by kernels)
struct BaPcgHost {
    BaPcgDev dev{};
    volatile unsigned long long* status_host = nullptr;
    unsigned seq = 0;
    double tol = 1e-7;   // relative residual |r| / |b|
    int max_it = 4000;
    long long iterations = 0;  // of this so_bundle_adjust call
    int solves = 0;
    int fault = 0;             // of this call: 1 = a solve did not finish within 30 s, 2 = the stream failed / drained mid-solve
    int wide = 0;              // 1: a block row of S holds enough blocks for a whole workgroup (pcg_spmv_wide_kernel); partA then has n_free entries
};

struct BaDev {
    // estimate buffers and LM state
    BaPose* pose[2];
    double* pt[2];
    BaLm* lm;
    // problem
    int n_poses, n_points, n_edges;
    int n_free;          // poses with a hessian index in the current stage
    int stage;           // tag of the optimize() call these launches belong to (1, 2): they run only while lm->active == stage
    int n_active_points;  // informational
    const double* intr;       // n_poses x 4
    const int* e_pose;        // sorted edges
    const int* e_point;
    const double* e_obs;      // x2
    const double* e_w;        // inv_sigma2
    uint8_t* e_active;        // level == 0 in this stage (ba_mark_outliers_kernel clears flags between stages)
    double* e_err;            // x2 stored error
    double* e_chi2;           // stored chi2 (refreshed with the error)
    const int* pt_off;        // n_points + 1
    uint8_t* pt_active;       // n_points: has at least one active edge
    const int* pose_hidx;     // n_poses: hessian index or -1
    const int* free_pose;     // n_free: pose index of hessian index i
    const int* pose_off;      // n_free + 1
    const int* pose_edges;    // edge ids per free pose
    const int* pose_edge_point;  // landmark of each of those edges (= e_point[pose_edges[.]])
    int* edge_tab;            // n_free x n_points: the ACTIVE edge joining (hessian index, landmark), -1 if none
                              // (ba_mark_outliers_kernel removes the edges it drops)
    // large maps (blocked-solver path): per upper block (i1 < i2) the (edge, edge) pairs of the landmarks both
    // keyframes see, built on the device; blocks with at most kBaSmallBlockPairs pairs are summed by one thread each
    int use_pairs;
    int fold_prep;  // local windows (no pair lists, single-workgroup solver): no ba_schur_prep launch, its products are formed by their consumers
    int* pr_off;              // n_blk + 1: exclusive scan of the pair counts (block g = i2 (i2 + 1) / 2 + i1)
    int* pr_cur;              // n_blk: counts, then fill cursors
    int* pr_l; int* pr_k1; int* pr_k2;  // landmark, edge of i1, edge of i2 (arrival order)
    int* ps_l; int* ps_k1; int* ps_k2;  // the same, every block sorted by landmark: what the gathers read
    int* big_list; int* big_n;          // blocks left to the wave-per-block walk: the diagonal and the crowded ones
    int big_cap;                        // capacity of big_list = launch bound of the walk
    // system
    double* Hpp; double* bp; double* Hll; double* bl; double* W;
    double* Dinv; double* db; double* BDinv;
    double* S; double* bs; double* xl;
    int ldS;           // leading dimension of S: 6 n_free, or that rounded up to 96 on the blocked dense path
    double* dense_ws;  // blocked dense path: inverted 96x96 diagonal blocks, one per panel
    double* dense_x;   //                     solution staging (ldS doubles)
    hipStream_t dense_side;     // host-side handles of the blocked path's look-ahead (not read by kernels)
    hipEvent_t* dense_events;   // 1 + 2 * kDenseMaxPanels events
    // block-skyline structure of S on the blocked path (96-row tiles): tile (I, J), J <= I, can be nonzero in S and
    // in its Cholesky factor only for J >= tile_first[I] (row envelope; Cholesky fill stays inside it)
    const int* tile_first;      // ldS / 96 entries (device)
    const int2* plan_tiles;     // (I, J) tiles of every trailing-update launch, in launch order (device)
    const struct DensePlan* plan;  // host side of the same (not read by kernels)
    // single-launch dataflow solve (dense_flow_kernel): the skyline's tiles column by column, epoch-stamped ready
    // flags, the tiles' contributions to the two substitutions; flow_tiles == nullptr: the multi-launch path
    const int2* flow_tiles;
    unsigned* flow_flags;       // kFlowFlagWords, zeroed once when allocated
    double* flow_vec;           // 256 x 96: L_IJ y_J of every off-diagonal tile
    unsigned* flow_epoch;       // host counter of the solver context, grows with every solve (not read by kernels)
    unsigned* flow_abort_host;  // host-mapped word a workgroup stamps with the epoch when a wait outlasts its budget
    unsigned long long flow_timeout_ticks;  // that budget in wall_clock64() ticks (100 MHz)
    int flow_nslots;            // ticketed kernel (large skylines): tile flag slots, T (T + 1) / 2
    int flow_grid;              //                  resident workgroups of its launch
    double* partial;  // reduction partials (chi2 | scale) + flags
    int use_pcg;          // blocked path with pair lists only: the reduced system is solved by block-Jacobi PCG (ba_pcg.hip)
    BaPcgHost* pcg_host;  // (host side; not read by kernels)
    int robust;
    double huber_delta;
    float huber_dsqr;  // stored as float in the reference (robust_kernel_impl.h:84)
};

// ---- local bundle adjustments of SEVERAL agents as one chain of launches (so_ba_group, ba.cpp) ----
// A member's so_bundle_adjust records the launches of its LM chain (launch_ba_* below append to g_ba_recorder instead of
// launching); the group merges the members' lists step by step: launches of the same kind go out as ONE launch with the
// member as blockIdx.y (rows of a BaDev table in HBM + per-member scalars in the argument block), the others one by one.
constexpr int kBaGroupMax = 8;
struct BaGroupArgs {
    int n;
    int grid[kBaGroupMax];  // the member's own gridDim.x: workgroups beyond it return; grid-stride loops and per-workgroup partials use it
    int row[kBaGroupMax];   // its row of the BaDev table
    int i0[kBaGroupMax], i1[kBaGroupMax], i2[kBaGroupMax];
    double f0[kBaGroupMax];
    void* p0[kBaGroupMax];
    void* p1[kBaGroupMax];
    void* p2[kBaGroupMax];
    void* p3[kBaGroupMax];
};
enum BaLaunchKind {
    kBaKErrors, kBaKBuild, kBaKStageBegin, kBaKGather4, kBaKSolveMfma, kBaKUpdateErrors, kBaKDecide, kBaKMarkOutliers, kBaKFinish,
    kBaKSignal, kBaKSolo, kBaKCount
};
struct BaLaunchRec {
    int kind = kBaKSolo;
    BaDev d{};
    int grid = 1;
    int i0 = 0, i1 = 0, i2 = 0;
    double f0 = 0.0;
    void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr, *p3 = nullptr;
    size_t lds = 0;
    int phase = 0;  // BaRecorder::phase when it was recorded: the group merges the members' lists phase by phase
    std::function<void(hipStream_t)> solo;  // kBaKSolo: a launch that has no grouped form, issued as it is
};
// phases of a call's chain: 0 stage 1's prologue, 1 + k its trial k, 100 stage 2's prologue, 101 + k its trial k, 1000 the epilogue
constexpr int kBaPhaseStage2 = 100, kBaPhaseEpilogue = 1000;
struct BaRecorder {
    std::vector<BaLaunchRec> list;
    int phase = 0;
};
// ---- a stage's Levenberg-Marquardt trials as ONE resident launch (ba_kernels.hip: ba_lm_resident_kernel) ----
// Device words the workgroups of such a launch meet at; one block per solver context, zeroed at creation.
struct BaResidentSync {
    unsigned arrive;   // grid barrier: workgroups that have arrived (back to 0 when the barrier opens)
    unsigned gen;      // grid barrier: (launch epoch << 11) | ordinal of the last barrier that opened
    unsigned abort_w;  // epoch of a launch in which a workgroup gave up waiting (BaDev::flow_timeout_ticks)
    unsigned pad;
};
constexpr int kBaResidentMaxTrials = 400;  // (five barriers per trial, and the barrier's ordinal has 11 bits beside the epoch's 20)
// false: this window cannot run resident (solver class, size); nothing was launched
bool launch_ba_trials_resident(const BaDev& d, int nb_upd, int max_trials, const uint8_t* abort_flag, BaLm* lm_host,
                               BaResidentSync* sync, unsigned epoch, int n_workgroups, hipStream_t s);

extern thread_local BaRecorder* g_ba_recorder;  // non-null: the calling thread's launch_ba_* record instead of launching
// one grouped launch: `kind` for the members of A (rows of d_rows), grid (max_grid, A.n)
void launch_ba_group(int kind, const BaDev* d_rows, const BaGroupArgs& A, int max_grid, size_t lds, hipStream_t s);
void launch_ba_solve_mfma_group(const BaDev* d_rows, const BaGroupArgs& A, size_t lds, hipStream_t s);

constexpr int kBaPartialChi = 0;       // [0, 1024): chi2 partials of the error kernel
constexpr int kBaPartialScale = 1024;  // [1024, 2048): scale partials of the update kernel
constexpr int kBaMaxDiag = 2048;       // [2048]: max |diag|
constexpr int kBaSolveOk = 2049;       // [2049]: 1.0 if the Cholesky succeeded
constexpr int kBaPartialCount = 2056;

// which: 0 = current estimate, 1 = trial.  gated: return immediately unless lm->active (and, for build, lm->need_build).
// several device buffers filled with a 32-bit pattern in one launch (sizes in bytes, multiples of 4; 16-byte aligned)
struct BaClearItem {
    void* p;
    size_t bytes;
    uint32_t value;
};
struct BaClearList {
    BaClearItem item[6];
    int n;
};
void launch_ba_clear(const BaClearList& L, hipStream_t s);
// gate: 0 = run, kBaGateActive = only while optimize() wants another trial, kBaGateIdle = only when no stage is running (a launch
// chained behind a stage's trials: it takes effect on the device the moment that stage is over, without a host round trip)
constexpr int kBaGateNone = 0, kBaGateActive = 1, kBaGateIdle = 2;
void launch_ba_errors(const BaDev& d, int which, int gate, int n_blocks, hipStream_t s);
void launch_ba_build(const BaDev& d, int gate, hipStream_t s);
// start of optimize(iterations): currentChi from the error partials, computeLambdaInit from the max diagonal
void launch_ba_stage_begin(const BaDev& d, int nb_err, int iterations, BaLm* lm_host, int gate, const uint8_t* abort_flag, hipStream_t s);
// one LM trial, seven launches, no host involvement: [build] -> schur prep -> gather -> solve -> update -> errors
// -> decide.  abort_flag (host-mapped, may be null) = g2o's forceStopFlag; lm_host (host-mapped) receives a copy
// of the state after every decision; ev0/ev1 (may be null) bracket the solve kernel.
void launch_ba_trial(const BaDev& d, int nb_err, int nb_upd, const uint8_t* abort_flag, BaLm* lm_host,
                     hipEvent_t ev0, hipEvent_t ev1, hipStream_t s);
void launch_ba_edge_table(const BaDev& d, hipStream_t s);
constexpr int kBaSmallBlockPairs = 8;
size_t ba_pairs_scan_temp_bytes(int n_blk);
// once per problem on the blocked-solver path: count, scan, fill the pair lists, classify the blocks
void launch_ba_build_pairs(const BaDev& d, void* scan_temp, size_t scan_temp_bytes, hipStream_t s);
// blocked dense path (ba_dense.hip), used when the system is too large for one workgroup (n_free > 43)
constexpr int kBaSmallSolverMaxFree = 43;
constexpr int kBaPairsMinFree = 80;       // from here on the Schur gather walks per-block pair lists instead of edge_tab
                                          // (window wall time, edge_tab vs pair lists: 44 keyframes 3.2 vs 3.7 ms, 64: 6.1 vs 6.3,
                                          // 96: 9.2 vs 8.9, 128: 12.8 vs 11.7)
constexpr int kBaMfmaSolverMinFree = 4;   // below: the register-resident look-ahead solver is as fast (measured 3..16)
constexpr int kFlowFlagWords = 1024;
constexpr int kFlowDefaultMaxTiles = 231;  // tiles the single-launch solve takes on (all its workgroups must be resident)
int dense_flow_max_tiles();
int launch_occupy(int workgroups, int lds_bytes, int ms, unsigned* d_sink, hipStream_t s);  // diagnostic load (so_runtime_occupy)
int dense_flow_resident_capacity(int device);  // workgroups of a dataflow launch the device can keep resident (occupancy x CUs)
constexpr int kDenseMaxPanels = 512;  // 49152 / 96: 8192 free keyframes, 19 GB of FP64 when stored densely
// Launch plan of the blocked solver for one problem structure: which tiles each trailing update touches.  Built on the
// host once per so_bundle_adjust call (the structure does not change between LM trials).
struct DensePlan {
    struct Update { int k, kw, rhs_panel, first_tile, n_tiles, n_rhs_blocks; };
    int T = 0, G = 1;
    bool lookahead = false;
    std::vector<Update> updates;  // in the order launch_ba_dense_solve issues them
    std::vector<int2> tiles;
    double flop_structural = 0.0; // FP64 flop of the factorisation + substitutions over nonzero tiles only
    double flop_dense = 0.0;      // n^3 / 3 + 2 n^2
    long long nnz_tiles = 0;      // tiles inside the envelope (lower triangle incl. diagonal)
    int flow_first_tile = 0, flow_n_tiles = 0;  // dataflow solve: its tile list inside `tiles` (0 tiles: not eligible)
    bool flow_big = false;        // more tiles than one launch can keep resident one per workgroup: the ticketed kernel
};
// flow_grid: workgroups the single-launch solve may keep resident (0: the chain-of-launches plan only)
void build_dense_plan(int T, const int* tile_first, bool has_side_stream, DensePlan* plan, int flow_grid);
bool launch_ba_solve_mfma(const BaDev& d, hipStream_t s);   // single-workgroup MFMA solves (4..29 free keyframes: tiles in LDS; 30..43: tiles in registers); false if neither applies
// ba_pcg.hip: block structure of S from the pair lists (indices_or_null == nullptr: counts + scan into indptr; otherwise
// the fill), and one PCG solve (S x = bs, x -> bs; looks at a host-mapped status word between chunks of iterations)
void launch_ba_pcg_structure(const BaDev& d, int* counts, int* indptr, int* indices_or_null, hipStream_t s);
void launch_ba_pcg_solve(const BaDev& d, hipStream_t s);
void launch_ba_dense_pad(const BaDev& d, hipStream_t s);    // once per problem: identity padding up to ldS
void launch_ba_dense_solve(const BaDev& d, hipStream_t s);  // per trial, in place of the single-workgroup solve  // fills edge_tab (memset to -1 beforehand)
// Optimizer.cc:644-656 on the device: edges of the current estimate with chi2 > threshold or non-positive depth
// leave the problem (level 1); landmarks left without an edge become inactive
void launch_ba_mark_outliers(const BaDev& d, double chi2_threshold, int gate, hipStream_t s);
// Optimizer.cc:682-739: current estimate, stored chi2 and the outlier flag (chi2 > threshold or depth <= 0) of
// every edge into the result block
void launch_ba_finish(const BaDev& d, double chi2_threshold, BaPose* pose_out, double* pt_out, double* chi2_out,
                      uint8_t* outlier_out, hipStream_t s);
void launch_ba_signal(int* word_host_mapped, int value, hipStream_t s);  // the word is stored when everything before it is done

struct PoseOptArgs {  // Optimizer::PoseOptimization, one workgroup (ba_kernels.hip)
    const float* Xw;          // n x 3
    const float* obs;         // n x 2
    const float* inv_sigma2;  // n
    double K[4];
    BaPose init;
    int n;
    double* err;       // n x 2 scratch (stored _error)
    uint8_t* outlier;  // n (out)
    BaPose* pose_out;
    int* info;         // [0] nBad, [1] LM iterations, [2] LM trials
    double* trace;     // optional: 4 doubles per LM trial (lambda, tempChi, rho, currentChi), 256 trials max
    int done_seq;      // != 0: the results are in host-mapped memory and info[3] = done_seq is stored behind them
                       // (system scope), so the host can spin on that word instead of waiting for the stream
    // Indexed inputs (pose_opt_chain_kernel: a tracking stage whose matches were resolved on the device, match_device.h
    // TrackResolveArgs): edge e = keypoint e_kp[e] of the frame against map slot e_slot[e]; head[0] = number of edges,
    // head[2] != 0 = the resolve gave up (the kernel then only publishes info[0] = -1).  All null / zero otherwise.
    const int32_t* e_kp = nullptr;
    const int32_t* e_slot = nullptr;
    const int32_t* head = nullptr;
    const float* map_Xw = nullptr;      // the map table's positions, 3 per slot
    const float2* kp_xy_un = nullptr;   // the frame's undistorted keypoints by index
    const int8_t* kp_octave = nullptr;
    float lvl_inv_sigma2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int32_t* kp_slot_clear = nullptr;   // != null: bindings by keypoint index; the outlier edges' entries are set to -1 at the end
                                        // (Tracking.cc:745-760: an outlier of TrackWithMotionModel's pose loses its map point)
};
// the indexed form, one launch: reads the edge count on the device and runs the register-resident body that fits it.
// range 0: up to 1024 edges, 1: 1025 .. 1792; a count outside the launched range, or a resolve that gave up: info[0] = -1
// and the caller takes the host path (fewer than three edges: -2, nothing to optimise)
void launch_pose_opt_chain(const PoseOptArgs& a, int range, hipStream_t s);
// rows of a so_track_group's table in HBM; range_mask bit r: a member's stage needed RANGE r last time
void launch_pose_opt_chain_group(const PoseOptArgs* d_tab, int n, int range_mask, hipStream_t s);
constexpr int kPoseChainMaxEdges = 1792;
// Converter::toSE3Quat(Tcw) / Converter::toCvMat(SE3Quat) as so_pose_optimization applies them (ba.cpp)
void pose_from_Tcw12(const float* Tcw12, BaPose& P);
void pose_to_Tcw12(const BaPose& P, float* Tcw12);
constexpr int kPoseOptLdsMax = 3072;  // matched points the LDS-resident kernel holds (41 B each)
void launch_pose_opt(const PoseOptArgs& a, hipStream_t s);
// n_problems problems in one launch (a workgroup each); d_args must be device-visible.  false: a problem has more than
// 1024 points (the caller falls back to one launch per problem)
bool launch_pose_opt_batch(const PoseOptArgs* d_args, int n_problems, int max_n, hipStream_t s);

}  // namespace so
