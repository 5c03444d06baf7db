// spike_internal.h -- shared declarations between the HIP kernels and the C-ABI engine.
// gfx950 only.  Not part of the public interface (see include/spike_mi355.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spike {

constexpr int BLK = 64;  // partition boundaries fall on multiples of 64 rows

// One chain of row blocks swept sequentially by one workgroup (a SPIKE partition, a sub-chain of one, or -- twisted
// factorisation -- one HALF of a partition).
//
// Two index spaces.  FACTOR space ("virtual rows"): where the chain's matrix rows live in the LU scratch, the packed tiles,
// 1/diag, the intermediate vector of an apply and every setup-time vector: rows row0 .. row0 + nrows - 1, chain-local row r
// at row0 + r.  VECTOR space: where chain-local row r lives in the caller's vectors x and y: vec0 + vdir * r.  For an
// ordinary chain the two coincide (vec0 = row0, vdir = +1).  The BOTTOM half of a twisted partition is factored from the
// partition's last row upward: its chain-local row 0 is the partition's last row (vec0 = that row, vdir = -1) and its band is
// stored flipped (rows and diagonals reversed), so that factorisation, packing and the spike solves see an ordinary chain
// whose first row is the partition's OUTER end and whose last row is the seam in the middle of the partition.
struct ChainDesc {
    int64_t row0;    // first row in factor space
    int32_t nrows;   // rows of the chain
    int32_t nsteps;  // ceil(nrows / R)
    int64_t vec0;    // vector index of chain-local row 0
    int32_t vdir;    // +1 | -1
    int32_t flags;   // CHAIN_HAS_TOP | CHAIN_HAS_BOT: a coupling block exists above the first / below the last chain-local row
};
constexpr int CHAIN_HAS_TOP = 1, CHAIN_HAS_BOT = 2;

// One workgroup = NW waves sweeping CPW = 64/R chains in lock-step.
struct GroupDesc {
    int64_t tile0;     // index of the group's first tile (tiles of a group are consecutive steps)
    int32_t maxsteps;  // max nsteps over the group's chains
    int32_t pad;
};

// Kernel configuration picked from the half-bandwidth.
struct SweepCfg {
    int R;    // rows per block (4,8,16,32,64)
    int DPW;  // diagonals per wave (= tile entries per lane per wave)
    int NW;   // waves per chain
    bool scan = false;  // K = 1: no tiles, one multiplier per row, wavefront scan (k_scan_sweep)
    bool nscan = false; // scan with four rows per lane (k_nscan_*, K = 1..3): coefficient arrays [K][lds]
    // sweep-time shape of PCApply (0 = the base shape above): the tile layout is independent of how the KP diagonals are
    // dealt to waves, so setup may pick another (DPW, NW, prefetch depth) for the apply sweeps (launch_sweep)
    int sDPW = 0, sNW = 0, sPF = 0;
    int basePF() const { return R == 4 ? 12 : R == 8 ? 8 : R == 16 ? 4 : (DPW == 64 ? 1 : 2); }
    int KP() const { return scan ? 1 : DPW * NW; }
    int CPW() const { return 64 / R; }
    int64_t tile_doubles() const { return (int64_t)NW * DPW * 64; }
};

struct SweepArgs {
    const double *tiles;
    const GroupDesc *groups;
    const ChainDesc *chains;
    int nchains;
    const double *in;
    double *out;
    const double *dinv;      // forward only
    const double *corr_top;  // [nchains*K] or null (forward only)
    const double *corr_bot;  // [nchains*K] or null
    int K;
    double *tipT = nullptr, *tipB = nullptr;   // k_scan_solve (K = 1): also store every chain's first / last solution value here
    double *seam = nullptr;                    // forward launch of a twisted apply: [nchains*K] staging of every chain's last K results
};

// One reduced (interface) system between partition "lo" and the partition below it.
struct IfaceDesc {
    const double *gb;  // K doubles: bottom tip of g of the upper partition (a neighbour rank's, from the exchange buffer),
    const double *gt;  // K doubles: top tip of g of the lower partition     or null: read g + gb_off / g + gt_off
    int64_t gb_off;    // local partitions: the tips are read in place from the swept vector g (no gather kernel)
    int64_t gt_off;
    const double *WT;  // K*K, column-major W^(t)   (WT[c*K+a] = W[a][c])
    const double *ST;  // K*K, column-major (I - W V)^{-1}
    const double *VT;  // K*K, column-major V^(b)
    const double *BT;  // K*K, column-major B (coupling block of the upper partition) or null
    const double *CT;  // K*K, column-major C (coupling block of the lower partition) or null
    double *corr_bot;  // K doubles: B * x^(t)   (for the upper partition) or null
    double *corr_top;  // K doubles: C * x^(b)   (for the lower partition) or null
    double *xb_out;    // K doubles: x^(b), bottom-tip solution of the upper partition, or null
    double *xt_out;    // K doubles: x^(t), top-tip solution of the lower partition, or null
};

constexpr int NSCAN_ROWS_PER_BLOCK = 256;   // k_nscan_*: 64 lanes x 4 rows
constexpr int DEFAULT_SCAN_KMAX = 3, DEFAULT_SCAN_ROWS = 4;   // (handle options narrow_scan_kmax / narrow_scan_rows)
bool pick_cfg(int K, SweepCfg *cfg, int scan_kmax = DEFAULT_SCAN_KMAX, int scan_rows = DEFAULT_SCAN_ROWS);
bool sweep_shape_exists(const SweepCfg &cfg, int dpw, int nw, int pf);

// launchers (spike_kernels.hip)
hipError_t launch_sweep(const SweepCfg &cfg, bool rev, int ngroups, const SweepArgs &a, hipStream_t st, int tag = 0);
// setup: sweep_multi_nr(cfg) right-hand sides per launch (R = 64 configurations), vector q at in/out + q*ldr.
// Round 1 (32 diagonals per wave): 1 rhs 145 us, 2 rhs 137 us, 3 rhs 300 us, 4 rhs 380 us per launch at N = 4M, K = 128 on
// partial chains of 26 row blocks -- beyond two vectors the kernel needed > 256 VGPRs.  Round 2: 16 diagonals per wave
// (the tile layout does not care), 167 VGPRs with FOUR vectors.
int sweep_multi_nr(const SweepCfg &cfg);   // right-hand sides per launch of launch_sweep_multi for this configuration
hipError_t launch_sweep_multi(const SweepCfg &cfg, bool rev, int nchains, const SweepArgs &a, int64_t ldr, hipStream_t st);
// K > 32: the LU scratch of launch_factor / launch_pack is the BLOCK-BAND layout (dense 16 x 16 tiles, see spike_kernels.hip)
// made by launch_band_to_blocks from the diagonal-major band; lu_blocks_doubles = its size (0: diagonal-major scratch)
size_t lu_blocks_doubles(int64_t n, int K);
// moff / mdir (per 64-row block, or null): the twisted factorisation's row / diagonal mirror, applied while copying
// copy != nullptr: the band entries are also written to copy (diagonal-major, row stride ldc) as they are read, with the slots
// whose column lies outside [0, n_global) zeroed (in the copy and in the tiles): the library's kept copy of a caller's device band
hipError_t launch_band_to_blocks(int64_t n, int K, const double *band, int64_t ld, double *T, hipStream_t st,
                                 const int64_t *moff = nullptr, const int *mdir = nullptr, double *copy = nullptr, int64_t ldc = 0,
                                 int64_t n_global = 0, int64_t row0 = 0);
hipError_t launch_factor(double *lu, int64_t ld, int K, const ChainDesc *chains, int nchains, double boost,
                         unsigned long long *nboost, hipStream_t st);
hipError_t launch_pack(const SweepCfg &cfg, const double *lu, int64_t ld, int K, const ChainDesc *chains,
                       const GroupDesc *groups, int nchains, int64_t total_blocks, const int64_t *blk_prefix,
                       double *Lt, double *Ut, double *dinv, hipStream_t st);
// tridiagonal path: LU band -> l[i] = L[i,i-1], c[i] = U[i,i+1]/U[i,i], dinv[i] = 1/U[i,i]; and the scan sweeps
hipError_t launch_pack_scan(const double *lu, int64_t ld, const ChainDesc *chains, int nchains, double *l, double *c,
                            double *dinv, hipStream_t st);
hipError_t launch_scan_sweep(bool rev, int nchains, const SweepArgs &a, hipStream_t st, int tag = 0);
// both sweeps in one launch, the intermediate vector in registers (chains of at most 4096 rows); a.tiles = l, cu = c
hipError_t launch_scan_solve(int nchains, int max_rows, const SweepArgs &a, const double *cu, hipStream_t st, int tag = 0);
// K = 1..3, four rows per lane (k_nscan_*): coefficient arrays [K][lds] (lds a multiple of 4), see launch_pack_nscan
int nscan_max_rows(int K);
hipError_t launch_nscan_solve(int K, int nchains, int max_rows, const SweepArgs &a, const double *cu, int64_t lds, hipStream_t st, int tag = 0);
hipError_t launch_nscan_sweep(int K, bool rev, int nchains, const SweepArgs &a, const double *coef, int64_t lds, hipStream_t st, int tag = 0);
hipError_t launch_pack_nscan(const double *lu, int64_t ld, int K, const ChainDesc *chains, int nchains, double *l, double *c,
                             int64_t lds, double *dinv, hipStream_t st);
hipError_t launch_absmax_diag(const double *band, int64_t ld, int K, int64_t n, double *out, hipStream_t st);
hipError_t launch_gen_band(int64_t N, int K, uint64_t seed, double delta, int64_t row0, int64_t nrows, double *band,
                           int64_t ld, hipStream_t st);
hipError_t launch_band_matvec(int64_t n_global, int64_t row0, int64_t n, int K, const double *band, int64_t ld,
                              const double *xh, double *y, hipStream_t st);
hipError_t launch_band_to_tiles(int64_t n, int K, const double *band, int64_t ld, double *At, hipStream_t st);
hipError_t launch_band_matvec_tiled(int64_t n, int K, const double *At, const double *xh, double *y, hipStream_t st);
// tips: rhs[row0+a] = block(a,b) for every chain (which: 0 = C at top rows of chains with has_top,
// 1 = B at bottom rows of chains with has_bot); gather copies K rows of sol into column b of out.
hipError_t launch_tip_rhs(const double *band, int64_t ld, int K, const ChainDesc *chains, int nchains, int which, int col,
                          double *rhs, hipStream_t st, int ncols = 1, int64_t ldr = 0);
hipError_t launch_tip_gather(const double *sol, int K, const ChainDesc *chains, int nchains, int which, int col,
                             double *out, hipStream_t st, int ncols = 1, int64_t ldr = 0);
// coupling blocks B (which=1) / C (which=0) of every chain as dense column-major K x K
hipError_t launch_coupling_blocks(const double *band, int64_t ld, int K, const ChainDesc *chains, int nchains, int which,
                                  double *out, hipStream_t st);
// S = I - W V, inverse by Gauss-Jordan with partial pivoting; W,V row-major K x K per interface in;
// outputs column-major WT, VT, ST.  flag[i] != 0 when interface i is singular.
// doubles of work area launch_iface_setup wants for nif systems (128 < K <= 256: the 2 x 2 block scheme keeps S, S^-1, the
// verification product and the half-size pieces there)
size_t iface_setup_work_doubles(int K, int nif);
hipError_t launch_iface_setup(int K, int nif, const double *W, const double *V, double *WT, double *VT, double *ST,
                              double *work, int *flag, hipStream_t st);
hipError_t launch_iface_apply(int K, int nif, const IfaceDesc *ifs, const double *g, hipStream_t st);
// one-stage form: desc.WT = MT (2K x 2K, see spike_kernels.hip); the setup pieces that build it from WT, VT, ST
hipError_t launch_iface_apply_m(int K, int nif, const IfaceDesc *ifs, const double *g, hipStream_t st);
hipError_t launch_gemm_kk(int K, int count, const double *A, int64_t sa, const double *B, int64_t sb, double *C, int64_t sc, hipStream_t st);
hipError_t launch_build_iface_m(int K, int nif, const double *ST, const double *P1T, const double *P2T, const double *P3T, double *MT,
                                hipStream_t st);
// stored (decayed) spikes: gather m rows of a spike column, measure what lies outside the window, and the
// second "pass" of the coupled variant as a dense correction  x -= W x_b(prev) (top m rows), x -= V x_t(next)
hipError_t launch_spike_gather(const double *sol, int K, int m, const ChainDesc *chains, int nchains, int which, int col,
                               double *out, double *absmax_in, double *absmax_out, hipStream_t st, int ncols = 1,
                               int64_t ldr = 0);
hipError_t launch_spike_extent(const double *sol, const ChainDesc *chains, int nchains, int which, double tol_abs,
                               int *extent, hipStream_t st);
// tips_ready: the chain-end values are already in `tips` (the fused tridiagonal solve stores them): one launch instead of two
hipError_t launch_couple_small(int nchains, int K, int m, const ChainDesc *chains, double *tips, const double *WT, const double *ST,
                               const double *VT, const double *Wf, const double *Vf, double *y, hipStream_t st, bool tips_ready = false);
// twisted: every chain has ONE window, at its chain-local top (Wf); a vdir = -1 chain takes the next partition's x_t, reversed
// m1 < m: the window's m1 rows next to the interface in fp64 (Wf / Vf: K x m1 per chain), the other m - m1 rows in fp32
// (Wf32 / Vf32: K x (m - m1)); launch_spike_split makes the two parts from a full fp64 window (spike_kernels.hip)
hipError_t launch_spike_correct(int K, int m, const ChainDesc *chains, int nchains, const double *Wf, const double *Vf,
                                const double *xb, const double *xt, double *x, hipStream_t st, int mode = 0, bool twisted = false,
                                int m1 = 0, const float *Wf32 = nullptr, const float *Vf32 = nullptr, int nthreads = 0);
hipError_t launch_spike_split(int K, int m, int m1, int nchains, int near_end, const double *full, double *p64, float *p32,
                              double *max32, hipStream_t st);

// twisted factorisation (setup, K <= 32): dst = the diagonal-major LU scratch = the band in factor space (the rows of a
// vdir = -1 chain reversed, its diagonals mirrored); wider bands mirror inside launch_band_to_blocks
hipError_t launch_band_flip(const double *src, int64_t lds, int K, const ChainDesc *chains, int nchains, int max_rows,
                            double *dst, int64_t ldd, hipStream_t st);
// K x K matrices of every chain's SEAM end (its last K rows), row-major per chain: Tb = (D^-1 L^-1)_bb B (the forward-swept
// bottom coupling block), Gb = (D^-1 U)_bb^-1.  launch_seam_small: K <= 32, from the diagonal-major LU scratch;
// wider bands get them from launch_spike_trsm (Tb / Gb arguments).
hipError_t launch_seam_small(const double *lu, int64_t ld, int K, const double *band, int64_t ldb, const ChainDesc *chains,
                             int nchains, double *Tb, double *Gb, hipStream_t st);
// per pair (chains 2t, 2t+1): W[t] = Tb[2t] J Gb[2t+1], V[t] = Tb[2t+1] J Gb[2t]  (J = reversal; row-major K x K)
hipError_t launch_seam_products(int K, int npairs, const double *Tb, const double *Gb, double *W, double *V, hipStream_t st);
// out[i] = J in[first + i stride] J for i < count (K x K row-major matrices; J = reversal of rows / columns)
hipError_t launch_flip_kk(int K, int count, const double *in, int first, int stride, double *out, hipStream_t st);

// spike columns by the blocked banded TRSM on MFMA (32 < K <= 128, chain lengths multiples of 16): reads the block-band LU
// scratch `lu` (after launch_factor), fills the tips Wt / Vb, the stored spikes Wf / Vf (m rows, may be null with m = 0) and
// the two extrema the caller checks the decay with; region = rows solved next to every interface (multiple of 64)
size_t spike_trsm_scratch_doubles(int K, int nchains, int region);
// Tb / Gb != nullptr (twisted factorisation): side 1 stops after its forward sweep and delivers Tb instead of Vb / Vf, a
// third side solves (D^-1 U) X = [0; I] on the last K rows and delivers Gb (dinv = 1 / U_ii in factor space)
hipError_t launch_spike_trsm(double *lu, int K, int m, int region, const ChainDesc *chains, int nchains, const double *band,
                             int64_t ld, double *Wt, double *Vb, double *Wf, double *Vf,
                             double *zscratch, double *absmax_in, double *absmax_edge, hipStream_t st,
                             double *Tb = nullptr, double *Gb = nullptr, const double *dinv = nullptr);
hipError_t launch_read_bw(const double *src, int64_t ndoubles, double *sink, hipStream_t st);

// Krylov pieces (spike_krylov.hip)
hipError_t launch_csr_matvec(int64_t n, const int64_t *ia, const int32_t *ja, const double *a, int tpr, const double *x,
                             double *y, hipStream_t st);
// ja = column - (first row of the rank), clamped into int32 (spike_setup_csr_dist); last value wins for a repeated pair
hipError_t launch_csr_to_band(int64_t n, const int64_t *ia, const int32_t *ja, const double *a, int K, double *band,
                              int64_t ld, hipStream_t st);
// deterministic grid reductions: ws = red_workspace_doubles() doubles of per-workgroup partials
size_t red_workspace_doubles();
hipError_t launch_dots(const double *V, int64_t ldv, int nvec, const double *w, int64_t n, double *out /*nvec*/,
                       double *ws, hipStream_t st);
hipError_t launch_axpys(const double *V, int64_t ldv, int nvec, const double *coef, double *w, int64_t n, double sign,
                        hipStream_t st);
// w += sign * sum coef_i V_i and norm2_out[0] = |w|^2 of the result, in the same pass
hipError_t launch_axpys_norm(const double *V, int64_t ldv, int nvec, const double *coef, double *w, int64_t n, double sign,
                             double *norm2_out, double *ws, hipStream_t st);
hipError_t launch_scale_copy(const double *w, const double *scal_dev, int invert, double *out, int64_t n,
                             hipStream_t st);
hipError_t launch_scale_value(double *w, double s, int64_t n, hipStream_t st, const double *norm2_dev = nullptr);
hipError_t launch_residual(const double *b, const double *ax, double *r, int64_t n, hipStream_t st);
hipError_t launch_lincomb(const double *V, int64_t ldv, int nvec, const double *y_dev, double *x, int64_t n,
                          hipStream_t st);

}  // namespace spike
