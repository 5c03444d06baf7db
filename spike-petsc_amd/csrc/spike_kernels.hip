// spike_kernels.hip -- hand-written gfx950 kernels of the SPIKE banded preconditioner.
//
// Reference slot filled: the inner PC of PCBANDED, /root/reference/src/matbanded.c:176-178
// (PCSetUp(b->pc): k_factor, k_pack, tips, k_iface_setup) and :190 (PCApply(b->pc,x,y):
// k_sweep forward/backward, k_iface_apply).  The reference holds no device code at all;
// everything here is new (DESIGN.md gives the data layout and the roofline of each kernel).
//
// Design notes for CDNA4:
//  * wave = 64 lanes.  In the sweeps a lane owns ONE ROW of a row block, a wave owns DPW
//    diagonals of the band, NW waves make up the KP = DPW*NW streamed diagonals of a chain.
//  * the packed factors ("tiles") are stored in exactly the order the lanes consume them:
//    every tile load is one 16-byte-per-lane, 1-KiB-per-wave contiguous request, and a chain
//    streams one contiguous region of HBM front to back.  Each factor byte is read once.
//  * the triangular recurrence inside a row block is removed at pack time: the R x R diagonal
//    blocks of L and of D^{-1}U are stored INVERTED, so a block step is two dependent
//    matrix-vector products (far part, near part) with no serial substitution.
//  * the right-hand-side window lives in LDS (2-3 KiB); band data never touches LDS: it is
//    used once, straight from VGPRs, with the next step's tile already in flight.
#include "spike_internal.h"

#include <cstdlib>
#include <vector>

namespace spike {

typedef double d2 __attribute__((ext_vector_type(2)));

// Lanes of ONE wave exchange values through LDS without a workgroup barrier (the hardware executes a wave's DS
// instructions in order).  The compiler, however, reasons per thread and may sink a store below loads it proves not to
// alias for that thread; this fence pins the program order of the LDS accesses at wavefront scope (no instruction).
#define WAVE_LDS_FENCE()                                          \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
    } while (0)

// ------------------------------------------------------------------------------------------
// configuration
// ------------------------------------------------------------------------------------------
bool pick_cfg(int K, SweepCfg *cfg, int scan_kmax, int scan_rows)
{
    if (K < 0 || K > 512) return false;
    if (scan_kmax > 3) scan_kmax = 3;
    // 256 < K <= 512 (round 3): supported, not tuned.  64 diagonals per wave, 5..8 waves per chain, one bundle in flight (the
    // bundle alone is 128 VGPRs); setup takes the generic paths (diagonal-major LU scratch, scalar right-looking
    // factorisation, spike columns one at a time through the sweep kernels, no twisting)
    if (K > 256) { *cfg = {64, 64, (K + 63) / 64}; return true; }
    // K <= scan_kmax: wavefront scan (no tiles); nscan = the four-rows-per-lane kernels (k_nscan_*: always for K = 2, 3)
    if (K >= 1 && K <= scan_kmax) { *cfg = {64, 2, 1}; cfg->scan = true; cfg->nscan = K > 1 || scan_rows == 4; }
    // 16 chains per wave; tridiagonal and pentadiagonal systems stream 4 diagonals, not 8.  (Round 3 measured a 2-row /
    // 2-diagonal configuration for K = 2 -- tiles that hold exactly the band, 9 instead of 13 doubles per row and pass, 32
    // chains per wave: 0.216-0.220 ms per apply at N = 8M against 0.179 for this one, gpurun_out/r3/k2_ab.log: twice the block
    // steps per row cost more than the bytes saved.  Removed again.)
    else if (K <= 4) *cfg = {4, 4, 1};
    else if (K <= 8) *cfg = {8, 8, 1};
    else if (K <= 16) *cfg = {16, 16, 1};
    else if (K <= 32) *cfg = {32, 32, 1};
    else { int nw = (K + 31) / 32; if (nw == 5) nw = 6; if (nw == 7) nw = 8; *cfg = {64, 32, nw}; }
    return true;
}

__host__ __device__ inline constexpr int next_pow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// element index (in doubles) of entry (lane, d) inside one tile; d in [1, KP]
__device__ __forceinline__ int64_t tile_elem(int DPW, int lane, int d)
{
    const int w = (d - 1) / DPW, dd = (d - 1) % DPW;
    return ((int64_t)(w * (DPW >> 1) + (dd >> 1)) * 64 + lane) * 2 + (dd & 1);
}

// ------------------------------------------------------------------------------------------
// k_sweep: one triangular sweep (forward with L, or backward with D^{-1}U) of every chain.
//   forward : out = D^{-1} L^{-1} (in - corrections)
//   backward: out = (D^{-1}U)^{-1} in
// Bound: HBM.  Algorithmic bytes per row: KP*8 (tile) + 8 (in) + 8 (out) [+8 dinv forward].
// ------------------------------------------------------------------------------------------
// TAG only names the instantiation: 0 = the sweeps of PCApply, 1 = the spike solves of setup (sub-ranges of the chains),
// so that a kernel trace keeps the two populations apart.
//
// Prefetch: a step's inputs -- its tile slice (NLD 16-byte loads per lane), its right-hand-side entry and (forward) its
// 1/diag entry -- form one BUNDLE; PF bundles are in flight per wave (registers, statically named: the loop is unrolled
// PF times).  Everything a step waits for comes from its own bundle, so the wait is "all but the (PF-1) younger
// bundles" (vector-memory operations complete in order) and never drains the queue.  Round 1 loaded the right-hand
// side inside the step, i.e. BEHIND the next tile's prefetch: every step then waited for a whole memory round trip and
// a CU could not pull more than ~30 GB/s -- invisible with one chain per CU on all 256 CUs (HBM-bound), but half speed
// with 128 chains (strong scaling, N/8 rows per GPU).  All loads of the main loop are unconditional (clamped
// addresses): a load under a branch would make the compiler's s_waitcnt counts conservative at the join.
template <int R, int DPW, int NW, int PF, bool REV, int TAG>
__global__ __launch_bounds__(NW * 64) void k_sweep(SweepArgs a)
{
    constexpr int CPW = 64 / R;
    constexpr int KP = DPW * NW;
    constexpr int NLD = DPW / 2;
    constexpr int WS = next_pow2(KP + R);
    constexpr int NWB = ((R - 2) / DPW + 1) < NW ? ((R - 2) / DPW + 1) : NW;  // waves that own in-block entries
    constexpr int64_t TILE2 = (int64_t)NW * NLD * 64;                         // tile size in double2

    __shared__ double W[CPW][2 * WS];   // circular window of finished values, per chain, stored TWICE (at i and
                                        // i + WS) so that a block's reads never wrap: one base address + immediates
    __shared__ double W2[CPW][KP + R];  // [KP zeros][R block-local intermediates]
    __shared__ double red[NW][64];
    __shared__ double red2[NWB][64];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane / R, lr = lane % R;
    const int grp = blockIdx.x;
    const int p = grp * CPW + c;
    const bool valid = p < a.nchains;
    ChainDesc cd;
    cd.row0 = 0; cd.nrows = 0; cd.nsteps = 0; cd.vec0 = 0; cd.vdir = 1; cd.flags = 0;
    if (valid) cd = a.chains[p];
    const GroupDesc gd = a.groups[grp];
    // The caller's vectors live in VECTOR space (chain-local row r at vec0 + vdir r: the bottom half of a twisted partition
    // runs through them backwards -- consecutive lanes still touch consecutive addresses), 1/diag and the intermediate
    // vector between the two launches in FACTOR space (row0 + r): the forward launch reads x mapped and writes factor
    // space, the backward launch reads factor space and writes y mapped.  Ordinary chains: the two spaces coincide.
    const int64_t in0 = REV ? cd.row0 : cd.vec0, out0 = REV ? cd.vec0 : cd.row0;
    const int64_t ind = REV ? 1 : cd.vdir, outd = REV ? cd.vdir : 1;

    for (int t = threadIdx.x; t < CPW * 2 * WS; t += NW * 64) (&W[0][0])[t] = 0.0;
    for (int t = threadIdx.x; t < CPW * (KP + R); t += NW * 64) (&W2[0][0])[t] = 0.0;
    __syncthreads();

    const d2 *tp = reinterpret_cast<const d2 *>(a.tiles) + gd.tile0 * TILE2 + (int64_t)(w * NLD) * 64 + lane;
    const int K = a.K;
    const double *ctop = a.corr_top, *cbot = a.corr_bot;
    const int ns = gd.maxsteps;
    if (ns <= 0) return;

    struct Bundle { d2 t[NLD]; double fv, dv; };
    Bundle bq[PF];
    int pos = 0;

    // local row of this lane in step s (may lie outside the chain: padded last block, shorter chain of the group)
    auto row_of = [&](int s) -> int { return REV ? (cd.nsteps * R - 1 - (s * R + lr)) : (s * R + lr); };

    auto issue = [&](Bundle &q, int s) {
        const int rl = row_of(s);
        const bool act = valid && s < cd.nsteps && rl >= 0 && rl < cd.nrows;
        const int rc = act ? rl : 0;                        // clamped: the load is unconditional, the value selected later
        q.fv = a.in[in0 + ind * rc];
        if (!REV) q.dv = a.dinv[cd.row0 + rc];
        const d2 *src = tp + (int64_t)s * TILE2;
#pragma unroll
        for (int i = 0; i < NLD; ++i) q.t[i] = __builtin_nontemporal_load(src + i * 64);
    };

    auto step = [&](const Bundle &q, int s) {
        const d2(&t)[NLD] = q.t;
        const bool actc = valid && s < cd.nsteps;
        const int rl = row_of(s);
        const bool act = actc && rl < cd.nrows;
        const int64_t gi = out0 + outd * rl;
        double fv = act ? q.fv : 0.0;
        double dv = 1.0;
        if (!REV) {
            if (act) dv = q.dv;
            if (ctop != nullptr && act) {
                if (rl < K) fv -= ctop[(int64_t)p * K + rl];
                if (rl >= cd.nrows - K) fv -= cbot[(int64_t)p * K + (rl - (cd.nrows - K))];
            }
        }
        // ---- phase A: far part (columns of earlier blocks).  The block's own window slots are
        // zero while it runs, so in-block entries (stored in the same tile) contribute nothing.
        const int slot = (pos + lr) & (WS - 1);           // where this lane's value of the current block lives
        double *wr = &W[c][slot];
        wr[0] = 0.0;
        wr[WS] = 0.0;
        WAVE_LDS_FENCE();
        // pos is a multiple of R and WS a multiple of R, so (pos & (WS-1)) + WS + lr - d stays inside [R, 2*WS)
        const double *wp = &W[c][(pos & (WS - 1)) + WS + lr - KP - w * DPW];
        // The lane's DPW window values are requested back to back and only then consumed: left to itself the compiler
        // (short of registers next to the prefetched bundles) issues read - wait - 2 FMA - read ..., i.e. NLD serial LDS
        // round trips of ~64 cycles each per phase, which made the block step, not HBM, the limit of a lone chain.
        // (64 diagonals per wave: in two batches of 32, so that window values + bundle stay inside the register file)
        constexpr int XB = DPW > 32 ? 32 : DPW;
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int hx = 0; hx < DPW / XB; ++hx) {
            double xa[XB];
#pragma unroll
            for (int i = 0; i < XB; ++i) xa[i] = wp[KP - 1 - (hx * XB + i)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < XB / 2; ++i) {
                acc0 = fma(t[hx * (XB / 2) + i].x, xa[2 * i], acc0);       // d = w*DPW + 1 + 2i
                acc1 = fma(t[hx * (XB / 2) + i].y, xa[2 * i + 1], acc1);   // d + 1
            }
        }
        double acc = acc0 + acc1;
        if (NW > 1) {
            red[w][lane] = acc;
            __syncthreads();
            acc = 0.0;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) acc += red[ww][lane];
        }
        const double tt = fv - acc;
        // ---- phase B: in-block part through the inverted diagonal block (entries stored negated)
        W2[c][KP + lr] = tt;
        WAVE_LDS_FENCE();
        double g = tt;
        if (w < NWB) {
            const double *w2p = &W2[c][KP + lr - w * DPW - 1];
            double b0 = 0.0, b1 = 0.0;
#pragma unroll
            for (int hx = 0; hx < DPW / XB; ++hx) {
                double xb[XB];
#pragma unroll
                for (int i = 0; i < XB; ++i) xb[i] = w2p[-(hx * XB + i)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < XB / 2; ++i) {
                    b0 = fma(t[hx * (XB / 2) + i].x, xb[2 * i], b0);       // d0 = w*DPW + 1 + 2i
                    b1 = fma(t[hx * (XB / 2) + i].y, xb[2 * i + 1], b1);
                }
            }
            if (NW > 1) red2[w][lane] = b0 + b1;
            else g = tt - (b0 + b1);
        }
        if (NW > 1) {
            __syncthreads();
            double s2 = 0.0;
#pragma unroll
            for (int ww = 0; ww < NWB; ++ww) s2 += red2[ww][lane];
            g = tt - s2;
        }
        WAVE_LDS_FENCE();  // all in-block reads of this step precede the overwrite of the block's slots
        wr[0] = g;
        wr[WS] = g;
        WAVE_LDS_FENCE();
        if (act && w == 0) {
            a.out[gi] = REV ? g : g * dv;
            // twisted, forward launch: the last K values of every chain ALSO go to a staging array -- the seam kernel reads
            // them there and writes the corrected values into the intermediate vector, so its workgroups never race
            if (!REV && a.seam != nullptr && rl >= cd.nrows - K) a.seam[(int64_t)p * K + (rl - (cd.nrows - K))] = g * dv;
        }
        pos += R;
    };

    // prologue: PF bundles in flight (a chain shorter than PF steps re-requests its last tile: harmless)
#pragma unroll
    for (int q = 0; q < PF; ++q) issue(bq[q], q < ns ? q : ns - 1);
    int s = 0;
    // main loop: every prefetch is in range, no load sits under a branch
    for (; s + 2 * PF <= ns; s += PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            step(bq[q], s + q);
            issue(bq[q], s + q + PF);
        }
    }
    // tail (< 2 PF steps): prefetch only what exists
    for (; s < ns; s += PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            if (s + q < ns) {
                step(bq[q], s + q);
                if (s + q + PF < ns) issue(bq[q], s + q + PF);
            }
        }
    }
}

// ---- sweep shapes ------------------------------------------------------------------------------------------
// The tile layout does not depend on how the KP streamed diagonals are dealt to waves: entry (lane, d) sits at
// ((d-1)/2 * 64 + lane) * 2 + (d-1)%2 whatever DPW is.  So the SAME packed factors can be swept by NW = KP/DPW waves
// for any DPW that divides KP, chosen at launch time:
//   * few chains (fewer workgroups than ~2 per CU): more, lighter waves per chain -- 8 waves x 16 diagonals instead of
//     4 x 32 at K = 128 -- put twice the loads in flight per CU with half the registers per bundle, and two waves per
//     SIMD hide each other's LDS round trips;
//   * many chains: the heavy shape (fewer barriers' worth of waves per byte).
// Prefetch depth PF: (PF - 1) bundles of NLD + 2 loads must stay below the 6-bit vmcnt range (63), and the bundles live
// in registers (NLD*4 + 4 VGPRs each).
template <int R, int DPW, int NW, int PF>
static hipError_t launch_sweep_t(bool rev, int ngroups, const SweepArgs &a, hipStream_t st, int tag)
{
    const dim3 g(ngroups), b(NW * 64);
    if (tag == 0) {
        if (rev) hipLaunchKernelGGL((k_sweep<R, DPW, NW, PF, true, 0>), g, b, 0, st, a);
        else hipLaunchKernelGGL((k_sweep<R, DPW, NW, PF, false, 0>), g, b, 0, st, a);
    } else {
        if (rev) hipLaunchKernelGGL((k_sweep<R, DPW, NW, PF, true, 1>), g, b, 0, st, a);
        else hipLaunchKernelGGL((k_sweep<R, DPW, NW, PF, false, 1>), g, b, 0, st, a);
    }
    return hipGetLastError();
}

// same, PCApply only (TAG 0): the alternative shapes are not used by setup's spike solves
template <int R, int DPW, int NW, int PF>
static hipError_t launch_sweep_a(bool rev, int ngroups, const SweepArgs &a, hipStream_t st)
{
    const dim3 g(ngroups), b(NW * 64);
    if (rev) hipLaunchKernelGGL((k_sweep<R, DPW, NW, PF, true, 0>), g, b, 0, st, a);
    else hipLaunchKernelGGL((k_sweep<R, DPW, NW, PF, false, 0>), g, b, 0, st, a);
    return hipGetLastError();
}

// shape list (R, DPW, NW, PF) beyond the base shapes; X(R, DPW, NW, PF)
#define SPIKE_ALT_SHAPES(X)                                                                          \
    X(32, 8, 4, 4) X(32, 16, 2, 3) X(32, 16, 2, 4) X(32, 16, 2, 6) X(32, 32, 1, 3) X(32, 32, 1, 4) X(32, 8, 4, 8) \
    X(64, 32, 2, 4) X(64, 32, 3, 4) X(64, 32, 4, 4)                                                  \
    X(64, 16, 4, 2) X(64, 16, 4, 4) X(64, 16, 6, 2) X(64, 16, 6, 4) X(64, 16, 8, 2) X(64, 16, 8, 4)  \
    X(64, 16, 12, 2) X(64, 16, 12, 3) X(64, 16, 16, 2) X(64, 16, 16, 3)

bool sweep_shape_exists(const SweepCfg &cfg, int dpw, int nw, int pf)
{
    if (dpw * nw != cfg.DPW * cfg.NW) return false;
    if (dpw == cfg.DPW && nw == cfg.NW && pf == cfg.basePF()) return true;
#define X(R_, D_, N_, P_) if (cfg.R == R_ && dpw == D_ && nw == N_ && pf == P_) return true;
    SPIKE_ALT_SHAPES(X)
#undef X
    return false;
}

hipError_t launch_sweep(const SweepCfg &cfg, bool rev, int ngroups, const SweepArgs &a, hipStream_t st, int tag)
{
    if (ngroups <= 0) return hipSuccess;
    const int dpw = (tag == 0 && cfg.sDPW > 0) ? cfg.sDPW : cfg.DPW, nw = (tag == 0 && cfg.sNW > 0) ? cfg.sNW : cfg.NW;
    const int pf = (tag == 0 && cfg.sPF > 0) ? cfg.sPF : cfg.basePF();
    if (!(dpw == cfg.DPW && nw == cfg.NW && pf == cfg.basePF())) {
#define X(R_, D_, N_, P_) if (cfg.R == R_ && dpw == D_ && nw == N_ && pf == P_) return launch_sweep_a<R_, D_, N_, P_>(rev, ngroups, a, st);
        SPIKE_ALT_SHAPES(X)
#undef X
        return hipErrorInvalidValue;
    }
    // base shapes (what pick_cfg gives; PF = SweepCfg::basePF)
    if (cfg.R == 4) return launch_sweep_t<4, 4, 1, 12>(rev, ngroups, a, st, tag);
    if (cfg.R == 8) return launch_sweep_t<8, 8, 1, 8>(rev, ngroups, a, st, tag);
    if (cfg.R == 16) return launch_sweep_t<16, 16, 1, 4>(rev, ngroups, a, st, tag);
    if (cfg.R == 32) return launch_sweep_t<32, 32, 1, 2>(rev, ngroups, a, st, tag);
    if (cfg.DPW == 64) {   // 256 < K <= 512
        switch (cfg.NW) {
        case 5: return launch_sweep_t<64, 64, 5, 1>(rev, ngroups, a, st, tag);
        case 6: return launch_sweep_t<64, 64, 6, 1>(rev, ngroups, a, st, tag);
        case 7: return launch_sweep_t<64, 64, 7, 1>(rev, ngroups, a, st, tag);
        case 8: return launch_sweep_t<64, 64, 8, 1>(rev, ngroups, a, st, tag);
        }
        return hipErrorInvalidValue;
    }
    switch (cfg.NW) {
    case 2: return launch_sweep_t<64, 32, 2, 2>(rev, ngroups, a, st, tag);
    case 3: return launch_sweep_t<64, 32, 3, 2>(rev, ngroups, a, st, tag);
    case 4: return launch_sweep_t<64, 32, 4, 2>(rev, ngroups, a, st, tag);
    case 6: return launch_sweep_t<64, 32, 6, 2>(rev, ngroups, a, st, tag);
    case 8: return launch_sweep_t<64, 32, 8, 2>(rev, ngroups, a, st, tag);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// k_sweep_multi: the same sweep for NR right-hand sides at once (setup: the 2K spike columns).  A tile is loaded into
// registers ONCE and used for NR vectors, so the spike solves read the factors 2K/NR times instead of 2K times; the
// two workgroup barriers of a block step are shared by the NR vectors.  R = 64, DPW = 32 only (one chain per workgroup);
// right-hand side q lives at in + q*ldr / out + q*ldr.  No corrections (setup never needs them).
// ------------------------------------------------------------------------------------------
template <int DPW, int NW, bool REV, int NR>
__global__ __launch_bounds__(NW * 64) void k_sweep_multi(SweepArgs a, int64_t ldr)
{
    constexpr int R = 64;
    constexpr int KP = DPW * NW;
    constexpr int NLD = DPW / 2;
    constexpr int WS = next_pow2(KP + R);
    constexpr int NWB = ((R - 2) / DPW + 1) < NW ? ((R - 2) / DPW + 1) : NW;   // waves that own in-block entries
    constexpr int64_t TILE2 = (int64_t)NW * NLD * 64;

    __shared__ double W[NR][2 * WS];
    __shared__ double W2[NR][KP + R];
    __shared__ double red[NR][NW][64];
    __shared__ double red2[NR][NWB][64];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x;
    if (p >= a.nchains) return;
    const ChainDesc cd = a.chains[p];
    const GroupDesc gd = a.groups[p];

    for (int t = threadIdx.x; t < NR * 2 * WS; t += NW * 64) (&W[0][0])[t] = 0.0;
    for (int t = threadIdx.x; t < NR * (KP + R); t += NW * 64) (&W2[0][0])[t] = 0.0;
    __syncthreads();

    const d2 *tp = reinterpret_cast<const d2 *>(a.tiles) + gd.tile0 * TILE2 + (int64_t)(w * NLD) * 64 + lane;
    d2 tA[NLD], tB[NLD];
    int pos = 0;

    auto load_tile = [&](d2(&t)[NLD], int s) {
        const d2 *q = tp + (int64_t)s * TILE2;
#pragma unroll
        for (int i = 0; i < NLD; ++i) t[i] = __builtin_nontemporal_load(q + i * 64);
    };

    auto step = [&](const d2(&t)[NLD], int s) {
        const int rl = REV ? (cd.nsteps * R - 1 - (s * R + lane)) : (s * R + lane);
        const bool act = s < cd.nsteps && rl < cd.nrows;
        const int64_t gi = cd.row0 + rl;
        double fv[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) fv[q] = act ? a.in[q * ldr + gi] : 0.0;
        double dv = 1.0;
        if (!REV && act) dv = a.dinv[gi];
        const int slot = (pos + lane) & (WS - 1);
#pragma unroll
        for (int q = 0; q < NR; ++q) { W[q][slot] = 0.0; W[q][slot + WS] = 0.0; }
        WAVE_LDS_FENCE();
        const int base = (pos & (WS - 1)) + WS + lane - KP - w * DPW;
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const double *wp = &W[q][base];
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                acc0 = fma(t[i].x, wp[KP - 1 - 2 * i], acc0);
                acc1 = fma(t[i].y, wp[KP - 2 - 2 * i], acc1);
            }
            red[q][w][lane] = acc0 + acc1;
            __builtin_amdgcn_sched_barrier(0);  // one vector's window values in registers at a time (VGPR budget: 2 waves/SIMD)
        }
        __syncthreads();
        double tt[NR], g[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            double acc = 0.0;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) acc += red[q][ww][lane];
            tt[q] = fv[q] - acc;
            W2[q][KP + lane] = tt[q];
        }
        WAVE_LDS_FENCE();
        if (w < NWB) {
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                double b0 = 0.0, b1 = 0.0;
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int d0 = w * DPW + 1 + 2 * i;
                    b0 = fma(t[i].x, W2[q][KP + lane - d0], b0);
                    b1 = fma(t[i].y, W2[q][KP + lane - d0 - 1], b1);
                }
                red2[q][w][lane] = b0 + b1;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            double s2 = 0.0;
#pragma unroll
            for (int ww = 0; ww < NWB; ++ww) s2 += red2[q][ww][lane];
            g[q] = tt[q] - s2;
        }
        WAVE_LDS_FENCE();
#pragma unroll
        for (int q = 0; q < NR; ++q) { W[q][slot] = g[q]; W[q][slot + WS] = g[q]; }
        WAVE_LDS_FENCE();
        if (act && w == 0) {
#pragma unroll
            for (int q = 0; q < NR; ++q) a.out[q * ldr + gi] = REV ? g[q] : g[q] * dv;
        }
        pos += R;
    };

    const int ns = gd.maxsteps;
    if (ns > 0) load_tile(tA, 0);
    for (int s = 0; s < ns; s += 2) {
        if (s + 1 < ns) load_tile(tB, s + 1);
        step(tA, s);
        if (s + 1 < ns) {
            if (s + 2 < ns) load_tile(tA, s + 2);
            step(tB, s + 1);
        }
    }
}

template <int DPW, int NW, int NR>
static hipError_t launch_sweep_multi_t(bool rev, int nchains, const SweepArgs &a, int64_t ldr, hipStream_t st)
{
    if (rev) hipLaunchKernelGGL((k_sweep_multi<DPW, NW, true, NR>), dim3(nchains), dim3(NW * 64), 0, st, a, ldr);
    else hipLaunchKernelGGL((k_sweep_multi<DPW, NW, false, NR>), dim3(nchains), dim3(NW * 64), 0, st, a, ldr);
    return hipGetLastError();
}

// right-hand sides per launch for this configuration (setup asks before it sizes its buffers).  The tile layout does not
// depend on how the diagonals are dealt to waves, so the batched solves use 16 diagonals per wave: half the tile
// registers per wave leave room for FOUR right-hand sides per pass over the factors (with 32 diagonals per wave a third
// vector already pushed the kernel past 256 VGPRs: round 1).  K > 192 would need 16 such waves at 128 VGPRs each (measured:
// spills, no gain) and keeps two vectors on 32-diagonal waves.
int sweep_multi_nr(const SweepCfg &cfg) { return (cfg.R == 64 && !cfg.scan && cfg.NW <= 6) ? 4 : 2; }

// one chain per workgroup (groups[p] describes chain p): configurations with R = 64 only
hipError_t launch_sweep_multi(const SweepCfg &cfg, bool rev, int nchains, const SweepArgs &a, int64_t ldr, hipStream_t st)
{
    if (nchains <= 0) return hipSuccess;
    if (cfg.R != 64 || cfg.scan) return hipErrorInvalidValue;
    switch (cfg.NW) {
    case 2: return launch_sweep_multi_t<16, 4, 4>(rev, nchains, a, ldr, st);
    case 3: return launch_sweep_multi_t<16, 6, 4>(rev, nchains, a, ldr, st);
    case 4: return launch_sweep_multi_t<16, 8, 4>(rev, nchains, a, ldr, st);
    case 6: return launch_sweep_multi_t<16, 12, 4>(rev, nchains, a, ldr, st);
    case 8: return launch_sweep_multi_t<32, 8, 2>(rev, nchains, a, ldr, st);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// k_scan_sweep: the narrow-band (K = 1) solve as a WAVEFRONT-LEVEL SCAN.  A first-order recurrence
//   forward  g_i = f_i - l_i g_{i-1},   backward  x_i = y_i - c_i x_{i+1}
// is the composition of affine maps  t -> a_i t + b_i; 64 consecutive rows sit on the 64 lanes of a wave, an inclusive
// scan of the maps (6 shuffle steps) resolves the whole segment at once, and one carry links the segments of a chain.
// One wave per chain, eight 64-row segments in flight per iteration.  HBM-bound: 32 B/row forward (l, f, 1/u, y),
// 24 B/row backward (c, y, x) -- against 104 B/row when a tridiagonal system is streamed as 4-diagonal tiles.
// ------------------------------------------------------------------------------------------
// Cross-lane moves of the scan are DPP modifiers on VALU moves (row_shr inside a 16-lane row, row_bcast:15 / row_bcast:31
// across rows -- the GFX9 wave-scan idiom), not LDS-crossbar shuffles: a lane without a source keeps `old`, which is the
// identity map (a = 1, b = 0), so the combine needs no per-lane condition.  With __shfl_up (ds_bpermute) the 24 crossbar
// operations per 64-row segment kept the LDS pipe busier than HBM.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_move(double old, double src)
{
    const int rl = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, ROWMASK, 0xF, false);
    const int rh = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(rh, rl);
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ void affine_scan_step(double &a, double &b)
{
    const double au = dpp_move<CTRL, ROWMASK>(1.0, a), bu = dpp_move<CTRL, ROWMASK>(0.0, b);
    b = fma(a, bu, b);
    a *= au;
}

__device__ __forceinline__ void affine_scan64(double &a, double &b, int /*lane*/)
{
    affine_scan_step<0x111, 0xF>(a, b);  // row_shr:1
    affine_scan_step<0x112, 0xF>(a, b);  // row_shr:2
    affine_scan_step<0x114, 0xF>(a, b);  // row_shr:4
    affine_scan_step<0x118, 0xF>(a, b);  // row_shr:8
    affine_scan_step<0x142, 0xA>(a, b);  // row_bcast:15 into rows 1 and 3
    affine_scan_step<0x143, 0xC>(a, b);  // row_bcast:31 into rows 2 and 3
}

// value of lane 63, in scalar registers (the carry between segments)
__device__ __forceinline__ double last_lane(double v)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

template <bool REV, int TAG>
__global__ __launch_bounds__(256) void k_scan_sweep(SweepArgs s)
{
    constexpr int U = 8;  // segments per iteration
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= s.nchains) return;
    const ChainDesc cd = s.chains[p];
    const double *coef = s.tiles;
    const int nseg = (cd.nrows + 63) / 64;
    double carry = 0.0;
    for (int sg = 0; sg < nseg; sg += U) {
        double a[U], b[U], dv[U];
        int64_t gi[U];
        bool act[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = (sg + u) * 64 + lane;           // position in sweep order
            const int rl = REV ? cd.nrows - 1 - q : q;    // local row
            act[u] = (sg + u < nseg) && q < cd.nrows;
            gi[u] = cd.row0 + rl;
            a[u] = act[u] ? -coef[gi[u]] : 0.0;
            b[u] = act[u] ? s.in[gi[u]] : 0.0;
            dv[u] = 1.0;
            if (!REV) {
                if (act[u]) dv[u] = s.dinv[gi[u]];
                if (s.corr_top != nullptr && act[u]) {   // K = 1: one corrected row at each end of the chain
                    if (rl == 0) b[u] -= s.corr_top[p];
                    if (rl == cd.nrows - 1) b[u] -= s.corr_bot[p];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) affine_scan64(a[u], b[u], lane);  // independent of the carry: overlaps across u
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double g = fma(a[u], carry, b[u]);
            carry = last_lane(g);
            if (act[u]) s.out[gi[u]] = REV ? g : g * dv[u];
        }
    }
}

// k_scan_solve: forward AND backward sweep of a tridiagonal chain in ONE launch.  The forward result of a whole chain
// (up to MAXSEG 64-row segments) stays in the wave's registers -- one value per lane and segment -- and the backward
// recurrence -- the same prefix scan with the lanes of a segment in reverse order -- consumes it from there: the intermediate vector
// never goes to HBM.  Traffic = l, f, 1/u, c in + x out = 40 B/row, exactly the algorithmic bytes of a tridiagonal
// solve (3 band entries + rhs + solution), against 56 B/row for two k_scan_sweep launches.
template <int MAXSEG, int TAG>
__global__ __launch_bounds__(256) void k_scan_solve(SweepArgs s, const double *cu)
{
    constexpr int U = 8;
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= s.nchains) return;
    const ChainDesc cd = s.chains[p];
    const double *lcoef = s.tiles;
    const int nseg = (cd.nrows + 63) / 64;   // host guarantees nseg <= MAXSEG
    double y[MAXSEG];
    double carry = 0.0;
#pragma unroll
    for (int sg = 0; sg < MAXSEG; sg += U) {
        if (sg < nseg) {
            double a[U], b[U], dv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rl = (sg + u) * 64 + lane;
                const bool act = rl < cd.nrows;
                const int64_t gi = cd.row0 + (act ? rl : 0);   // clamped: unconditional loads
                a[u] = -lcoef[gi];
                b[u] = s.in[gi];
                dv[u] = s.dinv[gi];
                if (!act) { a[u] = 0.0; b[u] = 0.0; dv[u] = 1.0; }
                if (s.corr_top != nullptr && act) {
                    if (rl == 0) b[u] -= s.corr_top[p];
                    if (rl == cd.nrows - 1) b[u] -= s.corr_bot[p];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) affine_scan64(a[u], b[u], lane);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double g = fma(a[u], carry, b[u]);
                carry = last_lane(g);
                y[sg + u] = g * dv[u];
            }
        }
    }
    carry = 0.0;
#pragma unroll
    for (int sg = MAXSEG - U; sg >= 0; sg -= U) {
        if (sg < nseg) {
            double a[U], b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rl = (sg + u) * 64 + 63 - lane;        // lanes in reverse row order: the recurrence runs upward
                const bool act = rl < cd.nrows;
                a[u] = -cu[cd.row0 + (act ? rl : 0)];
                b[u] = __shfl(y[sg + u], 63 - lane);             // the forward value of row rl sits in lane 63 - lane
                if (!act) { a[u] = 0.0; b[u] = 0.0; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) affine_scan64(a[u], b[u], lane);
#pragma unroll
            for (int u = U - 1; u >= 0; --u) {
                const double x = fma(a[u], carry, b[u]);
                carry = last_lane(x);
                const int rl = (sg + u) * 64 + 63 - lane;
                if (rl < cd.nrows) {
                    s.out[cd.row0 + rl] = x;
                    // the narrow-band coupling step wants the chain-end values of g BEFORE the corrections touch them
                    if (s.tipT != nullptr) { if (rl == 0) s.tipT[p] = x; if (rl == cd.nrows - 1) s.tipB[p] = x; }
                }
            }
        }
    }
}

// forward + backward in one launch when every chain fits the register-resident form (max_rows <= 64 segments)
hipError_t launch_scan_solve(int nchains, int max_rows, const SweepArgs &a, const double *cu, hipStream_t st, int tag)
{
    if (nchains <= 0) return hipSuccess;
    const dim3 g((nchains + 3) / 4), b(256);
    if (max_rows <= 32 * 64) {
        if (tag == 0) hipLaunchKernelGGL((k_scan_solve<32, 0>), g, b, 0, st, a, cu);
        else hipLaunchKernelGGL((k_scan_solve<32, 1>), g, b, 0, st, a, cu);
    } else if (max_rows <= 64 * 64) {
        if (tag == 0) hipLaunchKernelGGL((k_scan_solve<64, 0>), g, b, 0, st, a, cu);
        else hipLaunchKernelGGL((k_scan_solve<64, 1>), g, b, 0, st, a, cu);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_scan_sweep(bool rev, int nchains, const SweepArgs &a, hipStream_t st, int tag)
{
    if (nchains <= 0) return hipSuccess;
    const dim3 g((nchains + 3) / 4), b(256);
    if (tag == 0) {
        if (rev) hipLaunchKernelGGL((k_scan_sweep<true, 0>), g, b, 0, st, a);
        else hipLaunchKernelGGL((k_scan_sweep<false, 0>), g, b, 0, st, a);
    } else {
        if (rev) hipLaunchKernelGGL((k_scan_sweep<true, 1>), g, b, 0, st, a);
        else hipLaunchKernelGGL((k_scan_sweep<false, 1>), g, b, 0, st, a);
    }
    return hipGetLastError();
}

__global__ void k_pack_scan(const double *lu, int64_t ld, const ChainDesc *chains, double *l, double *c, double *dinv)
{
    const ChainDesc cd = chains[blockIdx.y];
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < cd.nrows; r += gridDim.x * blockDim.x) {
        const int64_t i = cd.row0 + r;
        const double di = 1.0 / lu[ld + i];                    // K = 1: diagonals 0 (sub), 1 (main), 2 (super)
        dinv[i] = di;
        l[i] = (r > 0) ? lu[i] : 0.0;                          // multipliers stop at the chain start
        c[i] = (r < cd.nrows - 1) ? lu[2 * ld + i] * di : 0.0; // and the super-diagonal at its end
    }
}

hipError_t launch_pack_scan(const double *lu, int64_t ld, const ChainDesc *chains, int nchains, double *l, double *c,
                            double *dinv, hipStream_t st)
{
    if (nchains <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack_scan, dim3(8, nchains), dim3(256), 0, st, lu, ld, chains, l, c, dinv);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// k_nscan_*: the wavefront scan for K = 1..3 with FOUR ROWS PER LANE (round 3).
//
// A banded recurrence  v_i = r_i - sum_{d=1..K} a_{i,d} v_{i-d}  carries the state s_i = (v_i, ..., v_{i-K+1}); a run of
// rows is an affine map of that state (K x K matrix + K vector).  A lane owns 4 consecutive rows: it forms their map by
// running the recurrence 1 + K times over its own rows (right-hand side, and one unit state component each) -- serial, but
// in registers and without any cross-lane traffic; ONE 64-lane scan of those maps (6 DPP steps) then resolves 256 rows.
// Against one row per lane (k_scan_solve, K = 1) the cross-lane work per row drops 4x; for K = 2 the scan step moves 6
// doubles and costs 12 multiply-adds, which at one row per lane is ~6x the work of the tridiagonal scan (why round 2 did
// not build it) and at four rows per lane about the same as that kernel's.  The lane's 4 rows are 32 contiguous bytes of
// every array: two 16-byte loads per array, 2 KiB contiguous per wave.
// The forward result stays in registers (4 * MAXIT doubles per lane), the backward recurrence runs on mirrored lanes and
// takes it from there: l_1..l_K, 1/u, rhs in, c_1..c_K, x out = (2K + 3) * 8 bytes per row, the algorithmic traffic of a
// banded solve.  Chains longer than 256 * MAXIT rows: k_nscan_sweep (two launches, intermediate vector in HBM).
// Coefficient arrays: diagonal-major, row stride lds (a multiple of 4 doubles), zero where the neighbour lies outside
// the chain, so a chain needs no special first / last rows.
// ------------------------------------------------------------------------------------------
constexpr int NSCAN_RPL = 4;
constexpr int NSCAN_BLK = 64 * NSCAN_RPL;

template <int KK>
struct ScanMap {
    double A[KK][KK];
    double c[KK];
};

template <int KK, int CTRL, int ROWMASK>
__device__ __forceinline__ void scanmap_step(ScanMap<KK> &m)
{
    ScanMap<KK> u, r;   // u: the map of the lanes before (identity where DPP has no source lane)
#pragma unroll
    for (int f = 0; f < KK; ++f) {
#pragma unroll
        for (int e = 0; e < KK; ++e) u.A[f][e] = dpp_move<CTRL, ROWMASK>(f == e ? 1.0 : 0.0, m.A[f][e]);
        u.c[f] = dpp_move<CTRL, ROWMASK>(0.0, m.c[f]);
    }
#pragma unroll
    for (int f = 0; f < KK; ++f) {
#pragma unroll
        for (int e = 0; e < KK; ++e) {
            double s = m.A[f][0] * u.A[0][e];
#pragma unroll
            for (int g = 1; g < KK; ++g) s = fma(m.A[f][g], u.A[g][e], s);
            r.A[f][e] = s;
        }
        double s = m.c[f];
#pragma unroll
        for (int g = 0; g < KK; ++g) s = fma(m.A[f][g], u.c[g], s);
        r.c[f] = s;
    }
    m = r;
}

template <int KK>
__device__ __forceinline__ void scanmap_scan64(ScanMap<KK> &m)
{
    scanmap_step<KK, 0x111, 0xF>(m);  // row_shr:1
    scanmap_step<KK, 0x112, 0xF>(m);  // row_shr:2
    scanmap_step<KK, 0x114, 0xF>(m);  // row_shr:4
    scanmap_step<KK, 0x118, 0xF>(m);  // row_shr:8
    scanmap_step<KK, 0x142, 0xA>(m);  // row_bcast:15 into rows 1 and 3
    scanmap_step<KK, 0x143, 0xC>(m);  // row_bcast:31 into rows 2 and 3
}

// Operand access: BUFFER loads / stores through a per-(array, block) descriptor whose range is the rows of the block that
// belong to the chain -- the hardware returns zeros beyond it (= "no neighbour": exactly what a row past the chain end must
// contribute) and drops stores, so the kernels have NO per-lane branches around memory operations.  (First version: a
// per-lane `if (4 rows valid) 16-byte loads else guarded 8-byte loads` -- the compiler merged the two paths' registers with
// copies right behind every load, i.e. waited for each load on the spot: 147 x s_waitcnt vmcnt(0) in the K = 2 kernel, one
// load in flight per wave however far ahead the source asked for the next block.)
// WIDE (vectors 32-byte aligned; the library's own arrays always): two 16-byte accesses per lane; the range is rounded up
// to whole 4-row groups, so no access straddles its end (a ragged chain end -- only the last chain of a rank, n not a
// multiple of 4 -- is masked after the load and stored element-wise).  Otherwise four 8-byte accesses, exact range.
typedef unsigned int nscan_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int nscan_u2 __attribute__((ext_vector_type(2)));
constexpr int NSCAN_AUX_NT = 2;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t nscan_rsrc(const double *base, int rows, bool round4)
{
    int r = rows < 0 ? 0 : (rows > NSCAN_BLK ? NSCAN_BLK : rows);
    if (round4) r = (r + 3) & ~3;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, r * 8, 0x00020000);
}

template <bool WIDE, int AUX>
__device__ __forceinline__ void nscan_bload4(__amdgpu_buffer_rsrc_t r, int voff, double (&v)[4])
{
    if (WIDE) {
        const d2 a = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, AUX));
        const d2 b = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16, 0, AUX));
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff + 8 * j, 0, AUX));
    }
}

template <bool WIDE>
__device__ __forceinline__ void nscan_bstore4(__amdgpu_buffer_rsrc_t r, int voff, const double (&v)[4])
{
    if (WIDE) {
        d2 a, b;
        a.x = v[0]; a.y = v[1]; b.x = v[2]; b.y = v[3];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(nscan_u4, a), r, voff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(nscan_u4, b), r, voff + 16, 0, 0);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(nscan_u2, v[j]), r, voff + 8 * j, 0, 0);
    }
}

// One block of 256 positions in sweep order: lane owns positions 4 lane .. 4 lane + 3 with coefficients a[d][j] (of
// v_{j-1-d}) and right-hand sides r[j].  S = state entering the block (wave-uniform: S[e] = v at position -1-e), replaced
// by the state leaving it.  Everything up to the scan is independent of S.
template <int KK>
struct NScanBlock {
    double h[KK][4], p[4];
    ScanMap<KK> m;
    __device__ __forceinline__ void prepare(const double (&a)[KK][4], const double (&r)[4])
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double pj = r[j];
#pragma unroll
            for (int d = 0; d < KK; ++d)
                if (j - 1 - d >= 0) pj = fma(-a[d][j], p[j - 1 - d], pj);
            p[j] = pj;
#pragma unroll
            for (int e = 0; e < KK; ++e) {
                double hj = 0.0;
#pragma unroll
                for (int d = 0; d < KK; ++d) {
                    const int idx = j - 1 - d;
                    if (idx >= 0) hj = fma(-a[d][j], h[e][idx], hj);
                    else if (-1 - idx == e) hj -= a[d][j];
                }
                h[e][j] = hj;
            }
        }
#pragma unroll
        for (int f = 0; f < KK; ++f) {
#pragma unroll
            for (int e = 0; e < KK; ++e) m.A[f][e] = h[e][3 - f];
            m.c[f] = p[3 - f];
        }
        scanmap_scan64<KK>(m);
    }
    __device__ __forceinline__ void finish(double (&S)[KK], double (&v)[4])
    {
        double out[KK], in[KK];
#pragma unroll
        for (int f = 0; f < KK; ++f) {
            double s = m.c[f];
#pragma unroll
            for (int e = 0; e < KK; ++e) s = fma(m.A[f][e], S[e], s);
            out[f] = s;
        }
#pragma unroll
        for (int f = 0; f < KK; ++f) in[f] = dpp_move<0x138, 0xF>(S[f], out[f]);   // wave_shr:1, lane 0 keeps S
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double s = p[j];
#pragma unroll
            for (int e = 0; e < KK; ++e) s = fma(h[e][j], in[e], s);
            v[j] = s;
        }
#pragma unroll
        for (int f = 0; f < KK; ++f) S[f] = last_lane(out[f]);
    }
};

struct NScanArgs {
    const double *lco, *uco, *dinv;   // [K][lds] multipliers, [K][lds] U / diag, [n] 1 / diag
    int64_t lds;
    const ChainDesc *chains;
    int nchains;
    const double *in;
    double *out;
    const double *corr_top, *corr_bot;
    double *tipT, *tipB;
};

// raw operands of one block: forward = multipliers + right-hand side, backward = U / diag + 1 / diag (the forward result is
// scaled by 1 / diag on its way INTO the backward recurrence: both directions then load K + 1 arrays per row)
template <int KK>
struct NScanRaw {
    double a[KK][4], r[4];
};

// wave-uniform view of a chain (the chain index comes from threadIdx: make uniformity provable for the descriptors)
struct NScanChain {
    int64_t row0;
    int nrows, p;
};

__device__ __forceinline__ bool nscan_chain(const NScanArgs &s, NScanChain &c)
{
    c.p = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (c.p >= s.nchains) return false;
    const ChainDesc cd = s.chains[c.p];
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(cd.row0 & 0xffffffffll));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)cd.row0 >> 32));
    c.row0 = (int64_t)(((uint64_t)hi << 32) | lo);
    c.nrows = __builtin_amdgcn_readfirstlane(cd.nrows);
    return true;
}

// The load functions ONLY issue loads (into the ring slot, rows in memory order); everything that reads the loaded values --
// ragged-end masks, boundary corrections, the reversal for the backward direction -- happens in the *_use functions right
// before the block is processed, so that nothing waits for a load before its block's turn.
template <int KK, bool AL>
__device__ __forceinline__ void nscan_load_fwd(NScanRaw<KK> &q, const NScanArgs &s, const NScanChain &c, int it, int lane)
{
    const int left = it < 0 ? 0 : c.nrows - it * NSCAN_BLK;   // (a block past the end: negative -> empty range)
    const int64_t g = c.row0 + (int64_t)it * NSCAN_BLK;
    const int voff = 32 * lane;
#pragma unroll
    for (int d = 0; d < KK; ++d) nscan_bload4<true, NSCAN_AUX_NT>(nscan_rsrc(s.lco + d * s.lds + g, left, true), voff, q.a[d]);
    nscan_bload4<AL, 0>(nscan_rsrc(s.in + g, left, AL), voff, q.r);
}

template <int KK>
__device__ __forceinline__ void nscan_use_fwd(NScanRaw<KK> &q, const NScanArgs &s, const NScanChain &c, int it, int lane)
{
    const int left = c.nrows - it * NSCAN_BLK;
    if (left < NSCAN_BLK && (left & 3)) {   // ragged end (wave-uniform test): the last 4-row group reaches past the chain
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = 4 * lane + j < left;
#pragma unroll
            for (int d = 0; d < KK; ++d) q.a[d][j] = ok ? q.a[d][j] : 0.0;
            q.r[j] = ok ? q.r[j] : 0.0;
        }
    }
    if (s.corr_top != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = it * NSCAN_BLK + 4 * lane + j;
            if (r < KK && r < c.nrows) q.r[j] -= s.corr_top[(int64_t)c.p * KK + r];
            if (r >= c.nrows - KK && r < c.nrows) q.r[j] -= s.corr_bot[(int64_t)c.p * KK + (r - (c.nrows - KK))];
        }
    }
}

// backward: position q of block `it` = local row 256 it + 255 - q; the lane's 4 rows start at 4 (63 - lane) and are taken in
// reverse.  q.r receives 1 / diag of those rows (0 past the chain end: the forward value there is 0 as well).
template <int KK>
__device__ __forceinline__ void nscan_load_bwd(NScanRaw<KK> &q, const NScanArgs &s, const NScanChain &c, int it, int lane)
{
    const int left = it < 0 ? 0 : c.nrows - it * NSCAN_BLK;
    const int64_t g = c.row0 + (int64_t)it * NSCAN_BLK;
    const int voff = 32 * (63 - lane);
#pragma unroll
    for (int d = 0; d < KK; ++d) nscan_bload4<true, NSCAN_AUX_NT>(nscan_rsrc(s.uco + d * s.lds + g, left, true), voff, q.a[d]);
    nscan_bload4<true, NSCAN_AUX_NT>(nscan_rsrc(s.dinv + g, left, true), voff, q.r);
}

template <int KK>
__device__ __forceinline__ void nscan_use_bwd(const NScanRaw<KK> &q, const NScanChain &c, int it, int lane, double (&a)[KK][4], double (&dv)[4])
{
    const int left = c.nrows - it * NSCAN_BLK;
    const bool ragged = left < NSCAN_BLK && (left & 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool ok = !ragged || 4 * (63 - lane) + 3 - j < left;
#pragma unroll
        for (int d = 0; d < KK; ++d) a[d][j] = ok ? q.a[d][3 - j] : 0.0;
        dv[j] = ok ? q.r[3 - j] : 0.0;
    }
}

template <int KK, bool AL>
__device__ __forceinline__ void nscan_store_bwd(const double (&v)[4], const NScanArgs &s, const NScanChain &c, int it, int lane)
{
    const int left = c.nrows - it * NSCAN_BLK;
    const int64_t g = c.row0 + (int64_t)it * NSCAN_BLK;
    const int voff = 32 * (63 - lane);
    double x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = v[3 - k];
    const __amdgpu_buffer_rsrc_t ro = nscan_rsrc(s.out + g, left, false);   // exact range: nothing is written past the chain
    if (AL && !(left < NSCAN_BLK && (left & 3))) nscan_bstore4<true>(ro, voff, x);
    else nscan_bstore4<false>(ro, voff, x);
    if (s.tipT != nullptr) {   // the coupling step wants the chain-end values BEFORE the corrections touch them
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = it * NSCAN_BLK + 4 * (63 - lane) + k;
            if (r < KK && r < c.nrows) s.tipT[(int64_t)c.p * KK + r] = x[k];
            if (r >= c.nrows - KK && r < c.nrows) s.tipB[(int64_t)c.p * KK + (r - (c.nrows - KK))] = x[k];
        }
    }
}

// NS = blocks in flight per wave and direction: a block's operands are requested NS blocks ahead, into the ring slot the
// block NS before it has just been read out of (slot = block index mod NS: compile-time in the unrolled loops).  A block
// is ~800 cycles of work against ~2 us of loaded memory latency and only two or three waves fit a SIMD (the forward
// result alone is 8 * MAXIT registers).
// Every load of the schedule is issued UNCONDITIONALLY -- a block past the chain's last one gets a zero-length descriptor
// and costs an instruction, no traffic.  With a load under `if (block < nblk)` the compiler's wait-count bookkeeping
// has to assume at each join that the load may not have been issued, i.e. that fewer loads are in flight than really
// are, and the wait for block i's operands (loads return in order) degenerates to "wait for everything": the prefetch
// was in the source and not in the machine code (vmcnt(0) before every block).
template <int KK, int MAXIT, bool AL, int TAG, int NS>
__global__ __launch_bounds__(256) void k_nscan_solve(NScanArgs s)
{
    const int lane = threadIdx.x & 63;
    NScanChain c;
    if (!nscan_chain(s, c)) return;
    const int nblk = (c.nrows + NSCAN_BLK - 1) / NSCAN_BLK;   // host guarantees nblk <= MAXIT
    double z[MAXIT][4];
    double S[KK];
#pragma unroll
    for (int e = 0; e < KK; ++e) S[e] = 0.0;
    NScanRaw<KK> q[NS];
#pragma unroll
    for (int it = -NS; it < MAXIT; ++it) {
        if (it >= 0) {
            if (it < nblk) {
                NScanBlock<KK> b;
                nscan_use_fwd<KK>(q[it % NS], s, c, it, lane);
                b.prepare(q[it % NS].a, q[it % NS].r);
                b.finish(S, z[it]);
            }
        }
        if (it + NS < MAXIT) nscan_load_fwd<KK, AL>(q[(it + NS) % NS], s, c, it + NS, lane);
    }
#pragma unroll
    for (int e = 0; e < KK; ++e) S[e] = 0.0;
#pragma unroll
    for (int it = MAXIT - 1 + NS; it >= 0; --it) {
        if (it < MAXIT) {
            if (it < nblk) {
                double a[KK][4], r[4];
                nscan_use_bwd<KK>(q[it % NS], c, it, lane, a, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] *= __shfl(z[it][3 - j], 63 - lane);   // the mirrored lane's row, / diag
                NScanBlock<KK> b;
                b.prepare(a, r);
                double v[4];
                b.finish(S, v);
                nscan_store_bwd<KK, AL>(v, s, c, it, lane);
            }
        }
        if (it - NS >= 0) nscan_load_bwd<KK>(q[it % NS], s, c, it - NS, lane);
    }
}

// one direction per launch, any chain length: forward writes D^{-1} L^{-1} r to `out`, backward solves with the unit upper
// factor (in = the forward result).  Three blocks in flight per wave (ring slots = step mod 3, the loop unrolled by 3 so that the
// slot is a compile-time index); loads unconditional as in k_nscan_solve (a block outside the chain: zero-length descriptor).
template <int KK, bool REV, bool AL, int TAG>
__global__ __launch_bounds__(256) void k_nscan_sweep(NScanArgs s)
{
    constexpr int NSW = 3;
    const int lane = threadIdx.x & 63;
    NScanChain c;
    if (!nscan_chain(s, c)) return;
    const int nblk = (c.nrows + NSCAN_BLK - 1) / NSCAN_BLK;
    double S[KK];
#pragma unroll
    for (int e = 0; e < KK; ++e) S[e] = 0.0;
    NScanRaw<KK> q[NSW];
    double x[NSW][4];   // forward: 1 / diag; backward: the forward result
    // step t handles block t (forward) or nblk - 1 - t (backward)
    auto load = [&](int slot, int t) {
        const int it = REV ? nblk - 1 - t : t;
        const int left = (it < 0 || it >= nblk) ? 0 : c.nrows - it * NSCAN_BLK;
        const int64_t g = c.row0 + (int64_t)it * NSCAN_BLK;
        if (!REV) {
            nscan_load_fwd<KK, AL>(q[slot], s, c, it, lane);
            nscan_bload4<true, NSCAN_AUX_NT>(nscan_rsrc(s.dinv + g, left, true), 32 * lane, x[slot]);
        } else {
            nscan_load_bwd<KK>(q[slot], s, c, it, lane);   // (1 / diag comes along unused: the forward launch has applied it)
            nscan_bload4<AL, 0>(nscan_rsrc(s.in + g, left, AL), 32 * (63 - lane), x[slot]);
        }
    };
#pragma unroll
    for (int j = 0; j < NSW; ++j) load(j, j);
    for (int t0 = 0; t0 < nblk; t0 += NSW) {
#pragma unroll
        for (int j = 0; j < NSW; ++j) {
            const int t = t0 + j;
            if (t < nblk) {
                const int it = REV ? nblk - 1 - t : t;
                const int left = c.nrows - it * NSCAN_BLK;
                NScanBlock<KK> b;
                double v[4];
                if (!REV) {
                    nscan_use_fwd<KK>(q[j], s, c, it, lane);
                    b.prepare(q[j].a, q[j].r);
                    b.finish(S, v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= x[j][e];
                    const __amdgpu_buffer_rsrc_t ro = nscan_rsrc(s.out + c.row0 + (int64_t)it * NSCAN_BLK, left, false);
                    if (AL && !(left < NSCAN_BLK && (left & 3))) nscan_bstore4<true>(ro, 32 * lane, v);
                    else nscan_bstore4<false>(ro, 32 * lane, v);
                } else {
                    double a[KK][4], dv[4], r[4];
                    nscan_use_bwd<KK>(q[j], c, it, lane, a, dv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = (4 * (63 - lane) + 3 - e < left) ? x[j][3 - e] : 0.0;
                    b.prepare(a, r);
                    b.finish(S, v);
                    nscan_store_bwd<KK, AL>(v, s, c, it, lane);
                }
            }
            load(j, t + NSW);
        }
    }
}

template <int KK, bool AL, int TAG>
static hipError_t launch_nscan_solve_t(int nchains, int max_rows, const NScanArgs &a, hipStream_t st)
{
    // NS = 2: measured equal to 3 and 4 blocks in flight (profiles/r3_nscan_depth.log, taken with a since-removed launch knob: K = 2, N = 8M: 0.0965 / 0.0976 / 0.0992 ms per
    // apply; K = 1, N = 16M: 0.1202 / 0.1207 / 0.1197) -- the kernel runs at 93 % of the box's measured read ceiling, the
    // registers are better spent on nothing
    constexpr int NS = 2;
    const dim3 g((nchains + 3) / 4), b(256);
    if (max_rows <= 4 * NSCAN_BLK) hipLaunchKernelGGL((k_nscan_solve<KK, 4, AL, TAG, NS>), g, b, 0, st, a);
    else if (max_rows <= 8 * NSCAN_BLK) hipLaunchKernelGGL((k_nscan_solve<KK, 8, AL, TAG, NS>), g, b, 0, st, a);
    else if (KK < 3 && max_rows <= 16 * NSCAN_BLK) hipLaunchKernelGGL((k_nscan_solve<KK, (KK < 3 ? 16 : 8), AL, TAG, NS>), g, b, 0, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

static NScanArgs nscan_args(const SweepArgs &a, const double *cu, int64_t lds)
{
    NScanArgs s;
    s.lco = a.tiles; s.uco = cu; s.dinv = a.dinv; s.lds = lds; s.chains = a.chains; s.nchains = a.nchains;
    s.in = a.in; s.out = a.out; s.corr_top = a.corr_top; s.corr_bot = a.corr_bot; s.tipT = a.tipT; s.tipB = a.tipB;
    return s;
}

// K = 3: 64 registers of forward result + three blocks of operands + the 3 x 3 maps do not fit; its 16-block instantiation
// kept the result in scratch memory (measured slower than 2048-row chains)
int nscan_max_rows(int K) { return (K < 3 ? 16 : 8) * NSCAN_BLK; }

// forward + backward in one launch (chains of at most nscan_max_rows() rows); a.tiles = multipliers, cu = U / diag
hipError_t launch_nscan_solve(int K, int nchains, int max_rows, const SweepArgs &a, const double *cu, int64_t lds, hipStream_t st, int tag)
{
    if (nchains <= 0) return hipSuccess;
    if (K < 1 || K > 3 || (lds & 3)) return hipErrorInvalidValue;
    const NScanArgs s = nscan_args(a, cu, lds);
    const bool al = (((uintptr_t)a.in | (uintptr_t)a.out) & 31) == 0;
#define NSCAN_GO(KK)                                                                                   \
    (al ? (tag == 0 ? launch_nscan_solve_t<KK, true, 0>(nchains, max_rows, s, st)                      \
                    : launch_nscan_solve_t<KK, true, 1>(nchains, max_rows, s, st))                     \
        : launch_nscan_solve_t<KK, false, 0>(nchains, max_rows, s, st))
    switch (K) {
    case 1: return NSCAN_GO(1);
    case 2: return NSCAN_GO(2);
    default: return NSCAN_GO(3);
    }
#undef NSCAN_GO
}

hipError_t launch_nscan_sweep(int K, bool rev, int nchains, const SweepArgs &a, const double *coef, int64_t lds, hipStream_t st, int tag)
{
    if (nchains <= 0) return hipSuccess;
    if (K < 1 || K > 3 || (lds & 3)) return hipErrorInvalidValue;
    NScanArgs s = nscan_args(a, coef, lds);
    s.lco = coef;
    const bool al = (((uintptr_t)a.in | (uintptr_t)a.out) & 31) == 0;
    const dim3 g((nchains + 3) / 4), b(256);
    (void)tag;
#define NSCAN_SW(KK)                                                                                   \
    do {                                                                                               \
        if (rev) { if (al) hipLaunchKernelGGL((k_nscan_sweep<KK, true, true, 0>), g, b, 0, st, s);     \
                   else hipLaunchKernelGGL((k_nscan_sweep<KK, true, false, 0>), g, b, 0, st, s); }     \
        else { if (al) hipLaunchKernelGGL((k_nscan_sweep<KK, false, true, 0>), g, b, 0, st, s);        \
               else hipLaunchKernelGGL((k_nscan_sweep<KK, false, false, 0>), g, b, 0, st, s); }        \
    } while (0)
    switch (K) {
    case 1: NSCAN_SW(1); break;
    case 2: NSCAN_SW(2); break;
    default: NSCAN_SW(3); break;
    }
#undef NSCAN_SW
    return hipGetLastError();
}

// LU band (diagonal-major) -> the scan's coefficient arrays
__global__ void k_pack_nscan(const double *lu, int64_t ld, int K, const ChainDesc *chains, double *l, double *c, int64_t lds,
                             double *dinv)
{
    const ChainDesc cd = chains[blockIdx.y];
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < cd.nrows; r += gridDim.x * blockDim.x) {
        const int64_t i = cd.row0 + r;
        const double di = 1.0 / lu[(int64_t)K * ld + i];
        dinv[i] = di;
        for (int d = 1; d <= K; ++d) {
            l[(int64_t)(d - 1) * lds + i] = (r - d >= 0) ? lu[(int64_t)(K - d) * ld + i] : 0.0;
            c[(int64_t)(d - 1) * lds + i] = (r + d < cd.nrows) ? lu[(int64_t)(K + d) * ld + i] * di : 0.0;
        }
    }
}

hipError_t launch_pack_nscan(const double *lu, int64_t ld, int K, const ChainDesc *chains, int nchains, double *l, double *c,
                             int64_t lds, double *dinv, hipStream_t st)
{
    if (nchains <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack_nscan, dim3(8, nchains), dim3(256), 0, st, lu, ld, K, chains, l, c, lds, dinv);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// k_factor: banded LU without pivoting (pivot boosting), in place on the diagonal-major band,
// one workgroup per partition, right-looking.  Generic-K first version: the (K x K) active
// window is updated in L2/MALL; the update of one diagonal is a coalesced wave access.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_factor(double *lu, int64_t ld, int K, const ChainDesc *chains, double boost,
                                                unsigned long long *nboost)
{
    extern __shared__ double sh[];
    double *u = sh, *l = sh + K;
    __shared__ double spiv;
    const ChainDesc cd = chains[blockIdx.x];
    const int64_t s = cd.row0, e = s + cd.nrows;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned long long nb = 0;
    for (int64_t i = s; i < e; ++i) {
        const int m = (int)((e - 1 - i) < K ? (e - 1 - i) : K);
        if (tid == 0) {
            double piv = lu[(int64_t)K * ld + i];
            if (fabs(piv) < boost) {
                piv = (piv < 0.0) ? -boost : boost;
                lu[(int64_t)K * ld + i] = piv;
                ++nb;
            }
            spiv = piv;
        }
        for (int t = tid; t < m; t += 256) u[t] = lu[(int64_t)(K + 1 + t) * ld + i];
        __syncthreads();
        const double piv = spiv;
        for (int t = tid; t < m; t += 256) {
            const int64_t ad = (int64_t)(K - 1 - t) * ld + i + 1 + t;
            const double lv = lu[ad] / piv;
            lu[ad] = lv;
            l[t] = lv;
        }
        __syncthreads();
        for (int delta = -(m - 1) + wave; delta <= m - 1; delta += 4) {
            const int lo = delta < 0 ? -delta : 0;
            const int hi = delta > 0 ? m - 1 - delta : m - 1;
            double *row = lu + (int64_t)(K + delta) * ld + i + 1;
            for (int tr = lo + lane; tr <= hi; tr += 64) row[tr] -= l[tr] * u[tr + delta];
        }
        __syncthreads();
    }
    if (tid == 0 && nb) atomicAdd(nboost, nb);
}

// ------------------------------------------------------------------------------------------
// k_factor_mfma: blocked right-looking banded LU (block = 16) for K padded to 16*KB, KB in {1,2,4,8,16}.
// The trailing K x K window of a partition lives in MFMA accumulator registers (KB x KB tiles of 16 x 16, a wave owns
// RPW tile rows); the panels of the current block step go through LDS; the rank-16 update of the window is
// v_mfma_f64_16x16x4_f64 (4 per tile).  The window slides diagonally by one tile per step: tiles are addressed by
// SLOT (block index mod KB), so nothing moves -- the slots of the finished pivot row/column are refilled with the
// tiles that enter the window.  Compute-bound part of setup: 2*N*K^2 flop (1.4e11 at N = 4M, K = 128).
// f64 MFMA lane maps (cdna_hip_programming.md section 3): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// C/D: col = l&15, row = (l>>4) + 4*reg.
// ------------------------------------------------------------------------------------------
// ---- LU scratch layouts --------------------------------------------------------------------------------------------
// The factorisation works on 16 x 16 tiles.  Reading them out of the DIAGONAL-major band is a gather: the 16 entries of
// a tile row lie on 16 different diagonals, N*8 bytes apart -- every lane of a tile load touches its own cache line,
// and a block step of the K = 128 factorisation moves 2 x 4352 such 8-byte accesses (measured round 1: 22 us per block
// step against 1.7 us of MFMA work).  For K > 32 the scratch is therefore the BLOCK-BAND layout: per 16-row block the
// NTL = 2*KB + 1 tiles (rb, rb-KB .. rb+KB) as dense row-major 16 x 16 squares, 2 KiB contiguous each --
//     T[((rbg * NTL) + (cb - rb + KB)) * 256 + row * 16 + col],  rbg = global row block.
// A tile load is then 4 x 512 contiguous bytes per wave, a finished tile goes back as one 2-KiB store, and the packing
// kernel finds the K entries of a row next to each other.  k_band_to_blocks makes it from the kept band in one
// transposing pass (it replaces the plain scratch copy).  ntl = 0 selects the diagonal-major scratch (K <= 32).
struct LuView {
    double *p;
    int64_t ld;    // diagonal-major: row count of a diagonal
    int K, KB, ntl;
    __device__ __forceinline__ int64_t bb(int64_t rbg, int rb_rel_slot, int row, int col) const
    {
        return ((rbg * ntl) + rb_rel_slot) * 256 + row * 16 + col;
    }
};

// element (r, c) of the chain that starts at local row rs (multiple of 64); r, c chain-local, |c - r| <= 16*KB assumed for
// the block-band layout when slot is in range
__device__ __forceinline__ double lu_get(const LuView &v, int64_t rs, int r, int c)
{
    if (v.ntl == 0) {
        const int d = c - r + v.K;
        if (d < 0 || d > 2 * v.K) return 0.0;
        return v.p[(int64_t)d * v.ld + rs + r];
    }
    const int rb = r >> 4, cb = c >> 4, slot = cb - rb + v.KB;
    if (slot < 0 || slot >= v.ntl) return 0.0;
    return v.p[v.bb((rs >> 4) + rb, slot, r & 15, c & 15)];
}
__device__ __forceinline__ void lu_put(const LuView &v, int64_t rs, int r, int c, double val)
{
    if (v.ntl == 0) {
        const int d = c - r + v.K;
        if (d >= 0 && d <= 2 * v.K) v.p[(int64_t)d * v.ld + rs + r] = val;
        return;
    }
    const int rb = r >> 4, cb = c >> 4, slot = cb - rb + v.KB;
    if (slot >= 0 && slot < v.ntl) v.p[v.bb((rs >> 4) + rb, slot, r & 15, c & 15)] = val;
}

// diagonal-major band -> block-band scratch; one workgroup per RB 16-row blocks, the strip transposed through LDS.
// (RB = 2 -- 256-byte reads -- was measured slower: its strip leaves room for two workgroups per CU only)
// moff / mdir (one entry per 64-row block, or null): factor-space row i is the caller's row moff[i / 64] + mdir[i / 64] i, and a
// mirrored chain (mdir = -1) has its diagonals mirrored -- the twisted factorisation's flip happens in this (transposing)
// copy, which reads the band anyway
template <int RB>
__global__ __launch_bounds__(256) void k_band_to_blocks(int64_t n, int K, int KB, const double *band, int64_t ld, double *T,
                                                        const int64_t *moff, const int *mdir, double *copy, int64_t ldc,
                                                        int64_t n_global, int64_t row0)
{
    extern __shared__ double strip[];   // 16 RB x (W + 1), W = 16 * NTL + 16 (RB - 1)
    constexpr int RW = 16 * RB;
    const int NTL = 2 * KB + 1, W = 16 * NTL + 16 * (RB - 1), LDW = W + 1;
    const int64_t rbg = (int64_t)blockIdx.x * RB;
    const int t = threadIdx.x, row = t % RW, sub = t / RW;
    constexpr int NSUB = 256 / RW;
    for (int q = t; q < RW * LDW; q += 256) strip[q] = 0.0;
    __syncthreads();
    const int64_t i = rbg * 16 + row;
    // strip column of entry (row, d): the columns of the strip start at block column (rbg - KB)
    // eight diagonals per thread in flight (unconditional loads from clamped addresses, selected afterwards): left as a plain
    // loop each load waited for the previous LDS store's slot -- 17 memory round trips per workgroup, 5.0 ms at the headline size
    if (i < n) {
        const int nd = 2 * K + 1;
        const bool mir = mdir != nullptr && mdir[i >> 6] < 0;
        const int64_t si = moff != nullptr ? (mir ? moff[i >> 6] - i : moff[i >> 6] + i) : i;
        for (int d0 = sub; d0 < nd; d0 += 8 * NSUB) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + u * NSUB, dc = d < nd ? d : nd - 1;
                v[u] = band[(int64_t)(mir ? nd - 1 - dc : dc) * ld + si];
            }
            if (copy != nullptr) {
                // the library's kept copy of the band is written on the way (this pass reads every band entry exactly once), slots
                // whose column falls outside [0, n_global) zeroed as k_zero_corners does for the plain copy
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int d = d0 + u * NSUB;
                    if (d < nd) {
                        const int ds = mir ? nd - 1 - d : d;
                        const int64_t c = row0 + si + ds - K;
                        if (c < 0 || c >= n_global) v[u] = 0.0;
                        copy[(int64_t)ds * ldc + si] = v[u];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + u * NSUB;
                if (d < nd) strip[row * LDW + row + d - K + 16 * KB] = v[u];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r2 = 0; r2 < RB; ++r2) {
        if ((rbg + r2) * 16 >= n) break;
        double *out = T + (rbg + r2) * NTL * 256;
        // tile j of row block rbg + r2 = block column (rbg + r2) - KB + j = strip columns 16 (r2 + j) ...
#pragma unroll 4
        for (int j = 0; j < NTL; ++j) out[j * 256 + t] = strip[(16 * r2 + (t >> 4)) * LDW + 16 * (r2 + j) + (t & 15)];
    }
}

// window width (in 16 x 16 tiles) of the blocked factorisation kernels = half-width of the block-band scratch
static inline int lu_kb(int K) { return K <= 64 ? 4 : (K <= 128 ? 8 : 16); }

hipError_t launch_band_to_blocks(int64_t n, int K, const double *band, int64_t ld, double *T, hipStream_t st, const int64_t *moff,
                                 const int *mdir, double *copy, int64_t ldc, int64_t n_global, int64_t row0)
{
    if (n <= 0) return hipSuccess;
    const int KB = lu_kb(K), NTL = 2 * KB + 1;
    const int RB = 1;   // measured at the headline size: RB = 2 (256-byte reads, 74 KiB of LDS, two workgroups per CU) 6.9 ms, RB = 1 5.0 ms
    const size_t shm = (size_t)16 * RB * (16 * NTL + 16 * (RB - 1) + 1) * sizeof(double);
    const void *fn = RB == 2 ? reinterpret_cast<const void *>(k_band_to_blocks<2>) : reinterpret_cast<const void *>(k_band_to_blocks<1>);
    if (shm > 65536) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
    }
    const unsigned nwg = (unsigned)(((n + 15) / 16 + RB - 1) / RB);
    if (RB == 2) hipLaunchKernelGGL(k_band_to_blocks<2>, dim3(nwg), dim3(256), shm, st, n, K, KB, band, ld, T, moff, mdir, copy, ldc, n_global, row0);
    else hipLaunchKernelGGL(k_band_to_blocks<1>, dim3(nwg), dim3(256), shm, st, n, K, KB, band, ld, T, moff, mdir, copy, ldc, n_global, row0);
    return hipGetLastError();
}

// doubles of the block-band scratch for n local rows (0 when the diagonal-major scratch is used)
size_t lu_blocks_doubles(int64_t n, int K)
{
    if (K <= 32 || K > 256) return 0;   // (K > 256: the generic paths work on the diagonal-major scratch)
    const int KB = lu_kb(K);
    return (size_t)((n + 15) / 16) * (size_t)(2 * KB + 1) * 256;
}

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int LDT = 17;            // LDS tile row stride (doubles), padded against bank conflicts
constexpr int TS = 16 * LDT;       // LDS tile size

// One block step's panel work, shared by both factor kernels: LU of the 16 x 16 diagonal tile (no pivoting, pivot
// boosting), the two panel solves, and the write-back of the finished block row / column to the LU band.
// Called by all threads after the panel tiles are in LDS and a barrier; returns after the write-back was issued
// (the caller's next barrier orders it).
template <int KB, int NT>
__device__ __forceinline__ void panel_phase(double *Pd, double *Pc, double *Pr, int s, int np, int64_t rs, const LuView &lv,
                                            double boost, unsigned long long &nb, int tid, int lane, int w)
{
    // ---- 16 x 16 diagonal block: LU without pivoting, pivot boosting (one wave, in LDS)
    if (w == 0) {
        const int r = lane >> 2, c0 = (lane & 3) * 4;
        for (int k = 0; k < 16; ++k) {
            double piv = Pd[k * LDT + k];
            const bool real_row = 16 * s + k < np;
            if (real_row && fabs(piv) < boost) {
                piv = (piv < 0.0) ? -boost : boost;
                if (lane == 0) ++nb;
            }
            double l = 0.0;
            if (r > k) l = Pd[r * LDT + k] / piv;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) Pd[k * LDT + k] = piv;
            if (r > k) {
                if ((lane & 3) == 0) Pd[r * LDT + k] = l;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
                    if (c0 + cc > k) Pd[r * LDT + c0 + cc] -= l * Pd[k * LDT + c0 + cc];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    // ---- panels: L21 = A21 U11^{-1} (one thread per row), U12 = L11^{-1} A12 (one thread per column)
    for (int t = tid; t < 2 * 16 * KB; t += NT) {
        const int which = t / (16 * KB), idx = t % (16 * KB), tile = idx >> 4, line = idx & 15;
        double x[16];
        if (which == 0) {
            double *T = Pc + tile * TS + line * LDT;
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = T[c];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                double v = x[c];
#pragma unroll
                for (int k = 0; k < c; ++k) v -= x[k] * Pd[k * LDT + c];
                x[c] = v / Pd[c * LDT + c];
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) T[c] = x[c];
        } else {
            double *T = Pr + tile * TS + line;
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = T[r * LDT];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double v = x[r];
#pragma unroll
                for (int k = 0; k < r; ++k) v -= Pd[r * LDT + k] * x[k];
                x[r] = v;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) T[r * LDT] = x[r];
        }
    }
    __syncthreads();
    // ---- write the finished block row / block column back (diagonal-major LU band)
    for (int t = tid; t < (2 * KB + 1) * 256; t += NT) {
        const int tile = t >> 8, row = (t >> 4) & 15, col = t & 15;
        int rb, cb;
        const double *T;
        if (tile == 0) { rb = s; cb = s; T = Pd; }
        else if (tile <= KB) { rb = s + tile; cb = s; T = Pc + (tile - 1) * TS; }
        else { rb = s; cb = s + tile - KB; T = Pr + (tile - KB - 1) * TS; }
        const int r = 16 * rb + row, c = 16 * cb + col;
        if (r < np && c < np) lu_put(lv, rs, r, c, T[row * LDT + col]);   // block-band: 256 consecutive threads = one 2-KiB tile
    }
}

template <int KB, int NW>
__global__ __launch_bounds__(NW * 64) void k_factor_mfma(LuView lv, const ChainDesc *chains, double boost,
                                                         unsigned long long *nboost)
{
    constexpr int RPW = KB / NW;
    constexpr int NT = NW * 64;
    extern __shared__ double lds[];
    double *Pd = lds, *Pc = lds + TS, *Pr = lds + TS + KB * TS;
    const ChainDesc cd = chains[blockIdx.x];
    const int64_t rs = cd.row0;
    const int np = cd.nrows;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = (np + 15) / 16;
    unsigned long long nb = 0;

    auto ldA = [&](int rb, int cb, int row, int col) -> double {
        const int r = 16 * rb + row, c = 16 * cb + col;  // partition-local
        if (r >= np || c >= np) return (r == c) ? 1.0 : 0.0;  // identity padding past the partition end
        return lu_get(lv, rs, r, c);
    };
    auto load_tile = [&](v4d &t, int rb, int cb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] = ldA(rb, cb, (lane >> 4) + 4 * q, lane & 15);
    };
    auto store_tile = [&](const v4d &t, double *dst) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[((lane >> 4) + 4 * q) * LDT + (lane & 15)] = t[q];
    };

    v4d acc[RPW][KB];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int b = 0; b < KB; ++b) load_tile(acc[rr][b], rr * NW + w, b);

    for (int s = 0; s < nblk; ++s) {
        const int as = s % KB;
        // pivot row / column tiles: registers -> LDS panels; then refill those slots with the entering tiles
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int a = rr * NW + w;
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                if (a == as) {
                    const int J = (b - as + KB) % KB;  // tile (s, s+J)
                    store_tile(acc[rr][b], J == 0 ? Pd : Pr + (J - 1) * TS);
                    const int Jn = (b - as - 1 + KB) % KB + 1;  // new tile (s+KB, s+Jn)
                    load_tile(acc[rr][b], s + KB, s + Jn);
                } else if (b == as) {
                    const int I = (a - as + KB) % KB;  // tile (s+I, s), I in 1..KB-1
                    store_tile(acc[rr][b], Pc + (I - 1) * TS);
                    load_tile(acc[rr][b], s + I, s + KB);  // new tile (s+I, s+KB)
                }
            }
        }
        // the two panel tiles that were not in the window yet: (s+KB, s) and (s, s+KB)
        for (int t = tid; t < 256; t += NT) {
            const int row = t >> 4, col = t & 15;
            Pc[(KB - 1) * TS + row * LDT + col] = ldA(s + KB, s, row, col);
            Pr[(KB - 1) * TS + row * LDT + col] = ldA(s, s + KB, row, col);
        }
        __syncthreads();
        panel_phase<KB, NT>(Pd, Pc, Pr, s, np, rs, lv, boost, nb, tid, lane, w);
        // ---- trailing update: tile(s+I, s+J) -= L21[I] * U12[J]   (4 x v_mfma_f64_16x16x4_f64 per tile)
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int a = rr * NW + w;
            const int I = (a - as - 1 + KB) % KB + 1;
            const double *Lp = Pc + (I - 1) * TS + (lane & 15) * LDT + (lane >> 4);
            double la[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) la[q] = -Lp[4 * q];
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const int J = (b - as - 1 + KB) % KB + 1;
                const double *Up = Pr + (J - 1) * TS + (lane >> 4) * LDT + (lane & 15);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[rr][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[q], Up[4 * q * LDT], acc[rr][b], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (lane == 0 && w == 0 && nb) atomicAdd(nboost, nb);
}

// 16 x 16 LU of a diagonal tile held in LDS (one wave), see the comment at its call site in k_factor_mfma_la
__device__ __forceinline__ void tile_lu_regs(int s, double *Pd, double *rd, int np, double boost, unsigned long long &nb, int lane)
{
    const int j = lane & 15;
    double e[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) e[r] = Pd[r * LDT + j];
    // this wave is the youngest of its SIMD and would lose every issue arbitration against the MFMA streams of the
    // update waves beside it (priority, then age); its chain is what the whole workgroup waits for
    __builtin_amdgcn_s_setprio(3);
    // Instruction count is what bounds this routine (the wave shares its SIMD with two MFMA-streaming update waves, so every
    // VALU slot is contended): per (pivot k, row r) TWO v_readlane (entry A[r][k] of lane k into an SGPR pair) and ONE fma
    //   e[r] -= A[r][k] * (U[k][j] / piv)        with the bracket formed once per pivot, zero in the lanes j <= k,
    // and the multipliers of column k (lane k's entries below the diagonal, untouched by later pivots) scaled by the lane's
    // own reciprocal pivot at the end.  Was: mul + 2 readlane + fma + selects per (k, r): ~1100 instructions, 20.8 k cycles.
    double myrinv = 1.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        double piv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(e[k]), k),
                                      __builtin_amdgcn_readlane(__double2loint(e[k]), k));
        const bool real_row = 16 * s + k < np;
        if (real_row && fabs(piv) < boost) {
            piv = (piv < 0.0) ? -boost : boost;
            if (lane == 0) ++nb;
            if (j == k) e[k] = piv;                                // the pivot's owner keeps the boosted pivot
        }
        double rinv = __builtin_amdgcn_rcp(piv);
        rinv = fma(fma(-piv, rinv, 1.0), rinv, rinv);
        rinv = fma(fma(-piv, rinv, 1.0), rinv, rinv);
        if (lane == 0) rd[k] = rinv;                               // reciprocal pivots for the division-free panel solves
        if (j == k) myrinv = rinv;
        const double ukr = (j > k) ? e[k] * rinv : 0.0;           // U[k][j] / piv (final U[k][j] stays in e[k])
#pragma unroll
        for (int r = k + 1; r < 16; ++r) {
            const double ark = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(e[r]), k),
                                                __builtin_amdgcn_readlane(__double2loint(e[r]), k));
            e[r] = fma(-ark, ukr, e[r]);
        }
    }
#pragma unroll
    for (int r = 1; r < 16; ++r)
        if (r > j) e[r] *= myrinv;                                 // L[r][j] = A[r][j] / U[j][j]
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (lane < 16) Pd[r * LDT + j] = e[r];
}

// one line per thread (the look-ahead kernel at KB = 8 runs at its 168-register cap: two lines per thread spill there)
template <int KB, int NT>
__device__ __forceinline__ void panel_solves1(const double *Pd, double *Pc, double *Pr, const double *rd, int tid)
{
    for (int t = tid; t < 2 * 16 * KB; t += NT) {
        const int which = t / (16 * KB), idx = t % (16 * KB), tile = idx >> 4, line = idx & 15;
        double x[16];
        if (which == 0) {
            double *T = Pc + tile * TS + line * LDT;
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = T[c];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                double v = x[c];
#pragma unroll
                for (int k = 0; k < c; ++k) v -= x[k] * Pd[k * LDT + c];
                x[c] = v * rd[c];
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) T[c] = x[c];
        } else {
            double *T = Pr + tile * TS + line;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) x[rr] = T[rr * LDT];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                double v = x[rr];
#pragma unroll
                for (int k = 0; k < rr; ++k) v -= Pd[rr * LDT + k] * x[k];
                x[rr] = v;
            }
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) T[rr * LDT] = x[rr];
        }
    }
}

// panel solves of one block step, division-free (rd = reciprocal pivots left by tile_lu_regs):
// L21 = A21 U11^{-1} (a thread per pair of rows), U12 = L11^{-1} A12 (a thread per pair of columns).
// TWO lines per thread: every entry of the diagonal tile read from LDS (a broadcast read: all lanes, one address) serves two
// multiply-adds -- with one line per thread the 136 broadcast reads per line made the LDS pipeline, not the arithmetic, the
// limit of this phase (4.5 k of the 21.7 k cycles of a block step at K = 128).
template <int KB, int NT>
__device__ __forceinline__ void panel_solves2(const double *Pd, double *Pc, double *Pr, const double *rd, int tid)
{
    for (int t = tid; t < 16 * KB; t += NT) {          // 2 x (8 KB line pairs)
        const int which = t / (8 * KB), idx = t % (8 * KB), tile = idx >> 3, line = 2 * (idx & 7);
        double x[16], y[16];
        if (which == 0) {
            double *T = Pc + tile * TS + line * LDT;
#pragma unroll
            for (int c = 0; c < 16; ++c) { x[c] = T[c]; y[c] = T[LDT + c]; }
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                double v = x[c], u = y[c];
#pragma unroll
                for (int k = 0; k < c; ++k) { const double p = Pd[k * LDT + c]; v -= x[k] * p; u -= y[k] * p; }
                const double r = rd[c];
                x[c] = v * r; y[c] = u * r;
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) { T[c] = x[c]; T[LDT + c] = y[c]; }
        } else {
            double *T = Pr + tile * TS + line;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) { x[rr] = T[rr * LDT]; y[rr] = T[rr * LDT + 1]; }
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                double v = x[rr], u = y[rr];
#pragma unroll
                for (int k = 0; k < rr; ++k) { const double p = Pd[rr * LDT + k]; v -= p * x[k]; u -= p * y[k]; }
                x[rr] = v; y[rr] = u;
            }
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) { T[rr * LDT] = x[rr]; T[rr * LDT + 1] = y[rr]; }
        }
    }
}

template <int KB, int NT, bool TWO = true>
__device__ __forceinline__ void panel_solves(const double *Pd, double *Pc, double *Pr, const double *rd, int tid)
{
    if (TWO) panel_solves2<KB, NT>(Pd, Pc, Pr, rd, tid);
    else panel_solves1<KB, NT>(Pd, Pc, Pr, rd, tid);
}

// ------------------------------------------------------------------------------------------
// k_factor_mfma_la: the same blocked LU with a LOOK-AHEAD schedule (round 2).  k_factor_mfma runs a block step as
// [extract panel] | [16x16 LU, one wave] | [panel solves] | [write back] | [trailing update] with everybody waiting at every
// bar: 21.6 us per step at K = 128 against 1.7 us of MFMA work (PMC: 67 % of the wave cycles parked at barriers/waits).
// Here NW update waves own the window (as before) and one more wave owns the panel factorisation; a step is three phases:
//   B(s)   panel wave: LU of the diagonal tile of step s      || update waves: the REST of the trailing update of step s-1
//   C(s)   everybody: the panel solves of step s (division-free: the LU leaves the reciprocal pivots), then the write-back
//   A(s+1) update waves: the update of step s restricted to the NEXT panel (block row / column s+1), which then moves from
//          the accumulators to the other LDS panel buffer; the slots it frees are refilled with the entering tiles
// so the serial chain per step is  next-panel update -> 16 pivots -> panel solves, and the bulk of the MFMA work (and all
// global loads: the entering tiles, and the two panel tiles that were never in the window, fetched a step ahead) hides
// behind the pivots.  Panels are double-buffered in LDS.
// ------------------------------------------------------------------------------------------
template <int KB, int NW>
__global__ __launch_bounds__((NW + 1) * 64) void k_factor_mfma_la(LuView lv, const ChainDesc *chains, double boost,
                                                                  unsigned long long *nboost, unsigned long long *stamps)
{
    constexpr int RPW = KB / NW;
    constexpr int NT = (NW + 1) * 64;
    constexpr int PSZ = (2 * KB + 1) * TS;
    constexpr int NTU = NW * 64;              // threads of the update waves
    constexpr int NX = (512 + NTU - 1) / NTU; // elements of the two prefetched panel tiles per update thread
    extern __shared__ double lds[];
    double *rdiag = lds + 2 * PSZ;            // [2][16] reciprocal pivots of the panel in flight
    const ChainDesc cd = chains[blockIdx.x];
    const int64_t rs = cd.row0;
    const int np = cd.nrows;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool upd = w < NW;                  // wave NW factors the panels
    const int nblk = (np + 15) / 16;
    unsigned long long nb = 0;
    // diagnostic build of the schedule (stamps != nullptr, SPIKE_FACTOR_STAMPS=1): shader-clock stamps of workgroup 0 at the
    // phase boundaries of steps 64..71, slot = (step - 64) * 16 + wave_kind * 8 + point; never set in normal runs
    auto stamp = [&](int s, int point) __attribute__((always_inline)) {
        if (stamps != nullptr && blockIdx.x == 0 && lane == 0 && (w == 0 || w == NW) && s >= 64 && s < 72)
            stamps[(s - 64) * 16 + (w == NW ? 8 : 0) + point] = __builtin_amdgcn_s_memtime();
    };

    auto ldA = [&](int rb, int cb, int row, int col) __attribute__((always_inline)) -> double {
        const int r = 16 * rb + row, c = 16 * cb + col;  // partition-local
        if (r >= np || c >= np) return (r == c) ? 1.0 : 0.0;  // identity padding past the partition end
        return lu_get(lv, rs, r, c);
    };
    // a tile that lies inside the chain and inside the scratch is 2 KiB contiguous, and (row = (lane>>4) + 4q, col = lane&15)
    // is element lane + 64 q of it: four fully coalesced 512-byte loads, one scalar base address per tile
    const int64_t rbg0 = rs >> 4;
    auto tile_inside = [&](int rb, int cb) __attribute__((always_inline)) -> bool {
        return 16 * (rb + 1) <= np && 16 * (cb + 1) <= np && cb - rb + lv.KB >= 0 && cb - rb + lv.KB < lv.ntl;
    };
    auto tile_base = [&](int rb, int cb) __attribute__((always_inline)) -> double * { return lv.p + ((rbg0 + rb) * lv.ntl + (cb - rb + lv.KB)) * 256; };
    auto load_tile = [&](v4d &t, int rb, int cb, auto fc) __attribute__((always_inline)) {
        if (decltype(fc)::value || tile_inside(rb, cb)) {
            const double *tp = tile_base(rb, cb);   // wave-uniform base (scalar registers) + the lane as a 32-bit index: a
                                                    // per-lane 64-bit pointer per tile was what the kernel spilled
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = tp[64 * q + lane];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = ldA(rb, cb, (lane >> 4) + 4 * q, lane & 15);
        }
    };
    auto store_tile = [&](const v4d &t, double *dst) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[((lane >> 4) + 4 * q) * LDT + (lane & 15)] = t[q];
    };

    v4d acc[RPW][KB];
    if (upd) {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
            for (int b = 0; b < KB; ++b) load_tile(acc[rr][b], rr * NW + w, b, std::false_type{});
    }

    // the two panel tiles of step sx that never enter the window, (sx+KB, sx) and (sx, sx+KB): raw band entries, fetched
    // into registers one step ahead and put into the panel buffer when that panel is extracted
    double xt[NX];
    auto fetch_extra = [&](double (&xd)[NX], int sx, auto fc) __attribute__((always_inline)) {
        const bool in0 = decltype(fc)::value || tile_inside(sx + KB, sx), in1 = decltype(fc)::value || tile_inside(sx, sx + KB);
        const double *b0 = in0 ? tile_base(sx + KB, sx) : lv.p, *b1 = in1 ? tile_base(sx, sx + KB) : lv.p;
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int e = tid + q * NTU;      // update threads only: the panel wave is busy with the LU when this runs
            const int row = (e >> 4) & 15, col = e & 15;
            if (e >= 512 || !upd) xd[q] = 0.0;
            else if ((e >> 8) == 0) xd[q] = in0 ? b0[e & 255] : ldA(sx + KB, sx, row, col);
            else xd[q] = in1 ? b1[e & 255] : ldA(sx, sx + KB, row, col);
        }
    };
    // panel of step s: registers -> LDS buffer P; the freed slots take the entering tiles
    auto extract = [&](int s, double *P, int mode, const double (&xs)[NX], auto fc) __attribute__((always_inline)) {   // mode 0: everything, 1: the diagonal tile only, 2: all but it
        double *Pd = P, *Pc = P + TS, *Pr = P + TS + KB * TS;
        const int as = s % KB;
        // Entering tiles: block row s+KB (tiles (s+KB, s+1 .. s+KB): slots 1 .. KB of that row block, 16 KiB contiguous) and
        // block column s+KB (tile (s+I, s+KB): slot 2KB - I of row block s+I).  They all lie inside the chain iff block
        // s+KB does -- ONE uniform test for the whole step instead of bounds logic per tile (the wave that owns the pivot
        // row moves eight tiles here: it was the slowest wave of every step).
        const bool fast = decltype(fc)::value || 16 * (s + KB + 1) <= np;
        const double *rowbase = lv.p + ((rbg0 + s + KB) * lv.ntl) * 256;
        // ONE store site and ONE load site per accumulator tile, source and destination chosen by wave-uniform selects, and
        // all stores of the step BEFORE its first load.  With a load in each of two branches the compiler merged them through
        // temporaries and waited for every tile's loads on the spot; with store and load of a tile next to each other its
        // wait-count bookkeeping (registers with a load pending from the previous trip round the loop) made the store of
        // tile k wait for the loads of tile k-2: the wave that owns the panel row (eight tiles) spent 21 k cycles in this
        // phase, everybody else 12 k.
        if (upd) {
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) {
                    const int a = rr * NW + w;
#pragma unroll
                    for (int b = 0; b < KB; ++b) {
                        const bool isdiag = a == as && b == as;
                        if ((mode == 1 && !isdiag) || (mode == 2 && isdiag)) continue;
                        const bool isrow = a == as, iscol = !isrow && b == as;
                        if (!isrow && !iscol) continue;
                        const int J = (b - as + KB) % KB;           // row tile (s, s+J)
                        const int Jn = (b - as - 1 + KB) % KB + 1;  // its successor (s+KB, s+Jn)
                        const int I = (a - as + KB) % KB;           // column tile (s+I, s), I in 1..KB-1; successor (s+I, s+KB)
                        if (pass == 0) {
                            double *dst = isrow ? (J == 0 ? Pd : Pr + (J - 1) * TS) : Pc + (I - 1) * TS;
                            store_tile(acc[rr][b], dst);
                        } else if (fast) {
                            const double *src = isrow ? rowbase + Jn * 256 : lv.p + ((rbg0 + s + I) * lv.ntl + (2 * KB - I)) * 256;
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[rr][b][q] = src[64 * q + lane];
                        } else load_tile(acc[rr][b], isrow ? s + KB : s + I, isrow ? s + Jn : s + KB, std::false_type{});
                    }
                }
            }
        }
        if (mode != 1 && upd) {
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                const int e = tid + q * NTU;
                if (e < 512) {
                    const int row = (e >> 4) & 15, col = e & 15;
                    ((e >> 8) == 0 ? Pc : Pr)[(KB - 1) * TS + row * LDT + col] = xs[q];
                }
            }
        }
    };
    // trailing update of step s from panel buffer P: tile(s+I, s+J) -= L21[I] * U12[J]; diag_only: just the next diagonal
    // tile (I == J == 1), else everything but it
    auto update = [&](int s, const double *P, bool diag_only) __attribute__((always_inline)) {
        const double *Pc = P + TS, *Pr = P + TS + KB * TS;
        const int as = s % KB;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int a = rr * NW + w;
            const int I = (a - as - 1 + KB) % KB + 1;
            if (diag_only && I != 1) continue;
            const double *Lp = Pc + (I - 1) * TS + (lane & 15) * LDT + (lane >> 4);
            double la[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) la[q] = -Lp[4 * q];
            // U operands two tiles at a time (the reads of the next pair are in flight while this pair's MFMAs issue)
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const int J = (b - as - 1 + KB) % KB + 1;
                if ((I == 1 && J == 1) != diag_only) continue;
                const double *Up = Pr + (J - 1) * TS + (lane >> 4) * LDT + (lane & 15);
                double u[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) u[q] = Up[4 * q * LDT];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[rr][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[q], u[q], acc[rr][b], 0, 0, 0);
                if (b & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // 16 x 16 LU of the diagonal tile in panel buffer P (panel wave only); leaves the reciprocal pivots in rd.
    // The 16 pivots are ONE dependent chain (16384 of them per chain run through here one after another), and this wave
    // shares the LDS with eight busy update waves, so the chain must not go through LDS: lane j (< 16) holds COLUMN j of
    // the tile in 16 registers; per pivot k the pivot and the multipliers live in lane k and reach the other lanes by
    // v_readlane (scalar registers, no memory pipe), every lane then updates its own column with scalar-operand FMAs.
    // Two LDS round trips per tile (load, store) instead of one per pivot (first round-2 version: 700 cycles per pivot
    // alone, 1400 beside the update waves) or four to five plus an IEEE divide (round 1: 1100).  The reciprocal pivot is
    // v_rcp_f64 + two Newton steps.
    auto panel_lu = [&](int s, double *Pd, double *rd) __attribute__((always_inline)) { tile_lu_regs(s, Pd, rd, np, boost, nb, lane); };

    // Schedule (three barriers per step):
    //   P1(s)  everybody: the panel solves of step s (division-free: the LU left the reciprocal pivots)
    //   P2(s)  the owner of the NEXT diagonal tile updates it (4 MFMAs) and moves it to the other panel buffer
    //   P3(s)  panel wave: LU of that tile  ||  update waves: the rest of the trailing update of step s, the rest of
    //          panel s+1 to the other buffer (freed slots refilled with the entering tiles), the write-back of step s
    // so the 16 serial pivots of step s+1 hide behind everything else of step s.
    fetch_extra(xt, 0, std::false_type{});
    extract(0, lds, 0, xt, std::false_type{});
    fetch_extra(xt, 1, std::false_type{});
    __syncthreads();
    if (!upd) panel_lu(0, lds, rdiag);
    __syncthreads();
    auto block_step = [&](int s, auto fc) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fc)::value;
        double *cur = lds + (s & 1) * PSZ, *nxt = lds + ((s + 1) & 1) * PSZ;
        double *Pd = cur, *Pc = cur + TS, *Pr = cur + TS + KB * TS;
        const double *rd = rdiag + (s & 1) * 16;
        const bool more = FAST || s + 1 < nblk;
        // ---- P1(s): L21 = A21 U11^{-1} (one thread per row), U12 = L11^{-1} A12 (one thread per column)
        stamp(s, 0);
        panel_solves<KB, NT, (KB < 8)>(Pd, Pc, Pr, rd, tid);
        stamp(s, 1);
        __syncthreads();
        stamp(s, 2);
        // ---- P2(s)
        if (more && upd) {
            update(s, cur, true);
            extract(s + 1, nxt, 1, xt, fc);
        }
        stamp(s, 3);
        __syncthreads();
        stamp(s, 4);
        // ---- P3(s)
        if (!upd) {
            if (more) panel_lu(s + 1, nxt, rdiag + ((s + 1) & 1) * 16);
            stamp(s, 5);
        } else {
            // The write-back comes FIRST: the entering-tile loads of extract() must be the youngest vector-memory operations
            // of the step -- anything issued after them that the wave has to wait for (a store's address reload from the
            // spill area is enough: vector-memory operations return in order) would pay their whole global round trip.
            // write the finished block row / block column back: a tile inside the chain is 2 KiB contiguous in the
            // scratch; the 2 KB+1 tiles are dealt to the update waves, a tile = 4 LDS reads + 4 coalesced 512-byte stores
            for (int tile = w; tile <= 2 * KB; tile += NW) {
                int rb, cb;
                const double *T;
                if (tile == 0) { rb = s; cb = s; T = Pd; }
                else if (tile <= KB) { rb = s + tile; cb = s; T = Pc + (tile - 1) * TS; }
                else { rb = s; cb = s + tile - KB; T = Pr + (tile - KB - 1) * TS; }
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = T[((lane >> 4) + 4 * q) * LDT + (lane & 15)];
                if (FAST || tile_inside(rb, cb)) {
                    double *tp = tile_base(rb, cb);
#pragma unroll
                    for (int q = 0; q < 4; ++q) tp[64 * q + lane] = v[q];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rr2 = 16 * rb + (lane >> 4) + 4 * q, c = 16 * cb + (lane & 15);
                        if (rr2 < np && c < np) lu_put(lv, rs, rr2, c, v[q]);
                    }
                }
            }
            if (more) {
                // the two raw panel tiles of step s+2 are requested BEFORE the entering-tile loads of extract(): whatever
                // waits for them afterwards (their copy, or their way into a spill slot) then does not wait for the eight
                // younger tile loads as well (vector-memory operations return in order)
                double xn[NX];
                fetch_extra(xn, s + 2, fc);
                update(s, cur, false);
                stamp(s, 5);
                // xt changes hands BEFORE extract() issues its loads: the wait for xn's load must not stand behind them
                double xo[NX];
#pragma unroll
                for (int q = 0; q < NX; ++q) { xo[q] = xt[q]; xt[q] = xn[q]; }
                extract(s + 1, nxt, 2, xo, fc);
            }
        }
        stamp(s, 6);
        __syncthreads();
        stamp(s, 7);
    };
    // Interior steps touch only tiles that lie inside the chain (step s reaches block s + KB + 2 at most): a loop of their
    // own, compiled without the element-wise edge paths -- whose loop-invariant per-lane indices otherwise stay live across
    // the hot loop and pushed it into spills (a reload costs a memory round trip on the critical wave).  The last
    // KB + 2 steps of a chain take the general body.
    const int sfast = np / 16 - KB - 2;
    int s0 = 0;
    for (; s0 < sfast; ++s0) block_step(s0, std::true_type{});
    for (; s0 < nblk; ++s0) block_step(s0, std::false_type{});
    if (lane == 0 && !upd && nb) atomicAdd(nboost, nb);
}

template <int KB, int NW>
static hipError_t launch_factor_mfma_la_t(const LuView &lv, const ChainDesc *chains, int nchains, double boost,
                                          unsigned long long *nboost, hipStream_t st)
{
    const size_t shm = ((size_t)2 * (2 * KB + 1) * TS + 32) * sizeof(double);
    if (shm > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_factor_mfma_la<KB, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
    }
    unsigned long long *stamps = nullptr;
    if (getenv("SPIKE_FACTOR_STAMPS")) {   // diagnostic: print the phase times of steps 64..71 of workgroup 0
        if (hipMalloc((void **)&stamps, 128 * sizeof(unsigned long long)) != hipSuccess) stamps = nullptr;
        else (void)hipMemsetAsync(stamps, 0, 128 * sizeof(unsigned long long), st);
    }
    hipLaunchKernelGGL((k_factor_mfma_la<KB, NW>), dim3(nchains), dim3((NW + 1) * 64), shm, st, lv, chains, boost, nboost, stamps);
    hipError_t e = hipGetLastError();
    if (stamps) {
        unsigned long long hs[128];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hs, stamps, sizeof hs, hipMemcpyDeviceToHost);
        (void)hipFree(stamps);
        const char *names[7] = {"P1: panel solves", "barrier", "P2: next diag tile", "barrier", "P3: LU | wb+update", "P3: extract", "barrier"};
        for (int kind = 0; kind < 2; ++kind) {
            fprintf(stderr, "[factor stamps] %s wave, shader cycles per phase, steps 64..71:\n", kind ? "panel" : "update");
            for (int ph = 0; ph < 7; ++ph) {
                fprintf(stderr, "  %-22s", names[ph]);
                for (int sidx = 0; sidx < 8; ++sidx) fprintf(stderr, " %6lld", (long long)(hs[sidx * 16 + kind * 8 + ph + 1] - hs[sidx * 16 + kind * 8 + ph]));
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "  %-22s", "whole step");
            for (int sidx = 0; sidx < 7; ++sidx) fprintf(stderr, " %6lld", (long long)(hs[(sidx + 1) * 16 + kind * 8] - hs[sidx * 16 + kind * 8]));
            fprintf(stderr, "\n");
        }
    }
    return e;
}

// (The first in-place kernel -- one block step per pass over the window, 65 ms at K = 256 -- was removed in round 3; its
// successor below does two steps per pass.)
// That first kernel read and wrote the whole K x K window (512 KiB per chain at K = 256: 256 chains = 128 MiB, which
// no L2 holds) once per 16-column step: 63 us per step of which ~53 us are that pass -- the kernel moves ~8 TB/s through
// the memory side and is bound by it, not by its 256 MFMA tiles.  Here panel s is factored as before, then only the NEXT
// panel (block row / column s+1: 2 KB + 1 tiles) is brought up to date and factored out of a second LDS buffer, and the
// window pass applies both rank-16 updates to a tile while it is in registers: half the passes, 8 MFMAs per tile moved.
template <int KB, int NW>
__global__ __launch_bounds__(NW * 64) void k_factor_mfma_inplace2(LuView lv, const ChainDesc *chains,
                                                                  double boost, unsigned long long *nboost)
{
    constexpr int NT = NW * 64;
    constexpr int PSZ = (2 * KB + 1) * TS;
    extern __shared__ double lds[];
    double *PA = lds, *PB = lds + PSZ;
    double *rdA = lds + 2 * PSZ, *rdB = rdA + 16;   // reciprocal pivots of the two panels
    const ChainDesc cd = chains[blockIdx.x];
    const int64_t rs = cd.row0;
    const int np = cd.nrows;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = (np + 15) / 16;
    unsigned long long nb = 0;
    auto ldA = [&](int rb, int cb, int row, int col) __attribute__((always_inline)) -> double {
        const int r = 16 * rb + row, c = 16 * cb + col;
        if (r >= np || c >= np) return (r == c) ? 1.0 : 0.0;   // identity padding past the chain end
        return lu_get(lv, rs, r, c);
    };
    const int64_t rbg0 = rs >> 4;
    auto tile_inside = [&](int rb, int cb) __attribute__((always_inline)) -> bool {
        return 16 * (rb + 1) <= np && 16 * (cb + 1) <= np && cb - rb + lv.KB >= 0 && cb - rb + lv.KB < lv.ntl;
    };
    auto tile_base = [&](int rb, int cb) __attribute__((always_inline)) -> double * {
        return lv.p + ((rbg0 + rb) * lv.ntl + (cb - rb + lv.KB)) * 256;
    };
    auto load_tile = [&](double(&t)[4], int rb, int cb) __attribute__((always_inline)) {
        if (tile_inside(rb, cb)) {
            const double *tp = tile_base(rb, cb);
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = tp[64 * q + lane];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = ldA(rb, cb, (lane >> 4) + 4 * q, lane & 15);
        }
    };
    auto store_tile_g = [&](const double(&t)[4], int rb, int cb) __attribute__((always_inline)) {
        if (tile_inside(rb, cb)) {
            double *tp = tile_base(rb, cb);
#pragma unroll
            for (int q = 0; q < 4; ++q) tp[64 * q + lane] = t[q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 16 * rb + (lane >> 4) + 4 * q, c = 16 * cb + (lane & 15);
                if (r < np && c < np) lu_put(lv, rs, r, c, t[q]);
            }
        }
    };
    // tile `tile` of the panel of step s held in buffer P: 0 = diagonal, 1..KB = (s+tile, s), KB+1..2KB = (s, s+tile-KB)
    auto panel_tile = [&](double *P, int tile, int s, int &rb, int &cb) __attribute__((always_inline)) -> double * {
        if (tile == 0) { rb = s; cb = s; return P; }
        if (tile <= KB) { rb = s + tile; cb = s; return P + TS + (tile - 1) * TS; }
        rb = s; cb = s + tile - KB; return P + TS + KB * TS + (tile - KB - 1) * TS;
    };
    auto lds_put = [&](double *T, const double(&t)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) T[((lane >> 4) + 4 * q) * LDT + (lane & 15)] = t[q];
    };
    // tile -= L(I) * U(J) of the panel in P (I, J in 1..KB): four MFMAs
    auto rank16 = [&](v4d &acc, const double *P, int I, int J) __attribute__((always_inline)) {
        const double *Lp = P + TS + (I - 1) * TS + (lane & 15) * LDT + (lane >> 4);
        const double *Up = P + TS + KB * TS + (J - 1) * TS + (lane >> 4) * LDT + (lane & 15);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lp[4 * q], Up[4 * q * LDT], acc, 0, 0, 0);
    };
    auto load_panel = [&](int s, double *P) __attribute__((always_inline)) {
        for (int tile = w; tile <= 2 * KB; tile += NW) {
            int rb, cb;
            double *T = panel_tile(P, tile, s, rb, cb);
            double t[4];
            load_tile(t, rb, cb);
            lds_put(T, t);
        }
    };
    // the panel of step s+1 from the in-place values, brought up to date with the rank-16 update of panel s (in P)
    auto next_panel = [&](int s, const double *P, double *Q) __attribute__((always_inline)) {
        for (int tile = w; tile <= 2 * KB; tile += NW) {
            int rb, cb;
            double *T = panel_tile(Q, tile, s + 1, rb, cb);
            double t[4];
            load_tile(t, rb, cb);
            const int I = rb - s, J = cb - s;               // >= 1; beyond KB the band of step s does not reach the tile
            if (I <= KB && J <= KB) {
                v4d acc = {t[0], t[1], t[2], t[3]};
                rank16(acc, P, I, J);
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] = acc[q];
            }
            lds_put(T, t);
        }
    };
    auto factor_panel = [&](int s, double *P, double *rd) __attribute__((always_inline)) {
        __syncthreads();
        if (w == 0) tile_lu_regs(s, P, rd, np, boost, nb, lane);
        __syncthreads();
        panel_solves<KB, NT>(P, P + TS, P + TS + KB * TS, rd, tid);
        __syncthreads();
    };
    auto writeback = [&](int s, double *P) __attribute__((always_inline)) {
        for (int tile = w; tile <= 2 * KB; tile += NW) {
            int rb, cb;
            const double *T = panel_tile(P, tile, s, rb, cb);
            double t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = T[((lane >> 4) + 4 * q) * LDT + (lane & 15)];
            store_tile_g(t, rb, cb);
        }
    };
    // window pass: tiles (b0+I, b0+J), I, J = 1..KB, b0 = s + d (d = 1 with a second panel): -= panel s (where its band
    // reaches) and -= panel s+1.  Software-pipelined as in k_factor_mfma_inplace: the loads of the next pair of tiles are
    // issued before the MFMAs and stores of the current pair.
    auto window_pass = [&](int s, const double *P, const double *Q) __attribute__((always_inline)) {
        const int d = Q != nullptr ? 1 : 0, b0 = s + d;
        auto tile_on = [&](int t) __attribute__((always_inline)) -> bool {
            const int I = t / KB + 1, J = t % KB + 1;
            return t < KB * KB && 16 * (b0 + I) < np && 16 * (b0 + J) < np;
        };
        auto tile_load = [&](int t, double(&a)[4]) __attribute__((always_inline)) {
            if (tile_on(t)) load_tile(a, b0 + t / KB + 1, b0 + t % KB + 1);
        };
        auto tile_update_store = [&](int t, double(&a)[4]) __attribute__((always_inline)) {
            if (!tile_on(t)) return;
            const int I = t / KB + 1, J = t % KB + 1;
            v4d acc = {a[0], a[1], a[2], a[3]};
            if (I + d <= KB && J + d <= KB) rank16(acc, P, I + d, J + d);
            if (Q != nullptr) rank16(acc, Q, I, J);
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = acc[q];
            store_tile_g(a, b0 + I, b0 + J);
        };
        // PD tiles per half of the pipeline: 2 x PD tiles (2 KiB each) in flight or in work per wave -- with two (as in the
        // one-step kernel) a CU moved only ~14 GB/s each way here, the wave waiting a memory round trip per pair of tiles
        constexpr int PD = 4;
        double pa[PD][4], pb[PD][4];
#pragma unroll
        for (int u = 0; u < PD; ++u) tile_load(w + u * NW, pa[u]);
        for (int t0 = w; t0 < KB * KB; t0 += 2 * PD * NW) {
#pragma unroll
            for (int u = 0; u < PD; ++u) tile_load(t0 + (PD + u) * NW, pb[u]);
#pragma unroll
            for (int u = 0; u < PD; ++u) tile_update_store(t0 + u * NW, pa[u]);
#pragma unroll
            for (int u = 0; u < PD; ++u) tile_load(t0 + (2 * PD + u) * NW, pa[u]);
#pragma unroll
            for (int u = 0; u < PD; ++u) tile_update_store(t0 + (PD + u) * NW, pb[u]);
        }
    };
    int s = 0;
    while (s < nblk) {
        load_panel(s, PA);
        factor_panel(s, PA, rdA);
        writeback(s, PA);
        if (s + 1 < nblk) {
            next_panel(s, PA, PB);
            factor_panel(s + 1, PB, rdB);
            writeback(s + 1, PB);
            window_pass(s, PA, PB);
            s += 2;
        } else {
            window_pass(s, PA, nullptr);
            s += 1;
        }
        __syncthreads();
    }
    if (lane == 0 && w == 0 && nb) atomicAdd(nboost, nb);
}

template <int KB, int NW>
static hipError_t launch_factor_mfma_inplace2_t(const LuView &lv, const ChainDesc *chains, int nchains,
                                                double boost, unsigned long long *nboost, hipStream_t st)
{
    const size_t shm = ((size_t)2 * (2 * KB + 1) * TS + 32) * sizeof(double);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_factor_mfma_inplace2<KB, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_factor_mfma_inplace2<KB, NW>), dim3(nchains), dim3(NW * 64), shm, st, lv, chains, boost, nboost);
    return hipGetLastError();
}

template <int KB, int NW>
static hipError_t launch_factor_mfma_t(const LuView &lv, const ChainDesc *chains, int nchains, double boost,
                                       unsigned long long *nboost, hipStream_t st)
{
    const size_t shm = (size_t)(2 * KB + 1) * TS * sizeof(double);
    if (shm > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_factor_mfma<KB, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_factor_mfma<KB, NW>), dim3(nchains), dim3(NW * 64), shm, st, lv, chains, boost, nboost);
    return hipGetLastError();
}

hipError_t launch_factor_generic(double *lu, int64_t ld, int K, const ChainDesc *chains, int nchains, double boost,
                                 unsigned long long *nboost, hipStream_t st);

// lu: diagonal-major scratch (ld = n) for K <= 32, block-band scratch (launch_band_to_blocks) for K > 32
hipError_t launch_factor(double *lu, int64_t ld, int K, const ChainDesc *chains, int nchains, double boost,
                         unsigned long long *nboost, hipStream_t st)
{
    if (nchains <= 0) return hipSuccess;
    if (K <= 8 || K > 256) return launch_factor_generic(lu, ld, K, chains, nchains, boost, nboost, st);
    LuView lv;
    lv.p = lu; lv.ld = ld; lv.K = K; lv.KB = (K + 15) / 16; lv.ntl = K > 32 ? 2 * lv.KB + 1 : 0;
    if (K <= 16) return launch_factor_mfma_t<1, 1>(lv, chains, nchains, boost, nboost, st);
    if (K <= 32) return launch_factor_mfma_t<2, 2>(lv, chains, nchains, boost, nboost, st);
    lv.KB = lu_kb(K); lv.ntl = 2 * lv.KB + 1;   // block-band scratch: as wide as the kernels' window
    if (K <= 64) return launch_factor_mfma_la_t<4, 4>(lv, chains, nchains, boost, nboost, st);
    if (K <= 128) return launch_factor_mfma_la_t<8, 8>(lv, chains, nchains, boost, nboost, st);
    return launch_factor_mfma_inplace2_t<16, 8>(lv, chains, nchains, boost, nboost, st);
}

hipError_t launch_factor_generic(double *lu, int64_t ld, int K, const ChainDesc *chains, int nchains, double boost,
                         unsigned long long *nboost, hipStream_t st)
{
    if (nchains <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_factor, dim3(nchains), dim3(256), (size_t)(2 * (K > 0 ? K : 1)) * sizeof(double), st, lu, ld,
                       K, chains, boost, nboost);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// k_pack: LU band -> sweep tiles.  One wave per (chain, row block): inverts the R x R diagonal
// blocks of L (unit lower) and of D^{-1}U (unit upper), negates them, and scatters near and
// far entries into the lane-linear tile order of k_sweep.
// ------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(64) void k_pack(int DPW, int NW, const double *lu, int64_t ld, int K,
                                             const ChainDesc *chains, const GroupDesc *groups, double *Lt, double *Ut,
                                             double *dinv)
{
    constexpr int CPW = 64 / R;
    __shared__ double Mb[R * R];
    __shared__ double X[R * R];
    const int sb = blockIdx.x, p = blockIdx.y;
    const ChainDesc cd = chains[p];
    if (sb >= cd.nsteps) return;
    const GroupDesc gd = groups[p / CPW];
    const int c = p % CPW;
    const int j = threadIdx.x;
    const int64_t i0 = cd.row0 + (int64_t)sb * R;
    const int64_t tdbl = (int64_t)NW * DPW * 64;
    const int rows_here = (cd.nrows - sb * R) < R ? (cd.nrows - sb * R) : R;  // valid rows in this block
    const int Kn = K < R - 1 ? K : R - 1;

    // ---------------- L ----------------
    for (int t = j; t < R * R; t += 64) Mb[t] = 0.0;
    __syncthreads();
    for (int d = 1; d <= Kn; ++d) {
        const int r = j;
        if (r < rows_here && r >= d) Mb[r * R + (r - d)] = lu[(int64_t)(K - d) * ld + i0 + r];
    }
    __syncthreads();
    if (j < R) {
        for (int r = 0; r < R; ++r) {
            double acc = (r == j) ? 1.0 : 0.0;
            for (int cc = 0; cc < r; ++cc) acc -= Mb[r * R + cc] * X[cc * R + j];
            X[r * R + j] = acc;
        }
    }
    __syncthreads();
    {
        double *T = Lt + (gd.tile0 + sb) * tdbl;
        // near: entry (row r, column j<r) -> d = r-j, stored negated
        if (j < R)
            for (int r = j + 1; r < R; ++r) T[tile_elem(DPW, c * R + r, r - j)] = -X[r * R + j];
        // far: columns before the block, inside the partition
        const int r = j;
        if (r < rows_here)
            for (int d = r + 1; d <= K; ++d)
                if ((int64_t)sb * R + r - d >= 0) T[tile_elem(DPW, c * R + r, d)] = lu[(int64_t)(K - d) * ld + i0 + r];
    }
    __syncthreads();
    // ---------------- U (unit upper after row scaling by 1/diag) ----------------
    double di = 1.0;
    if (j < rows_here) {
        di = 1.0 / lu[(int64_t)K * ld + i0 + j];
        dinv[i0 + j] = di;
    }
    for (int t = j; t < R * R; t += 64) Mb[t] = 0.0;
    __syncthreads();
    for (int d = 1; d <= Kn; ++d) {
        const int r = j;
        if (r < rows_here && r + d < rows_here) Mb[r * R + (r + d)] = lu[(int64_t)(K + d) * ld + i0 + r] * di;
    }
    __syncthreads();
    if (j < R) {
        for (int r = R - 1; r >= 0; --r) {
            double acc = (r == j) ? 1.0 : 0.0;
            for (int cc = r + 1; cc < R; ++cc) acc -= Mb[r * R + cc] * X[cc * R + j];
            X[r * R + j] = acc;
        }
    }
    __syncthreads();
    {
        double *T = Ut + (gd.tile0 + (cd.nsteps - 1 - sb)) * tdbl;
        if (j < R)
            for (int r = 0; r < j; ++r) T[tile_elem(DPW, c * R + (R - 1 - r), j - r)] = -X[r * R + j];
        const int r = j;
        if (r < rows_here)
            for (int d = R - r; d <= K; ++d)
                if ((int64_t)sb * R + r + d < cd.nrows)
                    T[tile_elem(DPW, c * R + (R - 1 - r), d)] = lu[(int64_t)(K + d) * ld + i0 + r] * di;
    }
}

// ------------------------------------------------------------------------------------------
// k_pack64: k_pack for R = 64 (one chain per workgroup, K > 32).
//  * the 64 x 64 unit-triangular diagonal block sits in LDS as one square and is inverted IN PLACE (row q of the inverse
//    overwrites row q of the block as soon as every lane has used it): 32 KiB per wave instead of the generic kernel's
//    two squares, i.e. 5 waves per CU instead of 2 -- the generic kernel was the largest item of setup;
//  * D^{-1}U is handled by the same routine on flipped indices (an upper triangle read back to front is a lower one);
//  * the tile is written with lane = tile row, one 16-byte store per lane and (wave, i) pair: 1 KiB contiguous per
//    instruction, every entry of the tile exactly once (zeros included).
// (A variant with the inverse's column in 64 registers and a fully unrolled substitution was tried: the 2016 independent
//  multiplier reads get hoisted above the serial FMA chain and spill ~7000 registers; measured 107 ms.)
// ------------------------------------------------------------------------------------------
// The 64 x 64 triangular block of k_pack64 in LDS: only its ten 16 x 16 tiles on and below the block diagonal, tile (tr, tc)
// at slot tr (tr + 1) / 2 + tc, rows of a tile 17 doubles apart (conflict-free by rows and by columns): 21.3 KiB instead of
// the 32.5 KiB of a padded square -- with the 16.1 KiB strip that is FOUR one-wave workgroups per CU instead of three, and
// the kernel's time goes with its occupancy (measured by giving it unused LDS: 1 / 2 / 3 workgroups per CU 10.9 / 5.7 / 4.0 ms).
constexpr int PACK_TS = 16 * 17;            // doubles per tile
constexpr int PACK_MS = 10 * PACK_TS;       // doubles of the block
__device__ __forceinline__ int pack_ms(int r, int c)   // element (r, c), c's tile column <= r's tile row
{
    const int tr = r >> 4, tc = c >> 4;
    return ((tr * (tr + 1)) / 2 + tc) * PACK_TS + (r & 15) * 17 + (c & 15);
}

// In-place inverse of a 64 x 64 UNIT LOWER triangular matrix held row-major in LDS (only the strictly lower triangle is
// read and written; the unit diagonal is implicit), by ONE wave:
//   A. the four 16 x 16 diagonal tiles by forward substitution, lane = (tile, column), the column in 16 registers;
//   B. the six off-diagonal tiles from  M X = I  in block form,  X_ij = -E_i * sum_{k=j}^{i-1} M_ik X_kj  (E_i = the inverted
//      diagonal tile, X_jj = E_j), as 16 x 16 x 16 products on v_mfma_f64_16x16x4: the A operands (M_ik, E_i) come from LDS,
//      the B operands are the D registers of earlier products -- the C/D layout of this instruction (row = (l>>4) + 4r,
//      column = l&15) IS its B layout for the k-slice r, so no value ever moves between lanes;
//   C. the X tiles go back over the M tiles.
// Round 1 ran a 64-step substitution with two LDS reads per multiply-add (about 150 us per block and wave, the 16 ms of
// k_pack64); this is 64 MFMAs plus 120 serial multiply-adds.
__device__ __forceinline__ void invert_unit_lower_64(double *Ms, int lane)
{
    typedef double v4 __attribute__((ext_vector_type(4)));
    const int li = lane & 15, lk = lane >> 4;
    // ---- A: diagonal tiles
    {
        const int tile = lk, j = li;
        const double *D = Ms + pack_ms(16 * tile, 16 * tile);   // diagonal tile: element (q, c) at D[q * 17 + c]
        double xc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            double acc = (q == j) ? 1.0 : 0.0;
#pragma unroll
            for (int c = 0; c < q; ++c) acc = fma(-D[q * 17 + c], xc[c], acc);
            xc[q] = acc;
        }
        WAVE_LDS_FENCE();
#pragma unroll
        for (int q = 1; q < 16; ++q)
            if (q > j) Ms[pack_ms(16 * tile + q, 16 * tile + j)] = xc[q];
        WAVE_LDS_FENCE();
    }
    // operand loaders.  A-layout: lane holds A[i = l&15][k = 4q + (l>>4)]; B-layout: B[k = 4q + (l>>4)][j = l&15]
    auto loadA_full = [&](int ti, int tk, double(&a)[4]) {     // off-diagonal tile (ti > tk) of M
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = Ms[pack_ms(16 * ti + li, 16 * tk + 4 * q + lk)];
    };
    auto loadA_diag = [&](int t, double(&a)[4]) {              // E_t: unit lower, stored strictly lower
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * q + lk;
            const double v = Ms[pack_ms(16 * t + li, 16 * t + (k < li ? k : 0))];
            a[q] = k < li ? v : (k == li ? 1.0 : 0.0);
        }
    };
    auto loadB_diag = [&](int t, v4 &b) {                      // E_t in B-layout
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * q + lk;                          // row of E_t, column li
            const double v = Ms[pack_ms(16 * t + (k > li ? k : 15), 16 * t + (k > li ? li : 0))];
            b[q] = k > li ? v : (k == li ? 1.0 : 0.0);
        }
    };
    auto gemm = [&](const double(&a)[4], const v4 &b, v4 acc) -> v4 {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
        return acc;
    };
    // ---- B: X[i][j] for i > j, by increasing distance i - j
    v4 X[4][4];
#pragma unroll
    for (int d = 1; d < 4; ++d) {
#pragma unroll
        for (int i = d; i < 4; ++i) {
            const int j = i - d;
            v4 S = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = j; k < i; ++k) {
                double a[4];
                loadA_full(i, k, a);
                v4 b;
                if (k == j) loadB_diag(j, b);
                else b = X[k][j];
                S = gemm(a, b, S);
            }
            double e[4];
            loadA_diag(i, e);
            const v4 zero = {0.0, 0.0, 0.0, 0.0};
            const v4 T = gemm(e, S, zero);
            X[i][j] = -T;
        }
    }
    WAVE_LDS_FENCE();   // every read of the M tiles precedes their replacement
    // ---- C
#pragma unroll
    for (int i = 1; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ms[pack_ms(16 * i + lk + 4 * r, 16 * j + li)] = X[i][j][r];
    WAVE_LDS_FENCE();
}

// SR = rows of the far-tile strip held in LDS at a time: 16 (a whole tile strip) or 8 (half of one)
template <bool UPPER, int SR>
__device__ __forceinline__ void pack64_side(double *Ms, double *Fs, double *dis, int DPW, int NW, const LuView &lv, int K,
                                            const ChainDesc &cd, int sb, double *T, double *dinv)
{
    constexpr int R = 64;
    const int lane = threadIdx.x;
    const int r = UPPER ? R - 1 - lane : lane;                 // block row this lane stands for
    const int64_t i0 = cd.row0 + (int64_t)sb * R;
    const int rows_here = (cd.nrows - sb * R) < R ? (cd.nrows - sb * R) : R;
    const bool rowok = r < rows_here;
    double di = 1.0;
    const int rl = sb * R + (rowok ? r : 0);                   // chain-local row of this lane (clamped)
    if (UPPER && rowok) {
        di = 1.0 / lu_get(lv, cd.row0, rl, rl);
        dinv[i0 + r] = di;
    }
    if (UPPER) dis[r] = di;                                     // 1/U_rr by block row, for the strip phase below
    // Ms[q][c] (row-major square): strict lower triangle = in-block entries (flipped row/column order for UPPER).
    // The lane's row of the 64 x 64 diagonal block is 4 x 128 contiguous bytes of the block-band scratch: all 32 16-byte
    // loads are issued unconditionally and back to back (one memory latency), the triangle is selected afterwards.
    // (Round 1 and the first block-band version loaded it one diagonal at a time under a per-lane condition: 63 dependent
    //  round trips per side -- that, not the inversion, was most of the kernel's 15 ms.)
    {
        const int rbase = sb * R;                                  // chain-local row/column of the block's first entry
        const int rb = rl >> 4;                                    // the lane's 16-row block (chain-local)
        const double *rowp = lv.p + (((cd.row0 >> 4) + rb) * (int64_t)lv.ntl) * 256 + (rl & 15) * 16;
        d2 rv[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int c = rbase + 2 * q;                           // chain-local column of the pair
            const int slot = (c >> 4) - rb + lv.KB;                // within [KB-3, KB+3]: always inside the scratch
            rv[q] = *reinterpret_cast<const d2 *>(rowp + (int64_t)slot * 256 + (c & 15));
        }
#pragma unroll
        for (int q = 0; q < 32; ++q) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int cc = 2 * q + e;                          // block column of this entry
                const double v = e == 0 ? rv[q].x : rv[q].y;
                if (!UPPER) { if ((cc >> 4) <= (lane >> 4)) Ms[pack_ms(lane, cc)] = (rowok && cc < lane) ? v : 0.0; }
                else {                                             // flipped: slot (lane, 63 - cc) <- U[r][cc] * di, cc > r
                    const bool use = rowok && cc > r && cc < rows_here;
                    if (((R - 1 - cc) >> 4) <= (lane >> 4)) Ms[pack_ms(lane, R - 1 - cc)] = use ? v * di : 0.0;
                }
            }
        }
    }
    WAVE_LDS_FENCE();
    invert_unit_lower_64(Ms, lane);   // the strictly lower triangle of Ms now holds that of M^{-1}
    // The tile is written 16 rows at a time.  Entry d of tile lane L (row L of the block; flipped for the upper factor):
    // in-block (d <= L) = -X[L][L-d] from Ms, else the band entry d columns away -- the FAR part, K entries per row that
    // lie in the KBv 16 x 16 scratch tiles beside the diagonal block.  Those tiles are brought in whole (two coalesced
    // 1-KiB loads each) into the LDS strip Fs and picked from there with lane = (row in the strip, slot group).  Round 1/2a
    // gathered them with lane = row, one 8-byte load per (row, d): every load instruction touched 64 different 128-byte
    // lines, 128 instructions per side, and the L1 asked the L2 four times for every line (PMC) -- 11.4 ms for a kernel that
    // moves 17.7 GB.
    (void)DPW;
    const int KBv = (K + 15) >> 4;                             // far tiles per strip, at most
    const int FS = 16 * KBv + 1;                               // strip row stride (odd: FS + 1 = 2 mod 16 -> no bank conflicts)
    // SR = 8 (K > 128): the strip in LDS holds half of a 16-row tile strip, 16 KiB at K = 256 -- four workgroups per CU instead
    // of two: 10.2 -> 6.3 ms at K = 256.  At K <= 128 the whole strip leaves room for four workgroups already and half strips
    // were measured slightly slower (3.45 -> 3.57 ms: half the prefetch distance, 128-byte instead of 256-byte store runs).
    constexpr int NH = 16 / SR;                               // strip pieces per tile strip
    constexpr int QN = 64 / SR;                               // slot groups among the lanes
    const int rowS = lane % SR, q = lane / SR;
    const int nblk16 = (cd.nrows + 15) >> 4;
    d2 *T2 = reinterpret_cast<d2 *>(T);
    const int nslots = 16 * NW;
    // strip rbl = rows 16*rbl .. 16*rbl+15 of the block in STORAGE order; tile lane of strip row t: L = 16*rbl + t for the
    // lower factor, 63 - (16*rbl + t) for the upper one (the upper tile is stored flipped).  The tiles of strip rbl+1 are
    // requested (into registers) before strip rbl is written out, so their latency hides behind that phase.
    d2 tv[2][16];                                              // up to 16 tiles (K = 256), two 16-byte pieces per lane each
    auto strip_of = [&](int rbl, int &rb, int &ntf, int &cbf) {
        rb = 4 * sb + rbl;                                     // chain-local 16-row block of the strip
        ntf = UPPER ? KBv + rbl - 3 : KBv - rbl;               // far tiles this strip needs (may be <= 0)
        cbf = UPPER ? 4 * sb + 4 : rb - KBv;                   // first of them (block column, chain-local)
    };
    auto issue_strip = [&](int rbl) {
        int rb, ntf, cbf;
        strip_of(rbl, rb, ntf, cbf);
        const int rbc = rb < nblk16 ? rb : nblk16 - 1;         // clamped for addresses (rows past the chain end read as 0)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            if (8 * ch >= ntf) continue;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = 8 * ch + jj < ntf ? 8 * ch + jj : ntf - 1;   // unconditional loads from clamped tiles
                const int slot = (cbf + j) - rb + lv.KB;                     // in [0, 2 KB] by construction
                const d2 *tp = reinterpret_cast<const d2 *>(lv.p + (((cd.row0 >> 4) + rbc) * (int64_t)lv.ntl + slot) * 256);
                tv[ch][2 * jj] = tp[lane];
                tv[ch][2 * jj + 1] = tp[lane + 64];
            }
        }
    };
    auto store_strip = [&](int rbl, int hh) {                  // SR = 8: rows 8 hh .. 8 hh + 7 of the strip's tiles = the lanes' piece hh
        int rb, ntf, cbf;
        strip_of(rbl, rb, ntf, cbf);
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            if (8 * ch >= ntf) continue;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = 8 * ch + jj;
                if (j >= ntf) continue;
                // pair index inside the tile: lane + 64 piece = row (lane >> 3) + 8 piece, columns 2 (lane & 7), +1
                if (SR == 8) {
                    double *dst = Fs + (lane >> 3) * FS + 16 * j + 2 * (lane & 7);
                    const d2 t = hh == 0 ? tv[ch][2 * jj] : tv[ch][2 * jj + 1];
                    dst[0] = t.x; dst[1] = t.y;
                } else {
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        double *dst = Fs + ((lane >> 3) + 8 * pc) * FS + 16 * j + 2 * (lane & 7);
                        dst[0] = tv[ch][2 * jj + pc].x; dst[1] = tv[ch][2 * jj + pc].y;
                    }
                }
            }
        }
    };
    issue_strip(0);
    for (int rbl = 0; rbl < 4; ++rbl) {
        int rb, ntf, cbf;
        strip_of(rbl, rb, ntf, cbf);
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            store_strip(rbl, hh);
            WAVE_LDS_FENCE();
            if (hh == NH - 1 && rbl < 3) issue_strip(rbl + 1);  // (the registers of this strip's tiles are free from here on)
            const int rin = 16 * rbl + SR * hh + rowS;             // row of the block (storage order)
            const int L = UPPER ? R - 1 - rin : rin;               // its tile lane
            const int rloc = sb * R + rin;                         // chain-local row
            const bool rok = rin < rows_here;
            const double dsc = UPPER ? dis[rin] : 1.0;
            auto entry = [&](int d) -> double {
                if (d <= L) return -Ms[pack_ms(L, L - d)];
                const bool far_ok = d <= K && rok && (UPPER ? (rloc + d < cd.nrows) : (rloc - d >= 0));
                const int c = UPPER ? rloc + d : rloc - d;         // chain-local column
                const int fc = far_ok ? c - 16 * cbf : 0;          // column inside the strip
                const double g = Fs[rowS * FS + fc];
                return far_ok ? g * dsc : 0.0;
            };
            for (int s0 = 0; s0 < nslots; s0 += 16) {
                d2 v[16 / QN];
#pragma unroll
                for (int u = 0; u < 16 / QN; ++u) {
                    const int slot = s0 + QN * u + q;
                    v[u].x = entry(2 * slot + 1);
                    v[u].y = entry(2 * slot + 2);
                }
#pragma unroll
                for (int u = 0; u < 16 / QN; ++u) T2[(int64_t)(s0 + QN * u + q) * 64 + L] = v[u];
            }
            WAVE_LDS_FENCE();                                      // the strip is overwritten by the next half
        }
    }
}

template <int SR>
__global__ __launch_bounds__(64) void k_pack64(int DPW, int NW, LuView lv, int K, const ChainDesc *chains,
                                               const GroupDesc *groups, double *Lt, double *Ut, double *dinv)
{
    __shared__ double Ms[PACK_MS];
    __shared__ double dis[64];
    extern __shared__ double Fs[];   // SR x (16 ceil(K/16) + 1): the strip of far tiles (SR = 8: half of it at a time)
    const int sb = blockIdx.x, p = blockIdx.y;
    const ChainDesc cd = chains[p];
    if (sb >= cd.nsteps) return;
    const GroupDesc gd = groups[p];
    const int64_t tdbl = (int64_t)NW * DPW * 64;
    pack64_side<false, SR>(Ms, Fs, dis, DPW, NW, lv, K, cd, sb, Lt + (gd.tile0 + sb) * tdbl, dinv);
    WAVE_LDS_FENCE();
    pack64_side<true, SR>(Ms, Fs, dis, DPW, NW, lv, K, cd, sb, Ut + (gd.tile0 + (cd.nsteps - 1 - sb)) * tdbl, dinv);
}

hipError_t launch_pack(const SweepCfg &cfg, const double *lu, int64_t ld, int K, const ChainDesc *chains,
                       const GroupDesc *groups, int nchains, int64_t maxsteps, const int64_t *, double *Lt, double *Ut,
                       double *dinv, hipStream_t st)
{
    if (nchains <= 0 || maxsteps <= 0) return hipSuccess;
    dim3 grid((unsigned)maxsteps, (unsigned)nchains);
    switch (cfg.R) {
    case 4: hipLaunchKernelGGL((k_pack<4>), grid, dim3(64), 0, st, cfg.DPW, cfg.NW, lu, ld, K, chains, groups, Lt, Ut, dinv); break;
    case 8: hipLaunchKernelGGL((k_pack<8>), grid, dim3(64), 0, st, cfg.DPW, cfg.NW, lu, ld, K, chains, groups, Lt, Ut, dinv); break;
    case 16: hipLaunchKernelGGL((k_pack<16>), grid, dim3(64), 0, st, cfg.DPW, cfg.NW, lu, ld, K, chains, groups, Lt, Ut, dinv); break;
    case 32: hipLaunchKernelGGL((k_pack<32>), grid, dim3(64), 0, st, cfg.DPW, cfg.NW, lu, ld, K, chains, groups, Lt, Ut, dinv); break;
    case 64: {
        if (K > 256) {   // generic path: diagonal-major scratch, two 64 x 64 squares in LDS (the kernel zero-fills nothing: the
                         // caller clears the tiles)
            hipLaunchKernelGGL((k_pack<64>), grid, dim3(64), 0, st, cfg.DPW, cfg.NW, lu, ld, K, chains, groups, Lt, Ut, dinv);
            break;
        }
        LuView lv;   // K > 32: the scratch is block-band (launch_band_to_blocks / launch_factor)
        lv.p = const_cast<double *>(lu); lv.ld = ld; lv.K = K; lv.KB = lu_kb(K); lv.ntl = 2 * lv.KB + 1;
        // (the occupancy experiment behind the LDS layout -- the same kernel handed 28 / 60 KiB of unused LDS, i.e. 2 / 1 workgroups
        //  per CU instead of 3: 5.7 / 10.9 ms against 4.0 -- is recorded in DESIGN.md section 4)
        const size_t fs_row = (size_t)(16 * ((K + 15) / 16) + 1) * sizeof(double);
        if (K > 128) hipLaunchKernelGGL((k_pack64<8>), grid, dim3(64), 8 * fs_row, st, cfg.DPW, cfg.NW, lv, K, chains, groups, Lt, Ut, dinv);
        else hipLaunchKernelGGL((k_pack64<16>), grid, dim3(64), 16 * fs_row, st, cfg.DPW, cfg.NW, lv, K, chains, groups, Lt, Ut, dinv);
        break;
    }
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// small setup helpers
// ------------------------------------------------------------------------------------------
__global__ void k_absmax_diag(const double *band, int64_t ld, int K, int64_t n, unsigned long long *out)
{
    double m = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = fabs(band[(int64_t)K * ld + i]);
        m = v > m ? v : m;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double t = __shfl_down(m, o);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

hipError_t launch_absmax_diag(const double *band, int64_t ld, int K, int64_t n, double *out, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double), st);
    if (e != hipSuccess) return e;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_absmax_diag, dim3(grid), dim3(256), 0, st, band, ld, K, n, (unsigned long long *)out);
    return hipGetLastError();
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// must stay bit-identical to oracle/spike_oracle.c:orc_gen_band (no fp contraction)
__global__ void k_gen_band(int64_t N, int K, uint64_t seed, double delta, int64_t row0, int64_t nrows, double *band,
                           int64_t ld)
{
#pragma clang fp contract(off)
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    const int nd = 2 * K + 1;
    const int64_t gi = row0 + i;
    double s = 0.0;
    for (int d = 0; d < nd; ++d) {
        const int64_t c = gi + d - K;
        double v = 0.0;
        if (d != K && c >= 0 && c < N) {
            const uint64_t z = splitmix64(seed ^ ((uint64_t)gi * (uint64_t)nd + (uint64_t)d));
            v = (double)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
            s += fabs(v);
        }
        band[(int64_t)d * ld + i] = v;
    }
    band[(int64_t)K * ld + i] = delta * s + (s == 0.0 ? 1.0 : 0.0);
}

hipError_t launch_gen_band(int64_t N, int K, uint64_t seed, double delta, int64_t row0, int64_t nrows, double *band,
                           int64_t ld, hipStream_t st)
{
    if (nrows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_gen_band, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, st, N, K, seed, delta, row0,
                       nrows, band, ld);
    return hipGetLastError();
}

// y = A x, rows local, xh = x extended by K halo entries on both sides (xh[K + j] = x[j]).
// HBM-bound on the band ((2K+1)*8 bytes per row): a block owns 512 rows, stages its x window (512 + 2K doubles)
// in LDS once, and every lane streams TWO adjacent rows of each diagonal with one 16-byte load.
template <int ROWS>
__global__ __launch_bounds__(256) void k_band_matvec(int64_t n_global, int64_t row0, int64_t n, int K, const double *band,
                                                     int64_t ld, const double *xh, double *y)
{
    extern __shared__ double xs[];  // ROWS + 2K
    const int64_t i0 = (int64_t)blockIdx.x * ROWS;
    const int nw = ROWS + 2 * K;
    for (int t = threadIdx.x; t < nw; t += 256) xs[t] = (i0 + t < n + 2 * K) ? xh[i0 + t] : 0.0;
    __syncthreads();
    const int r = 2 * threadIdx.x;  // local row pair (r, r+1)
    const int64_t i = i0 + r;
    if (i >= n) return;
    const bool pair = (i + 1 < n) && ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(band) & 15) == 0);
    double s0 = 0.0, s1 = 0.0;
    if (pair) {
        for (int d = 0; d <= 2 * K; ++d) {
            const d2 b = *reinterpret_cast<const d2 *>(band + (int64_t)d * ld + i);
            s0 = fma(b.x, xs[r + d], s0);
            s1 = fma(b.y, xs[r + 1 + d], s1);
        }
    } else {
        for (int d = 0; d <= 2 * K; ++d) {
            s0 = fma(band[(int64_t)d * ld + i], xs[r + d], s0);
            if (i + 1 < n) s1 = fma(band[(int64_t)d * ld + i + 1], xs[r + 1 + d], s1);
        }
    }
    // entries whose column falls outside [0, n_global) are ignored by contract: the generator and the CSR scatter
    // store zeros there, and the halo of the first/last rank is zero, so no masking is needed here.
    y[i] = s0;
    if (i + 1 < n) y[i + 1] = s1;
}

hipError_t launch_band_matvec(int64_t n_global, int64_t row0, int64_t n, int K, const double *band, int64_t ld,
                              const double *xh, double *y, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    constexpr int ROWS = 512;
    hipLaunchKernelGGL((k_band_matvec<ROWS>), dim3((unsigned)((n + ROWS - 1) / ROWS)), dim3(256),
                       (size_t)(ROWS + 2 * K) * sizeof(double), st, n_global, row0, n, K, band, ld, xh, y);
    return hipGetLastError();
}

// Tile-major copy of the band for the Krylov mat-vec: block b = 128 consecutive rows, inside it diagonal-major with two
// adjacent rows per lane:  At[((b*(2K+1) + d)*64 + lane)*2 + e] = A[128 b + 2 lane + e, . + d - K].
// A wave then streams ONE contiguous (2K+1) KiB region instead of 2K+1 regions that lie N*8 bytes apart
// (one DRAM/TLB page each): 5.9 -> 6.6 TB/s at N = 4M, K = 128.  Rows past n are stored as zeros.
__global__ __launch_bounds__(256) void k_band_to_tiles(int64_t n, int K, const double *band, int64_t ld, double *At)
{
    const int nd = 2 * K + 1;
    const int64_t b = blockIdx.x;
    for (int t = threadIdx.x; t < nd * 128; t += 256) {
        const int d = t / 128, r = t % 128;  // r = 2*lane + e: consecutive threads -> consecutive rows of one diagonal
        const int64_t i = b * 128 + r;
        At[(b * nd + d) * 128 + r] = (i < n) ? band[(int64_t)d * ld + i] : 0.0;
    }
}

hipError_t launch_band_to_tiles(int64_t n, int K, const double *band, int64_t ld, double *At, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_band_to_tiles, dim3((unsigned)((n + 127) / 128)), dim3(256), 0, st, n, K, band, ld, At);
    return hipGetLastError();
}

// y = A x from the tile-major copy; a workgroup owns 512 rows (4 waves x 128 rows), x window in LDS as above
__global__ __launch_bounds__(256) void k_band_matvec_tiled(int64_t n, int K, const double *At, const double *xh, double *y)
{
    extern __shared__ double xs[];  // 512 + 2K
    const int64_t i0 = (int64_t)blockIdx.x * 512;
    const int nw = 512 + 2 * K;
    for (int t = threadIdx.x; t < nw; t += 256) xs[t] = (i0 + t < n + 2 * K) ? xh[i0 + t] : 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * 4 + w;
    if (b * 128 >= n) return;
    const int nd = 2 * K + 1;
    const d2 *T = reinterpret_cast<const d2 *>(At) + b * nd * 64 + lane;
    const double *xp = xs + w * 128 + 2 * lane;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
    for (int d = 0; d < nd; ++d) {
        const d2 a = __builtin_nontemporal_load(T + (int64_t)d * 64);
        s0 = fma(a.x, xp[d], s0);
        s1 = fma(a.y, xp[d + 1], s1);
    }
    const int64_t i = b * 128 + 2 * lane;
    if (i < n) y[i] = s0;
    if (i + 1 < n) y[i + 1] = s1;
}

hipError_t launch_band_matvec_tiled(int64_t n, int K, const double *At, const double *xh, double *y, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_band_matvec_tiled, dim3((unsigned)((n + 511) / 512)), dim3(256), (size_t)(512 + 2 * K) * sizeof(double),
                       st, n, K, At, xh, y);
    return hipGetLastError();
}

// ---- spike tips by K pairs of sweeps (setup) -------------------------------------------------
// C_p(a,b) = A[s+a, s-K+b]  -> band slot d = b - a        (a <= b)
// B_p(a,b) = A[e-K+a, e+b]  -> band slot d = 2K + b - a   (b <= a)
// (whether a chain end has a neighbour is the host's knowledge: ChainDesc::flags -- for the bottom half of a twisted
//  partition "top" is the partition's LAST row and "bottom" the seam, see spike_internal.h)
// entry (chain-local row rl, diagonal d) of the chain's band in FACTOR space, read from the caller's band (vector space): a
// vdir = -1 chain is stored with rows reversed and diagonals mirrored (d <-> 2K - d).  Setup's readers of the coupling blocks
// take the chains WITH their vector map and the natural band, so no flipped copy of the band is ever made.
__device__ __forceinline__ double band_at(const ChainDesc &cd, const double *band, int64_t ld, int K, int rl, int d)
{
    return band[(int64_t)(cd.vdir > 0 ? d : 2 * K - d) * ld + cd.vec0 + (int64_t)cd.vdir * rl];
}
__device__ __forceinline__ bool has_top(const ChainDesc &cd) { return (cd.flags & CHAIN_HAS_TOP) != 0; }
__device__ __forceinline__ bool has_bot(const ChainDesc &cd) { return (cd.flags & CHAIN_HAS_BOT) != 0; }

__global__ void k_tip_rhs(const double *band, int64_t ld, int K, const ChainDesc *chains, int nchains, int which, int col,
                          double *rhs, int64_t ldr)
{
    const int p = blockIdx.x;
    const ChainDesc cd = chains[p];
    col += blockIdx.y;            // column col0 + q goes to right-hand side q
    rhs += blockIdx.y * ldr;
    for (int a = threadIdx.x; a < K; a += blockDim.x) {
        if (which == 0) {
            if (!has_top(cd)) continue;
            rhs[cd.row0 + a] = (a <= col) ? band_at(cd, band, ld, K, a, col - a) : 0.0;
        } else {
            if (!has_bot(cd)) continue;
            rhs[cd.row0 + cd.nrows - K + a] = (col <= a) ? band_at(cd, band, ld, K, cd.nrows - K + a, 2 * K + col - a) : 0.0;
        }
    }
}

hipError_t launch_tip_rhs(const double *band, int64_t ld, int K, const ChainDesc *chains, int nchains, int which, int col,
                          double *rhs, hipStream_t st, int ncols, int64_t ldr)
{
    if (nchains <= 0 || K <= 0 || ncols <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_tip_rhs, dim3(nchains, ncols), dim3(64), 0, st, band, ld, K, chains, nchains, which, col, rhs, ldr);
    return hipGetLastError();
}

__global__ void k_tip_gather(const double *sol, int K, const ChainDesc *chains, int which, int col, double *out, int64_t ldr)
{
    const int p = blockIdx.x;
    const ChainDesc cd = chains[p];
    col += blockIdx.y;
    sol += blockIdx.y * ldr;
    for (int a = threadIdx.x; a < K; a += blockDim.x) {
        const int64_t r = (which == 0) ? cd.row0 + a : cd.row0 + cd.nrows - K + a;
        out[((int64_t)p * K + a) * K + col] = sol[r];
    }
}

hipError_t launch_tip_gather(const double *sol, int K, const ChainDesc *chains, int nchains, int which, int col,
                             double *out, hipStream_t st, int ncols, int64_t ldr)
{
    if (nchains <= 0 || K <= 0 || ncols <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_tip_gather, dim3(nchains, ncols), dim3(64), 0, st, sol, K, chains, which, col, out, ldr);
    return hipGetLastError();
}

__global__ void k_coupling_blocks(const double *band, int64_t ld, int K, const ChainDesc *chains, int which, double *out)
{
    const int p = blockIdx.x;
    const ChainDesc cd = chains[p];
    const bool on = which == 0 ? has_top(cd) : has_bot(cd);
    for (int t = threadIdx.x; t < K * K; t += blockDim.x) {
        const int a = t % K, b = t / K;  // column-major: out[b*K + a]
        double v = 0.0;
        if (on) {
            if (which == 0) { if (a <= b) v = band_at(cd, band, ld, K, a, b - a); }
            else { if (b <= a) v = band_at(cd, band, ld, K, cd.nrows - K + a, 2 * K + b - a); }
        }
        out[(int64_t)p * K * K + t] = v;
    }
}

hipError_t launch_coupling_blocks(const double *band, int64_t ld, int K, const ChainDesc *chains, int nchains, int which,
                                  double *out, hipStream_t st)
{
    if (nchains <= 0 || K <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_coupling_blocks, dim3(nchains), dim3(256), 0, st, band, ld, K, chains, which, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// k_spike_trsm: the 2K spike columns of every chain by a blocked banded triangular solve on MFMA (round 2).
// Round 1/2a solved them as right-hand sides of the SWEEP kernels, 2-4 columns per pass over the packed factors: 32-64
// passes, each bound by the per-column LDS window traffic (19-21 ms at the headline size).  Here the factors are read in
// their dense 16 x 16 tile form (the block-band LU scratch, still alive at this point of setup) and ALL K columns go
// through at once: workgroup = (chain, side), wave = 16 columns, and a wave needs nobody else --
//   forward   Z_rb = L_rb,rb^{-1} (R_rb - sum_{d=1..KB} L_rb,rb-d Z_rb-d)          row blocks of 16, top to bottom
//   backward  X_rb = U_rb,rb^{-1} (Z_rb - sum_{d=1..KB} U_rb,rb+d X_rb+d)          bottom to top
// with the products on v_mfma_f64_16x16x4: A operand = factor tile straight from the scratch (L2), B operand = the last KB
// solved tiles, which stay in REGISTERS (the C/D layout of the instruction is its B layout), the 16 x 16 triangular solve
// with the diagonal tile by 16 lanes through a 2-KiB LDS tile (which also transposes the result for the column-major
// spike store).  Z goes through a scratch area in tile form (written once, read once).  Side 0 = W (right-hand side
// C_p in the top K rows, region = the first `region` rows: exact forward, backward started where the spike is below
// rounding), side 1 = V (B_p in the bottom K rows, region = the last `region` rows: the forward sweep starts at the block
// that holds row nrows-K, exact; backward from the chain end, exact).  Same truncation as the sweep-based solves.
// Needs K <= 128 (window of KB <= 8 tiles per wave) and every chain length a multiple of 16.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_max_pos(double *addr, double v);

struct TrsmArgs {
    LuView lv;
    const ChainDesc *chains;
    const double *band;      // band in factor space (diagonal-major): coupling blocks C, B for the right-hand sides
    int64_t ld;
    int K, m, region;        // region = rows solved next to the interface (multiple of 64)
    double *Wt, *Vb;         // tips, row-major K x K per chain
    double *Wf, *Vf;         // stored spikes, column-major K x m per chain (may be null when m == 0)
    double *zscratch;        // per (chain, side, wave): region/16 tiles of 256 doubles
    double *absmax_in, *absmax_edge;
    double *Tb, *Gb;         // twisted factorisation (else null): seam matrices, row-major K x K per chain (spike_internal.h)
    const double *dinv;      // 1 / U_ii in factor space (the sweeps work with D^-1 L^-1 and D^-1 U, this solve with L and U)
};

// ---- the solve without a serial core ---------------------------------------------------------------------------------------
// The first version of this solve (round 2a, removed in round 3) spent a block step on (i) eight to sixteen factor-tile loads
// that the compiler left next to their MFMAs (a memory round trip each), (ii) a 16-lane triangular substitution through LDS
// (120 dependent multiply-adds) that EVERY wave of a chain side repeated with the same diagonal tile: 9 us per step at
// K = 128.  Here the diagonal tiles of the region are inverted once per chain side (k_trsm_diag_inv: one wave per tile, lanes 0-15 a column of L^-1, lanes 16-31 a column of
// U^-1), so the triangular solve of a step is one more 16 x 16 x 16 product, and the A operands of step rb+1 (off-diagonal
// tiles, inverse tile, Z tile) are requested into registers before the MFMAs of step rb: no LDS, one exposed chain of
// 4 (KB + 1) dependent MFMAs per step.  The pack kernel treats the 64 x 64 diagonal blocks of the sweeps the same way.
__global__ __launch_bounds__(64) void k_trsm_diag_inv(TrsmArgs a, double *inv)
{
    __shared__ double T[16 * 17];
    const int lane = threadIdx.x;
    const int p = blockIdx.y, side = blockIdx.z;
    const ChainDesc cd = a.chains[p];
    const int np = cd.nrows;
    const int region = a.region < np ? a.region : np;
    const int NB = region / 16;
    const int rb = blockIdx.x;
    if (rb >= NB) return;
    const int rb0 = side == 0 ? 0 : (np - region) / 16;
    const double *tp = a.lv.p + (((cd.row0 >> 4) + rb0 + rb) * (int64_t)a.lv.ntl + a.lv.KB) * 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) T[((lane >> 4) + 4 * q) * 17 + (lane & 15)] = tp[64 * q + lane];
    WAVE_LDS_FENCE();
    const int NBmax = a.region / 16;
    double *out = inv + ((((int64_t)p * 2 + side) * NBmax + rb) * 2) * 256;   // [L^-1 | U^-1], row-major 16 x 16 each
    const int j = lane & 15;
    double x[16];
    if (lane < 16) {                     // column j of L^-1 (L unit lower: the stored strict lower triangle of the tile)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double v = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) v = fma(-T[i * 17 + k], x[k], v);
            x[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) out[i * 16 + j] = x[i];
    } else if (lane < 32) {              // column j of U^-1 (upper triangle with its diagonal)
#pragma unroll
        for (int i = 15; i >= 0; --i) {
            double v = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int k = i + 1; k < 16; ++k) v = fma(-T[i * 17 + k], x[k], v);
            x[i] = v / T[i * 17 + i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) out[256 + i * 16 + j] = x[i];
    }
}

template <int KB>
__global__ __launch_bounds__(KB == 16 ? 256 : 512) void k_spike_trsm2(TrsmArgs a, const double *inv)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwv = blockDim.x >> 6;
    const int wg = w + nwv * blockIdx.z;         // column tile of this wave
    const int ncw = nwv * gridDim.z;
    // side 0: W (top coupling block C), side 1: V (bottom coupling block B), side 2 (twisted factorisation only): the
    // backward solve of [0; I] on the last K rows, i.e. Gb = (D^-1 U)_bb^-1; sides 1 and 2 work on the same (bottom) region
    const int p = blockIdx.x, side = blockIdx.y, bside = side == 0 ? 0 : 1;
    const ChainDesc cd = a.chains[p];
    const int K = a.K, m = a.m;
    const int np = cd.nrows;
    const bool on = side == 0 ? has_top(cd) : has_bot(cd);
    const int c0 = 16 * wg;
    double *tips = (side == 0 ? a.Wt : (side == 1 ? a.Vb : a.Gb)) + (int64_t)p * K * K;
    if (!on) return;                             // no neighbour on this side: the tips stay zero (setup cleared them)
    const bool fwd_only = side == 1 && a.Tb != nullptr;   // twisted: the seam needs the forward-swept block only
    const int region = a.region < np ? a.region : np;
    const int NB = region / 16;
    const int rb0 = bside == 0 ? 0 : (np - region) / 16;
    const int rbs = bside == 0 ? 0 : (np - K) / 16 - rb0;
    const int64_t rbg0 = (cd.row0 >> 4) + rb0;
    const LuView lv = a.lv;
    double *zs = a.zscratch + (((int64_t)p * 2 + bside) * ncw + wg) * (int64_t)NB * 256;
    const double *invp = inv + (((int64_t)p * 2 + bside) * (a.region / 16)) * 512;
    const int li = lane & 15, lk = lane >> 4;
    typedef double v4 __attribute__((ext_vector_type(4)));
    // ROW PERMUTATION.  The MFMA layouts are fixed: A operand lane (i = li, k = lk + 4q), C/D element r of lane = row lk + 4r.
    // Fed with the rows of a factor tile in natural order, the A operand of slice q is tile[li][lk + 4q]: four 8-byte loads
    // per tile, each touching all sixteen 128-byte lines of the tile for 32 bytes -- the step was bound by line requests to
    // the L1, not by memory or MFMA (K = 128: ~8 us per step whatever else changed).  With the block's rows taken in the
    // order pi(i') = 4 (i' mod 4) + i' / 4 -- consistently: A'[i'][k'] = tile[pi(i')][pi(k')], and every solved tile lives
    // in registers as X[pi(.)][.] -- the products are unchanged (pi is a bijection of the contraction index), element r of
    // a solved tile is row 4 lk + r, and the A operand of a lane is tile[pi(li)][4 lk .. 4 lk + 3]: ONE 32-byte load per
    // tile and lane, every line read once.
    const int pli = 4 * (li & 3) + (li >> 2);

    v4 Xw[KB];
#pragma unroll
    for (int q = 0; q < KB; ++q) Xw[q] = v4{0.0, 0.0, 0.0, 0.0};

    auto rhs_tile = [&](int rb) __attribute__((always_inline)) -> v4 {
        v4 t = {0.0, 0.0, 0.0, 0.0};
        const int col = c0 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (rb0 + rb) * 16 + 4 * lk + r;       // chain-local row (permuted layout: element r = row 4 lk + r)
            double v = 0.0;
            if (col < K) {
                if (side == 0) {
                    if (row < K && row <= col) v = band_at(cd, a.band, a.ld, K, row, col - row);
                } else {
                    const int aa = row - (np - K);
                    if (aa >= 0 && col <= aa) v = band_at(cd, a.band, a.ld, K, row, 2 * K + col - aa);
                }
            }
            t[r] = v;
        }
        return t;
    };
    // A operands of one block step: the KB off-diagonal factor tiles, the inverse of the diagonal tile, (backward) the Z tile
    struct StepOps { v4 A[KB]; v4 I; v4 Z; };
    auto fetch = [&](StepOps &o, int rb, bool bwd) __attribute__((always_inline)) {
        const int rbc = rb < 0 ? 0 : (rb >= NB ? NB - 1 : rb);  // clamped: a step past the range fetches a valid block, unused
        const double *rowp = lv.p + ((rbg0 + rbc) * lv.ntl) * 256 + pli * 16 + 4 * lk;
#pragma unroll
        for (int d = 1; d <= KB; ++d)
            o.A[d - 1] = *reinterpret_cast<const v4 *>(rowp + (int64_t)(bwd ? lv.KB + d : lv.KB - d) * 256);
        o.I = *reinterpret_cast<const v4 *>(invp + (int64_t)rbc * 512 + (bwd ? 256 : 0) + pli * 16 + 4 * lk);
        if (bwd) o.Z = *reinterpret_cast<const v4 *>(zs + (int64_t)rbc * 256 + 4 * lane);
    };
    auto solve_diag = [&](const v4 &I, const v4 &B) __attribute__((always_inline)) -> v4 {
        v4 r = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) r = __builtin_amdgcn_mfma_f64_16x16x4f64(I[q], B[q], r, 0, 0, 0);
        return r;
    };

    // ---------------- forward: Z ----------------
    StepOps cur, nxt;
    const int fstart = rbs - rbs % KB;
    if (side != 2) fetch(nxt, rbs, false);
    for (int rbb = fstart; rbb < NB && side != 2; rbb += KB) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int rb = rbb + u;
            if (rb < rbs || rb >= NB) continue;
            cur = nxt;
            fetch(nxt, rb + 1, false);
            v4 acc = rhs_tile(rb);
#pragma unroll
            for (int d = 1; d <= KB; ++d)
                if (rb - d >= rbs) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-cur.A[d - 1][q], Xw[(u - d + KB) % KB][q], acc, 0, 0, 0);
                }
            const v4 Z = solve_diag(cur.I, acc);
            Xw[u] = Z;
            *reinterpret_cast<v4 *>(zs + (int64_t)rb * 256 + 4 * lane) = Z;   // the wave's own scratch: 32 bytes per lane
            if (fwd_only && c0 + li < K) {                                    // Tb = the last K rows of (L D)^-1 [0; B]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (rb0 + rb) * 16 + 4 * lk + r, aa = row - (np - K);
                    if (aa >= 0) a.Tb[((int64_t)p * K + aa) * K + c0 + li] = Z[r] * a.dinv[cd.row0 + row];   // D^-1 L^-1 [0; B]
                }
            }
        }
    }
    if (fwd_only) return;
    // ---------------- backward: X ----------------
#pragma unroll
    for (int q = 0; q < KB; ++q) Xw[q] = v4{0.0, 0.0, 0.0, 0.0};
    double mi = 0.0, mo = 0.0;
    double *spike = (side == 0 ? a.Wf : (side == 1 ? a.Vf : nullptr));
    // the Z tiles this wave wrote are read back by the same lanes: program order within the wave; the loads below are issued
    // after the stores above, no other wave touches them
    fetch(nxt, NB - 1, true);
    for (int rbb = ((NB - 1) / KB) * KB; rbb >= 0; rbb -= KB) {
#pragma unroll
        for (int u = KB - 1; u >= 0; --u) {
            const int rb = rbb + u;
            if (rb >= NB) continue;
            if (side == 2 && rb < rbs) continue;     // only the last K rows matter (they do not depend on the rows above)
            cur = nxt;
            fetch(nxt, rb - 1, true);
            v4 acc = {0.0, 0.0, 0.0, 0.0};
            if (rb >= rbs) acc = cur.Z;
            if (side == 2) {                         // (D^-1 U) X = [0; I]  <=>  U X = [0; D]: entry (row, col) = U_row,row iff row - (np - K) == col
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (rb0 + rb) * 16 + 4 * lk + r;
                    acc[r] = (row - (np - K) == c0 + li) ? 1.0 / a.dinv[cd.row0 + row] : 0.0;
                }
            }
#pragma unroll
            for (int d = 1; d <= KB; ++d)
                if (rb + d < NB) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-cur.A[d - 1][q], Xw[(u + d) % KB][q], acc, 0, 0, 0);
                }
            const v4 X = solve_diag(cur.I, acc);
            Xw[u] = X;
            // ---- outputs: element r of lane = (row 4 lk + r of the block, column li)
            const int col = c0 + li;
            if (col < K) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (rb0 + rb) * 16 + 4 * lk + r;    // chain-local row
                    const double v = X[r];
                    const int dist = side == 0 ? row : np - 1 - row; // rows from the interface
                    if (dist < K) tips[(int64_t)(side == 0 ? row : row - (np - K)) * K + col] = v;
                    if (spike != nullptr && dist < m) {
                        spike[((int64_t)p * K + col) * m + (side == 0 ? row : row - (np - m))] = v;
                        mi = fmax(mi, fabs(v));
                        if (dist >= m - 32) mo = fmax(mo, fabs(v));
                    }
                }
            }
        }
    }
    if (spike != nullptr) {
        for (int o = 32; o > 0; o >>= 1) { mi = fmax(mi, __shfl_down(mi, o)); mo = fmax(mo, __shfl_down(mo, o)); }
        if (lane == 0) {
            if (mi > __hip_atomic_load(a.absmax_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_max_pos(a.absmax_in, mi);
            if (mo > __hip_atomic_load(a.absmax_edge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_max_pos(a.absmax_edge, mo);
        }
    }
}

// doubles of Z scratch launch_spike_trsm needs
size_t spike_trsm_scratch_doubles(int K, int nchains, int region)
{
    const int nz = K > 128 ? 2 : 1, nwv = ((K + nz - 1) / nz + 15) / 16;
    const int nz4 = K > 128 ? 4 : 1, nwv4 = ((K + nz4 - 1) / nz4 + 15) / 16;            // k_spike_trsm2 deals a wide chain side to four workgroups
    const int waves = nwv * nz > nwv4 * nz4 ? nwv * nz : nwv4 * nz4;
    return (size_t)nchains * 2 * (size_t)waves * (size_t)(region / 16) * 256           // Z tiles per (chain, side, wave)
           + (size_t)nchains * 2 * (size_t)(region / 16) * 512;                        // inverted diagonal tiles [L^-1 | U^-1]
}

hipError_t launch_spike_trsm(double *lu, int K, int m, int region, const ChainDesc *chains, int nchains, const double *band,
                             int64_t ld, double *Wt, double *Vb, double *Wf, double *Vf, double *zscratch, double *absmax_in,
                             double *absmax_edge, hipStream_t st, double *Tb, double *Gb, const double *dinv)
{
    if (nchains <= 0 || K <= 32 || K > 256 || (Tb == nullptr) != (Gb == nullptr) || (Tb != nullptr && dinv == nullptr)) return hipErrorInvalidValue;
    TrsmArgs a;
    a.lv.p = lu; a.lv.ld = 0; a.lv.K = K; a.lv.KB = lu_kb(K); a.lv.ntl = 2 * a.lv.KB + 1;
    a.chains = chains; a.band = band; a.ld = ld;
    a.K = K; a.m = m; a.region = region; a.Wt = Wt; a.Vb = Vb; a.Wf = Wf; a.Vf = Vf; a.zscratch = zscratch;
    a.absmax_in = absmax_in; a.absmax_edge = absmax_edge; a.Tb = Tb; a.Gb = Gb; a.dinv = dinv;
    // K > 128: four workgroups of four waves per chain side (one wave per SIMD: the operands of two steps and the window of
    // 16 solved tiles need ~420 registers)
    const int nz = K > 128 ? 2 : 1, nwv = ((K + nz - 1) / nz + 15) / 16;
    const int nz4 = K > 128 ? 4 : 1, nwv4 = ((K + nz4 - 1) / nz4 + 15) / 16;
    const int waves = nwv * nz > nwv4 * nz4 ? nwv * nz : nwv4 * nz4;
    const int nsides = Tb != nullptr ? 3 : 2;
    double *inv = zscratch + (size_t)nchains * 2 * (size_t)waves * (size_t)(region / 16) * 256;
    hipLaunchKernelGGL(k_trsm_diag_inv, dim3(region / 16, nchains, 2), dim3(64), 0, st, a, inv);
    if (a.lv.KB == 4) hipLaunchKernelGGL(k_spike_trsm2<4>, dim3(nchains, nsides, nz4), dim3(nwv4 * 64), 0, st, a, inv);
    else if (a.lv.KB == 8) hipLaunchKernelGGL(k_spike_trsm2<8>, dim3(nchains, nsides, nz4), dim3(nwv4 * 64), 0, st, a, inv);
    else hipLaunchKernelGGL(k_spike_trsm2<16>, dim3(nchains, nsides, nz4), dim3(nwv4 * 64), 0, st, a, inv);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// interface systems
// ------------------------------------------------------------------------------------------
// work: per interface K x 2K row-major [S | I] -> [I | S^{-1}] by Gauss-Jordan with partial pivoting
__global__ __launch_bounds__(256) void k_iface_setup(int K, const double *Wall, const double *Vall, double *WTall,
                                                     double *VTall, double *STall, double *workall, int *flag)
{
    extern __shared__ double sh[];
    double *rowk = sh;           // 2K
    double *colk = sh + 2 * K;   // K
    __shared__ double rmax[256];
    __shared__ int ridx[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int64_t kk = (int64_t)K * K;
    const double *W = Wall + f * kk, *V = Vall + f * kk;
    double *WT = WTall + f * kk, *VT = VTall + f * kk, *ST = STall + f * kk;
    double *M = workall + f * 2 * kk;
    const int K2 = 2 * K;
    for (int t = tid; t < K * K; t += 256) {
        const int a = t / K, b = t % K;
        double s = (a == b) ? 1.0 : 0.0;
        for (int c = 0; c < K; ++c) s -= W[a * K + c] * V[c * K + b];
        M[a * K2 + b] = s;
        M[a * K2 + K + b] = (a == b) ? 1.0 : 0.0;
        WT[b * K + a] = W[t];
        VT[b * K + a] = V[t];
    }
    __syncthreads();
    int bad = 0;
    for (int k = 0; k < K; ++k) {
        double best = -1.0;
        int bi = k;
        for (int r = k + tid; r < K; r += 256) {
            const double v = fabs(M[r * K2 + k]);
            if (v > best) { best = v; bi = r; }
        }
        rmax[tid] = best; ridx[tid] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                if (rmax[tid + o] > rmax[tid] || (rmax[tid + o] == rmax[tid] && ridx[tid + o] < ridx[tid])) {
                    rmax[tid] = rmax[tid + o]; ridx[tid] = ridx[tid + o];
                }
            }
            __syncthreads();
        }
        const int pr = ridx[0];
        const double pv = rmax[0];
        if (!(pv > 0.0)) { bad = 1; break; }
        // swap rows k and pr, scale pivot row
        for (int cI = tid; cI < K2; cI += 256) {
            const double a = M[pr * K2 + cI], b = M[k * K2 + cI];
            if (pr != k) M[pr * K2 + cI] = b;
            rowk[cI] = a;
        }
        __syncthreads();
        const double inv = 1.0 / rowk[k];
        for (int cI = tid; cI < K2; cI += 256) {
            const double v = rowk[cI] * inv;
            M[k * K2 + cI] = v;
        }
        for (int r = tid; r < K; r += 256) colk[r] = (r == k) ? 0.0 : M[r * K2 + k];
        __syncthreads();
        for (int cI = tid; cI < K2; cI += 256) rowk[cI] *= inv;
        __syncthreads();
        for (int t = tid; t < K * K2; t += 256) {
            const int r = t / K2, cI = t % K2;
            if (r != k) M[t] -= colk[r] * rowk[cI];
        }
        __syncthreads();
    }
    if (bad) { if (tid == 0) flag[f] = 1; return; }
    for (int t = tid; t < K * K; t += 256) {
        const int a = t / K, b = t % K;
        ST[b * K + a] = M[a * K2 + K + b];
    }
}

// The same for K <= 128 with the whole K x K matrix in LDS (row stride K + 1: conflict-free by rows and by columns;
// 132 KiB of the CU's 160 KiB at K = 128): S = I - W V is formed there and inverted IN PLACE by Gauss-Jordan with partial
// pivoting (row swaps recorded, undone as one column permutation when the inverse is written out).  One 1024-thread
// workgroup per interface, lane = column: a block step touches LDS only -- the global-memory version above moves the
// augmented K x 2K matrix through L2 once per pivot (8 ms for 255 interfaces at K = 128, against 0.3 ms here).
// GLOBAL = true (K > 128: the matrix does not fit the CU's LDS): the same in-place algorithm with the K x (K+1) matrix in a
// global work area (it stays in L2: 0.5 MB per interface at K = 256) and only the pivot row/column in LDS -- half the traffic
// of the augmented [S | I] form of k_iface_setup and four times its threads per interface.
template <bool GLOBAL>
__global__ __launch_bounds__(1024) void k_iface_setup_lds(int K, const double *Wall, const double *Vall, double *WTall,
                                                          double *VTall, double *STall, int *flag, double *workall)
{
    extern __shared__ double sh[];
    const int LD = K + 1;
    double *A = GLOBAL ? workall + (size_t)blockIdx.x * K * LD : sh;   // K x LD
    double *rowk = GLOBAL ? sh : sh + (size_t)K * LD;                  // K
    double *colk = rowk + K;            // K
    double *rmax = colk + K;            // 16 (one per wave)
    int *ridx = reinterpret_cast<int *>(rmax + 16);  // 16
    int *perm = ridx + 16;              // K
    int *cidx = perm + K;               // K
    __shared__ int s_pr;
    __shared__ double s_pv;
    const int f = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int64_t kk = (int64_t)K * K;
    const double *W = Wall + f * kk, *V = Vall + f * kk;
    double *WT = WTall + f * kk, *VT = VTall + f * kk, *ST = STall + f * kk;
    const int c = tid % K, r0 = tid / K, rstep = nt / K;   // thread = (column c, rows r0, r0 + rstep, ...)
    const bool active = r0 < rstep;                         // the last nt % K threads have no rows
    // transposes of the inputs (what k_iface_apply streams) and S = I - W V
    for (int t = tid; t < K * K; t += nt) {
        const int a = t / K, b = t % K;
        WT[b * K + a] = W[t];
        VT[b * K + a] = V[t];
    }
    for (int r = active ? r0 : K; r < K; r += rstep) {
        double acc = (r == c) ? 1.0 : 0.0;
        for (int k = 0; k < K; ++k) acc = fma(-W[r * K + k], V[k * K + c], acc);
        A[r * LD + c] = acc;
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
    for (int k = 0; k < K; ++k) {
        // pivot search in column k, rows >= k (ties: smallest row, as the global-memory version)
        double best = -1.0;
        int bi = K;
        for (int r = k + tid; r < K; r += nt) {
            const double v = fabs(A[r * LD + k]);
            if (v > best) { best = v; bi = r; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_down(best, o);
            const int oi = __shfl_down(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { rmax[wv] = best; ridx[wv] = bi; }
        __syncthreads();
        if (tid == 0) {
            double bb = rmax[0];
            int ii = ridx[0];
            for (int q = 1; q < nw; ++q)
                if (rmax[q] > bb || (rmax[q] == bb && ridx[q] < ii)) { bb = rmax[q]; ii = ridx[q]; }
            s_pr = ii; s_pv = bb; perm[k] = ii;
        }
        __syncthreads();
        const int pr = s_pr;
        if (!(s_pv > 0.0)) { if (tid == 0) flag[f] = 1; return; }
        // swap rows k and pr; pivot row scaled, pivot column kept aside
        if (tid < K) {
            const double a = A[pr * LD + tid], b = A[k * LD + tid];
            if (pr != k) A[pr * LD + tid] = b;
            rowk[tid] = a;
        }
        __syncthreads();
        const double inv = 1.0 / rowk[k];
        if (tid < K) colk[tid] = (tid == k) ? 0.0 : A[tid * LD + k];
        __syncthreads();
        const double rk = (c == k) ? inv : rowk[c] * inv;   // row k of the in-place form: A[k][k] <- 1/pivot
        if (r0 == 0) A[k * LD + c] = rk;
        // eight rows per thread in flight: as a plain loop every row waited for its own load (K > 128: the matrix lives in a
        // global work area -- 64 dependent L2 round trips per pivot step and thread, 61 us per step)
        for (int rb = active ? r0 : K; rb < K; rb += 8 * rstep) {
            double old[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + u * rstep;
                old[u] = (r < K) ? A[r * LD + c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + u * rstep;
                if (r < K && r != k) A[r * LD + c] = fma(-colk[r], rk, (c == k) ? 0.0 : old[u]);
            }
        }
        __syncthreads();
    }
    // undo the row swaps: column j of the inverse is column cidx[j] of the in-place result
    if (tid == 0) {
        for (int j = 0; j < K; ++j) cidx[j] = j;
        for (int k = K - 1; k >= 0; --k) { const int p2 = perm[k], t = cidx[k]; cidx[k] = cidx[p2]; cidx[p2] = t; }
    }
    __syncthreads();
    for (int t = tid; t < K * K; t += nt) {
        const int a = t % K, b = t / K;
        ST[b * K + a] = A[a * LD + cidx[b]];
    }
}

static hipError_t iface_setup_blocked(int K, int nif, const double *W, const double *V, double *WT, double *VT, double *ST,
                                      double *work, int *flag, hipStream_t st);

// smallest K that goes through 2 x 2 blocks (iface_setup_blocked): measurement knob SPIKE_IFACE_BLOCKED_KMIN, default 129
static int iface_blocked_kmin()
{
    const char *e = getenv("SPIKE_IFACE_BLOCKED_KMIN");
    const int v = e ? atoi(e) : 129;
    return v < 34 ? 34 : v;
}
#define IFACE_BLOCKED_KMIN iface_blocked_kmin()

hipError_t launch_iface_setup(int K, int nif, const double *W, const double *V, double *WT, double *VT, double *ST,
                              double *work, int *flag, hipStream_t st)
{
    if (nif <= 0 || K <= 0) return hipSuccess;
    const bool blocked = work != nullptr && K >= IFACE_BLOCKED_KMIN && K <= 256 && getenv("SPIKE_IFACE_UNBLOCKED") == nullptr;
    if (K <= 128 && !blocked) {
        const int nt = K <= 8 ? 64 : (K <= 32 ? 256 : 1024);
        const size_t lds = ((size_t)K * (K + 1) + 2 * K + 16) * sizeof(double) + (16 + 2 * (size_t)K) * sizeof(int);
        // beyond 48 KiB the kernel needs its dynamic-LDS limit raised (cheap, so simply every time: no process-global state)
        bool ok = lds <= 48 * 1024 || hipFuncSetAttribute(reinterpret_cast<const void *>(k_iface_setup_lds<false>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        if (!ok) (void)hipGetLastError();   // this device cannot give one workgroup that much LDS: global-memory version
        if (ok) {
            hipLaunchKernelGGL(k_iface_setup_lds<false>, dim3(nif), dim3(nt), lds, st, K, W, V, WT, VT, ST, flag, (double *)nullptr);
            return hipGetLastError();
        }
    }
    if (blocked) {
        // 2 x 2 blocks through the LDS kernel (the work area holds iface_setup_work_doubles); a zero pivot inside a diagonal block
        // or a failed verification of any system sends the whole batch to the unblocked kernel below
        hipError_t e = iface_setup_blocked(K, nif, W, V, WT, VT, ST, work, flag, st);
        if (e != hipSuccess) return e;
        std::vector<int> fl((size_t)nif, 0);
        e = hipMemcpyAsync(fl.data(), flag, sizeof(int) * nif, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
        bool ok = true;
        for (int i = 0; i < nif; ++i) ok = ok && fl[(size_t)i] == 0;
        if (ok) return hipSuccess;
        e = hipMemsetAsync(flag, 0, sizeof(int) * nif, st);
        if (e != hipSuccess) return e;
    }
    if (work != nullptr) {   // the caller's work area holds 2 K^2 doubles per interface >= K (K+1)
        const size_t lds = ((size_t)2 * K + 16) * sizeof(double) + (16 + 2 * (size_t)K) * sizeof(int);
        hipLaunchKernelGGL(k_iface_setup_lds<true>, dim3(nif), dim3(1024), lds, st, K, W, V, WT, VT, ST, flag, work);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_iface_setup, dim3(nif), dim3(256), (size_t)3 * K * sizeof(double), st, K, W, V, WT, VT, ST,
                       work, flag);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// 128 < K <= 256 (round 3): the K x K interface systems through 2 x 2 blocks.  The in-place Gauss-Jordan above keeps a matrix of
// K <= 128 in LDS (1.4 ms for 255 of them); at K = 256 the matrix lived in L2 and 127 inverses took 11-12 ms -- twice per setup
// since the seams have systems of their own.  With S = [A B; C D] split at h1 = ceil(K/2):
//     Ai = A^-1,  X = Ai B,  Y = C Ai,  Sc = D - C X,  Sci = Sc^-1,  Z = -Sci Y,
//     S^-1 = [Ai - X Z, -X Sci; Z, Sci]
// -- two inverses of size <= 128 (the LDS kernel, fed with W' = I - A, V' = I so that its S is A) and six products on MFMA.
// Pivoting is partial INSIDE the two diagonal blocks only, so the result is verified: R = S S^-1 must be the identity to 1e-9
// in every entry, else (or on a zero pivot) the caller falls back to the unblocked kernel for the whole batch.
// ------------------------------------------------------------------------------------------
// C (M x N, ldc) = alpha A (M x Kc, lda) B (Kc x N, ldb) + beta Cin (ldcin; null: 0); batch strides in doubles
struct GemmArgs {
    const double *A, *B, *Cin;
    double *C;
    int M, N, Kc, lda, ldb, ldc, ldcin;
    int64_t sa, sb, sc, scin;
    double alpha, beta;
};
__global__ __launch_bounds__(256) void k_gemm_gen(GemmArgs g)
{
    typedef double v4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int i0 = blockIdx.y * 16, j0 = (blockIdx.x * 4 + w) * 16;
    if (j0 >= g.N) return;
    const double *a = g.A + (int64_t)blockIdx.z * g.sa, *b = g.B + (int64_t)blockIdx.z * g.sb;
    v4 acc = {0.0, 0.0, 0.0, 0.0};
    const int ai = i0 + li, bj = j0 + li;
    for (int k0 = 0; k0 < g.Kc; k0 += 4) {
        const int k = k0 + lk;
        const double av = (ai < g.M && k < g.Kc) ? a[(int64_t)ai * g.lda + k] : 0.0;
        const double bv = (bj < g.N && k < g.Kc) ? b[(int64_t)k * g.ldb + bj] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    double *c = g.C + (int64_t)blockIdx.z * g.sc;
    const double *ci = g.Cin ? g.Cin + (int64_t)blockIdx.z * g.scin : nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = i0 + lk + 4 * r, col = j0 + li;
        if (row < g.M && col < g.N) {
            double v = g.alpha * acc[r];
            if (ci) v += g.beta * ci[(int64_t)row * g.ldcin + col];
            c[(int64_t)row * g.ldc + col] = v;
        }
    }
}
static hipError_t gemm_gen(int count, int M, int N, int Kc, double alpha, const double *A, int lda, int64_t sa, const double *B, int ldb,
                           int64_t sb, double beta, const double *Cin, int ldcin, int64_t scin, double *C, int ldc, int64_t sc,
                           hipStream_t st)
{
    if (count <= 0 || M <= 0 || N <= 0) return hipSuccess;
    GemmArgs g{A, B, Cin, C, M, N, Kc, lda, ldb, ldc, ldcin, sa, sb, sc, scin, alpha, beta};
    hipLaunchKernelGGL(k_gemm_gen, dim3((N + 63) / 64, (M + 15) / 16, count), dim3(256), 0, st, g);
    return hipGetLastError();
}
// dst (R x Cc, ldd) = op(src): mode 0 copy, 1 transpose (src is Cc x R), 2 identity minus src, 3 identity
__global__ void k_block_op(int mode, int R, int Cc, const double *src, int lds_, int64_t ss, double *dst, int ldd, int64_t sd)
{
    const double *s0 = src ? src + (int64_t)blockIdx.y * ss : nullptr;
    double *d0 = dst + (int64_t)blockIdx.y * sd;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < R * Cc; t += gridDim.x * blockDim.x) {
        const int r = t / Cc, c = t % Cc;
        double v;
        if (mode == 0) v = s0[(int64_t)r * lds_ + c];
        else if (mode == 1) v = s0[(int64_t)c * lds_ + r];
        else if (mode == 2) v = (r == c ? 1.0 : 0.0) - s0[(int64_t)r * lds_ + c];
        else v = r == c ? 1.0 : 0.0;
        d0[(int64_t)r * ldd + c] = v;
    }
}
static hipError_t block_op(int mode, int count, int R, int Cc, const double *src, int lds_, int64_t ss, double *dst, int ldd, int64_t sd,
                           hipStream_t st)
{
    if (count <= 0 || R <= 0 || Cc <= 0) return hipSuccess;
    int gx = (R * Cc + 255) / 256;
    if (gx > 32) gx = 32;
    hipLaunchKernelGGL(k_block_op, dim3(gx, count), dim3(256), 0, st, mode, R, Cc, src, lds_, ss, dst, ldd, sd);
    return hipGetLastError();
}
// flag[f] = 1 when some entry of R_f (K x K) differs from the identity by more than tol (or is not a number)
__global__ void k_check_identity(int K, const double *R, double tol, int *flag)
{
    const double *r0 = R + (int64_t)blockIdx.y * K * K;
    bool bad = false;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < K * K; t += gridDim.x * blockDim.x) {
        const double d = fabs(r0[t] - ((t / K == t % K) ? 1.0 : 0.0));
        if (!(d <= tol)) bad = true;
    }
    if (bad) atomicExch(&flag[blockIdx.y], 1);
}

size_t iface_setup_work_doubles(int K, int nif)
{
    const size_t kk = (size_t)K * K;
    if (K < IFACE_BLOCKED_KMIN || K > 256) return (size_t)nif * 2 * kk;   // the unblocked kernels: K (K + 1) per interface
    return (size_t)nif * 7 * kk;                             // S, S^-1, R + the half-size pieces (12 x <= kk/4 ... rounded up)
}

static hipError_t iface_setup_blocked(int K, int nif, const double *W, const double *V, double *WT, double *VT, double *ST,
                                      double *work, int *flag, hipStream_t st)
{
    const int h1 = (K + 1) / 2, h2 = K - h1;
    const int64_t kk = (int64_t)K * K, q1 = (int64_t)h1 * h1, q2 = (int64_t)h2 * h2, q12 = (int64_t)h1 * h2;
    double *S = work, *Si = S + nif * kk, *R = Si + nif * kk;
    double *p = R + nif * kk;
    double *Wb = p; p += nif * q1;      // I - A, later I - Sc (sized for the larger block)
    double *Ib = p; p += nif * q1;      // identities
    double *T1 = p; p += nif * q1;      // scratch transposes the LDS kernel writes (its WT / VT outputs)
    double *T2 = p; p += nif * q1;
    double *ATi = p; p += nif * q1;     // (A^-1)^T, then (Sc^-1)^T
    double *Ai = p; p += nif * q1;
    double *Sci = p; p += nif * q2;
    double *X = p; p += nif * q12;
    double *Y = p; p += nif * q12;
    double *Sc = p; p += nif * q2;
    hipError_t e;
#define BCHK(x) do { if ((e = (x)) != hipSuccess) return e; } while (0)
    // S = I - W V; the transposes the apply kernels stream
    BCHK(block_op(3, nif, K, K, nullptr, 0, 0, Si, K, kk, st));                                    // Si = I (used as Cin)
    BCHK(gemm_gen(nif, K, K, K, -1.0, W, K, kk, V, K, kk, 1.0, Si, K, kk, S, K, kk, st));
    BCHK(block_op(1, nif, K, K, W, K, kk, WT, K, kk, st));
    BCHK(block_op(1, nif, K, K, V, K, kk, VT, K, kk, st));
    auto invert = [&](int h, const double *blk, int64_t q, double *outT) -> hipError_t {          // outT = (blk^-1)^T, h x h contiguous
        if ((e = block_op(2, nif, h, h, blk, K, kk, Wb, h, q, st)) != hipSuccess) return e;       // W' = I - blk
        if ((e = block_op(3, nif, h, h, nullptr, 0, 0, Ib, h, q, st)) != hipSuccess) return e;    // V' = I
        const int nt = h <= 8 ? 64 : (h <= 32 ? 256 : 1024);
        const size_t lds = ((size_t)h * (h + 1) + 2 * h + 16) * sizeof(double) + (16 + 2 * (size_t)h) * sizeof(int);
        if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(k_iface_setup_lds<false>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_iface_setup_lds<false>, dim3(nif), dim3(nt), lds, st, h, Wb, Ib, T1, T2, outT, flag, (double *)nullptr);
        return hipGetLastError();
    };
    // Ai = A^-1
    BCHK(invert(h1, S, q1, ATi));
    BCHK(block_op(1, nif, h1, h1, ATi, h1, q1, Ai, h1, q1, st));
    // X = Ai B (h1 x h2), Y = C Ai (h2 x h1), Sc = D - C X
    BCHK(gemm_gen(nif, h1, h2, h1, 1.0, Ai, h1, q1, S + h1, K, kk, 0.0, nullptr, 0, 0, X, h2, q12, st));
    BCHK(gemm_gen(nif, h2, h1, h1, 1.0, S + (int64_t)h1 * K, K, kk, Ai, h1, q1, 0.0, nullptr, 0, 0, Y, h1, q12, st));
    BCHK(gemm_gen(nif, h2, h2, h1, -1.0, S + (int64_t)h1 * K, K, kk, X, h2, q12, 1.0, S + (int64_t)h1 * K + h1, K, kk, Sc, h2, q2, st));
    // Sci = Sc^-1 (Sc is contiguous h2 x h2: give it to the inverter as a block with leading dimension h2)
    {
        if ((e = block_op(2, nif, h2, h2, Sc, h2, q2, Wb, h2, q2, st)) != hipSuccess) return e;
        if ((e = block_op(3, nif, h2, h2, nullptr, 0, 0, Ib, h2, q2, st)) != hipSuccess) return e;
        const int nt = h2 <= 8 ? 64 : (h2 <= 32 ? 256 : 1024);
        const size_t lds = ((size_t)h2 * (h2 + 1) + 2 * h2 + 16) * sizeof(double) + (16 + 2 * (size_t)h2) * sizeof(int);
        if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void *>(k_iface_setup_lds<false>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL(k_iface_setup_lds<false>, dim3(nif), dim3(nt), lds, st, h2, Wb, Ib, T1, T2, ATi, flag, (double *)nullptr);
        BCHK(hipGetLastError());
    }
    BCHK(block_op(1, nif, h2, h2, ATi, h2, q2, Sci, h2, q2, st));
    // S^-1 = [Ai - X Z, -X Sci; Z, Sci],  Z = -Sci Y
    BCHK(gemm_gen(nif, h2, h1, h2, -1.0, Sci, h2, q2, Y, h1, q12, 0.0, nullptr, 0, 0, Si + (int64_t)h1 * K, K, kk, st));                  // Z -> bottom left
    BCHK(gemm_gen(nif, h1, h1, h2, -1.0, X, h2, q12, Si + (int64_t)h1 * K, K, kk, 1.0, Ai, h1, q1, Si, K, kk, st));                       // top left
    BCHK(gemm_gen(nif, h1, h2, h2, -1.0, X, h2, q12, Sci, h2, q2, 0.0, nullptr, 0, 0, Si + h1, K, kk, st));                              // top right
    BCHK(block_op(0, nif, h2, h2, Sci, h2, q2, Si + (int64_t)h1 * K + h1, K, kk, st));                                                   // bottom right
    // verify, then hand out the column-major form
    BCHK(gemm_gen(nif, K, K, K, 1.0, S, K, kk, Si, K, kk, 0.0, nullptr, 0, 0, R, K, kk, st));
    hipLaunchKernelGGL(k_check_identity, dim3(16, nif), dim3(256), 0, st, K, R, 1e-9, flag);
    BCHK(hipGetLastError());
    BCHK(block_op(1, nif, K, K, Si, K, kk, ST, K, kk, st));
#undef BCHK
    return hipSuccess;
}

// y[a] = sum_c MT[c*K + a] * x[c];  IFT threads = nparts groups of KA lanes, partial sums via LDS.
// 1024 threads per interface: the three dependent mat-vecs are latency-bound, so the only lever is loads in flight
// (K/nparts = 16 sequential loads per thread at K = 128 instead of 64 with 256 threads).
constexpr int IFT = 1024;  // upper bound; narrow bands launch fewer threads (blockDim.x is what the kernel uses)
__device__ __forceinline__ double iface_matvec(const double *MT, const double *xs, int K, int KA, int nparts, int a,
                                               int part, double *redb)
{
    double acc = 0.0;
    if (a < K && MT != nullptr)
        for (int c = part; c < K; c += nparts) acc = fma(MT[(int64_t)c * K + a], xs[c], acc);
    redb[threadIdx.x] = acc;
    __syncthreads();
    double s = 0.0;
    for (int q = 0; q < nparts; ++q) s += redb[q * KA + a];
    __syncthreads();
    return s;
}

// same mat-vec from matrix entries already in registers (mr[j] = MT[(part + j nparts) K + a], j < PRE)
template <int PRE>
__device__ __forceinline__ double iface_matvec_regs(const double (&mr)[PRE], const double *xs, int K, int KA, int nparts, int a,
                                                    int part, double *redb)
{
    double acc = 0.0;
    if (a < K) {
#pragma unroll
        for (int j = 0; j < PRE; ++j) {
            const int c = part + j * nparts;
            if (c < K) acc = fma(mr[j], xs[c], acc);
        }
    }
    redb[threadIdx.x] = acc;
    __syncthreads();
    double s = 0.0;
    for (int q = 0; q < nparts; ++q) s += redb[q * KA + a];
    __syncthreads();
    return s;
}

// PRE > 0: a thread's slices of W^T, S^-T and V^T (at most PRE entries each) are requested up front -- the three mat-vecs
// depend on each other through the VECTOR only, so their matrix loads need not wait for one another: one memory latency
// instead of three (K = 128: 26 -> ~10 us for 255 interfaces).  Same multiply-adds in the same order: identical bits.
template <int PRE>
__global__ __launch_bounds__(IFT) void k_iface_apply(int K, const IfaceDesc *ifs, const double *g)
{
    extern __shared__ double sh[];
    double *gb = sh, *gt = sh + K, *v1 = sh + 2 * K, *v2 = sh + 3 * K;
    __shared__ double redb[IFT];
    const IfaceDesc d = ifs[blockIdx.x];
    const int tid = threadIdx.x;
    int KA = 1;
    while (KA < K) KA <<= 1;
    const int nparts = (int)blockDim.x / KA;
    const int a = tid % KA, part = tid / KA;
    constexpr int PR = PRE > 0 ? PRE : 1;
    double mw[PR], ms[PR], mv[PR];
    if (PRE > 0) {
#pragma unroll
        for (int j = 0; j < PR; ++j) {
            const int c = part + j * nparts;
            const bool on = a < K && c < K;
            const int64_t off = on ? (int64_t)c * K + a : 0;       // clamped: unconditional loads
            mw[j] = d.WT[off]; ms[j] = d.ST[off]; mv[j] = d.VT[off];
        }
    }
    const double *pgb = d.gb != nullptr ? d.gb : g + d.gb_off, *pgt = d.gt != nullptr ? d.gt : g + d.gt_off;
    for (int t = tid; t < K; t += (int)blockDim.x) { gb[t] = pgb[t]; gt[t] = pgt[t]; }
    __syncthreads();
    // t = gt - W gb
    double s = PRE > 0 ? iface_matvec_regs<PR>(mw, gb, K, KA, nparts, a, part, redb) : iface_matvec(d.WT, gb, K, KA, nparts, a, part, redb);
    if (part == 0 && a < K) v1[a] = gt[a] - s;
    __syncthreads();
    // xt = S^{-1} t
    s = PRE > 0 ? iface_matvec_regs<PR>(ms, v1, K, KA, nparts, a, part, redb) : iface_matvec(d.ST, v1, K, KA, nparts, a, part, redb);
    if (part == 0 && a < K) v2[a] = s;  // v2 = xt
    __syncthreads();
    // xb = gb - V xt
    s = PRE > 0 ? iface_matvec_regs<PR>(mv, v2, K, KA, nparts, a, part, redb) : iface_matvec(d.VT, v2, K, KA, nparts, a, part, redb);
    if (part == 0 && a < K) v1[a] = gb[a] - s;  // v1 = xb
    __syncthreads();
    if (part == 0 && a < K) {
        if (d.xb_out != nullptr) d.xb_out[a] = v1[a];
        if (d.xt_out != nullptr) d.xt_out[a] = v2[a];
    }
    if (d.corr_top != nullptr) {
        s = iface_matvec(d.CT, v1, K, KA, nparts, a, part, redb);
        if (part == 0 && a < K) d.corr_top[a] = s;
    }
    if (d.corr_bot != nullptr) {
        s = iface_matvec(d.BT, v2, K, KA, nparts, a, part, redb);
        if (part == 0 && a < K) d.corr_bot[a] = s;
    }
}

hipError_t launch_iface_apply(int K, int nif, const IfaceDesc *ifs, const double *g, hipStream_t st)
{
    if (nif <= 0 || K <= 0) return hipSuccess;
    int nt = 64;
    while (nt < IFT && nt < 8 * K) nt <<= 1;  // K = 128 -> 1024 threads, K <= 8 -> one wave
    int KA = 1;
    while (KA < K) KA <<= 1;
    const int per = (K + nt / KA - 1) / (nt / KA);        // matrix entries per thread and mat-vec
    if (per <= 8) hipLaunchKernelGGL(k_iface_apply<8>, dim3(nif), dim3(nt), (size_t)4 * K * sizeof(double), st, K, ifs, g);
    else if (per <= 16) hipLaunchKernelGGL(k_iface_apply<16>, dim3(nif), dim3(nt), (size_t)4 * K * sizeof(double), st, K, ifs, g);
    else hipLaunchKernelGGL(k_iface_apply<0>, dim3(nif), dim3(nt), (size_t)4 * K * sizeof(double), st, K, ifs, g);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// One-stage interface solve (round 3).  k_iface_apply runs three DEPENDENT K x K mat-vecs in one workgroup per interface: a
// workgroup cannot pull more than ~50 GB/s through its CU, so 3 K^2 8 bytes take >= 8 us at K = 128 however many loads are
// in flight -- on the critical path of every apply twice (seam, outer interfaces) and a third time behind the tip exchange
// of a multi-GPU job.  Here the three stages are multiplied out at setup:
//     [x_b; x_t] = M [g_b; g_t],   M = [[I + V S^-1 W, -V S^-1], [-S^-1 W, S^-1]]     (2K x 2K)
// The rows of M are independent, so an interface is dealt to 2K/64 workgroups (64 outputs each): 4/3 of the bytes, spread
// over four CUs, one stage.  MT = M^T row-major (MT[c 2K + a] = M[a][c]): consecutive lanes read consecutive addresses.
// Desc: WT = MT; gb / gt (+ offsets) as k_iface_apply; xb_out / xt_out.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_iface_apply_m(int K, const IfaceDesc *ifs, const double *g)
{
    extern __shared__ double shm[];
    double *g2 = shm;                 // [g_b | g_t], 2K
    double *red = shm + 2 * K;        // 256 partial sums
    const IfaceDesc d = ifs[blockIdx.x];
    const int tid = threadIdx.x, K2 = 2 * K;
    const int nout = K2 < 64 ? K2 : 64;               // outputs of this workgroup
    const int a0 = blockIdx.y * 64;
    const double *pgb = d.gb != nullptr ? d.gb : g + d.gb_off, *pgt = d.gt != nullptr ? d.gt : g + d.gt_off;
    for (int t = tid; t < K2; t += 256) g2[t] = t < K ? pgb[t] : pgt[t - K];
    __syncthreads();
    const int al = tid % nout, part = tid / nout, nparts = 256 / nout;
    const int a = a0 + al;
    double acc = 0.0;
    if (a < K2 && part < nparts) {
        const double *col = d.WT + a;
#pragma unroll 8
        for (int c = part; c < K2; c += nparts) acc = fma(col[(int64_t)c * K2], g2[c], acc);
    }
    red[tid] = (part < nparts) ? acc : 0.0;
    __syncthreads();
    if (tid < nout && a0 + tid < K2) {
        double s = 0.0;
        for (int q = 0; q < nparts; ++q) s += red[q * nout + tid];
        const int o = a0 + tid;
        if (o < K) { if (d.xb_out != nullptr) d.xb_out[o] = s; }
        else if (d.xt_out != nullptr) d.xt_out[o - K] = s;
    }
}

hipError_t launch_iface_apply_m(int K, int nif, const IfaceDesc *ifs, const double *g, hipStream_t st)
{
    if (nif <= 0 || K <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_iface_apply_m, dim3(nif, (2 * K + 63) / 64), dim3(256), (size_t)(2 * K + 256) * sizeof(double), st, K, ifs, g);
    return hipGetLastError();
}

// batched C_i = A_i B_i for row-major K x K matrices on v_mfma_f64_16x16x4 (setup: the products that make up M).  A wave owns
// one 16 x 16 tile of C; lane maps: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], C/D: col = l&15, row = (l>>4) + 4 reg.
__global__ __launch_bounds__(256) void k_gemm_kk(int K, const double *A, int64_t sa, const double *B, int64_t sb, double *C, int64_t sc)
{
    typedef double v4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int i0 = blockIdx.y * 16, j0 = (blockIdx.x * 4 + w) * 16;
    if (j0 >= K) return;
    const double *a = A + (int64_t)blockIdx.z * sa, *b = B + (int64_t)blockIdx.z * sb;
    double *c = C + (int64_t)blockIdx.z * sc;
    v4 acc = {0.0, 0.0, 0.0, 0.0};
    const int ai = i0 + li, bj = j0 + li;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + lk;
        const double av = (ai < K && k < K) ? a[(int64_t)ai * K + k] : 0.0;
        const double bv = (bj < K && k < K) ? b[(int64_t)k * K + bj] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = i0 + lk + 4 * r, col = j0 + li;
        if (row < K && col < K) c[(int64_t)row * K + col] = acc[r];
    }
}

hipError_t launch_gemm_kk(int K, int count, const double *A, int64_t sa, const double *B, int64_t sb, double *C, int64_t sc, hipStream_t st)
{
    if (count <= 0 || K <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_gemm_kk, dim3((K + 63) / 64, (K + 15) / 16, count), dim3(256), 0, st, K, A, sa, B, sb, C, sc);
    return hipGetLastError();
}

// MT (2K x 2K row-major, = M^T) from ST (S^-T, i.e. the column-major S^-1 of k_iface_setup) and the products
// P1T = WT ST = (S^-1 W)^T, P2T = ST VT = (V S^-1)^T, P3T = P1T VT = (V S^-1 W)^T
__global__ __launch_bounds__(256) void k_build_iface_m(int K, const double *ST, const double *P1T, const double *P2T, const double *P3T,
                                                       double *MT)
{
    const int64_t kk = (int64_t)K * K, f = blockIdx.y;
    const int K2 = 2 * K;
    const double *st = ST + f * kk, *p1 = P1T + f * kk, *p2 = P2T + f * kk, *p3 = P3T + f * kk;
    double *m = MT + f * 4 * kk;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < 4 * kk; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t / K2), a = (int)(t % K2);
        double v;
        if (c < K) v = a < K ? ((a == c ? 1.0 : 0.0) + p3[(int64_t)c * K + a]) : -p1[(int64_t)c * K + (a - K)];
        else v = a < K ? -p2[(int64_t)(c - K) * K + a] : st[(int64_t)(c - K) * K + (a - K)];
        m[t] = v;
    }
}

hipError_t launch_build_iface_m(int K, int nif, const double *ST, const double *P1T, const double *P2T, const double *P3T, double *MT,
                                hipStream_t st)
{
    if (nif <= 0 || K <= 0) return hipSuccess;
    int gx = (int)((4 * (int64_t)K * K + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_build_iface_m, dim3(gx, nif), dim3(256), 0, st, K, ST, P1T, P2T, P3T, MT);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// stored spikes.  For diagonally dominant bands the spikes V_j = A_j^{-1}[0;B_j], W_j = A_j^{-1}[C_j;0] decay
// away from the interface; only their m significant rows are kept (column-major per partition:
// out[(p*K + col)*m + r]), so the second pass of the coupled variant becomes a small dense correction.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_max_pos(double *addr, double v)
{
    atomicMax((unsigned long long *)addr, (unsigned long long)__double_as_longlong(v));
}

// which = 0: W (rows [0,m) of each chain); which = 1: V (rows [nrows-m, nrows)).  absmax_in = peak magnitude inside the
// window, absmax_edge = peak over the 32 (windows shorter than 128 rows: 8) window rows farthest from the interface (must
// be ~0 for a decayed spike).
__global__ __launch_bounds__(256) void k_spike_gather(const double *sol, int K, int m, const ChainDesc *chains, int which,
                                                      int col, double *out, double *absmax_in, double *absmax_edge, int64_t ldr)
{
    const int p = blockIdx.y;
    const ChainDesc cd = chains[p];
    col += blockIdx.z;
    sol += blockIdx.z * ldr;
    double mi = 0.0, mo = 0.0;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < m; w += gridDim.x * blockDim.x) {
        const int r = which == 0 ? w : cd.nrows - m + w;
        const double v = sol[cd.row0 + r];
        out[((int64_t)p * K + col) * m + w] = v;
        mi = fmax(mi, fabs(v));
        const int dist = which == 0 ? w : m - 1 - w;  // distance from the interface, in rows
        if (dist >= m - (m >= 128 ? 32 : 8)) mo = fmax(mo, fabs(v));   // (the tight windows of K = 1 have 8 rows of margin)
    }
    for (int o = 32; o > 0; o >>= 1) { mi = fmax(mi, __shfl_down(mi, o)); mo = fmax(mo, __shfl_down(mo, o)); }
    // thousands of waves share two words: only touch them when the running maximum would actually grow
    if ((threadIdx.x & 63) == 0) {
        if (mi > __hip_atomic_load(absmax_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_max_pos(absmax_in, mi);
        if (mo > __hip_atomic_load(absmax_edge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_max_pos(absmax_edge, mo);
    }
}

hipError_t launch_spike_gather(const double *sol, int K, int m, const ChainDesc *chains, int nchains, int which, int col,
                               double *out, double *absmax_in, double *absmax_out, hipStream_t st, int ncols, int64_t ldr)
{
    if (nchains <= 0 || K <= 0 || m <= 0 || ncols <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_spike_gather, dim3((m + 255) / 256 < 8 ? (m + 255) / 256 : 8, nchains, ncols), dim3(256), 0, st, sol, K, m,
                       chains, which, col, out, absmax_in, absmax_out, ldr);
    return hipGetLastError();
}

// extent[0] = max over chains of the distance (in rows, from the interface) of the farthest entry with |v| > tol_abs
__global__ __launch_bounds__(256) void k_spike_extent(const double *sol, const ChainDesc *chains, int which, double tol_abs,
                                                      int *extent)
{
    const ChainDesc cd = chains[blockIdx.y];
    int far = 0;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < cd.nrows; r += gridDim.x * blockDim.x) {
        const int dist = which == 0 ? r + 1 : cd.nrows - r;
        if (fabs(sol[cd.row0 + r]) > tol_abs && dist > far) far = dist;
    }
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_down(far, o); far = t > far ? t : far; }
    if ((threadIdx.x & 63) == 0 && far > 0) atomicMax(extent, far);
}

hipError_t launch_spike_extent(const double *sol, const ChainDesc *chains, int nchains, int which, double tol_abs,
                               int *extent, hipStream_t st)
{
    if (nchains <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_spike_extent, dim3(8, nchains), dim3(256), 0, st, sol, chains, which, tol_abs, extent);
    return hipGetLastError();
}

// x[top m rows of chain p]    -= W_p x_b(p-1)   (xb slot p;   slot 0   = the previous rank's last partition)
// x[bottom m rows of chain p] -= V_p x_t(p+1)   (xt slot p+2; slot P+1 = the next rank's first partition)
// lane = row (coalesced column-major spike reads), K sequential FMAs per lane, tip vectors from LDS.
// mode 0: every (chain, end); 1: all but the two rank-boundary ends (top of chain 0, bottom of the last chain), whose tip
// solutions come from the exchange; 2: exactly those two (launched on the exchange stream once they are known).
// TWISTED: a chain has ONE window, the m rows at its chain-local top (the outer end of its partition), spike Wf_p in chain-
// local orientation.  A vdir = +1 chain (top half) takes x_b of the partition above as ever; a vdir = -1 chain (bottom half,
// stored flipped) is the partition's natural V spike with rows and columns reversed: it takes x_t of the partition below
// with its entries REVERSED, and its rows run backwards through x.
//
// MIXED PRECISION (round 3).  A spike decays by orders of magnitude over its window; where every entry is below 2^-28 of the
// spikes' peak, an fp32 copy is as good as the fp64 one -- its rounding error, 2^-24 relative, is 2^-52 of the peak, i.e. what
// fp64 rounding does to the LARGE entries anyway.  So the window is stored in two parts: rows [0, m1) (from the interface)
// in fp64, rows [m1, m) in fp32 (Wf32 / Vf32, column-major K x (m - m1) per chain); setup measures m1 and verifies the bound.
// The arithmetic is fp64 throughout (the fp32 entries are widened on load).  grid.x = blocks of the fp64 part (512 rows each:
// two rows per lane, 16-byte loads) followed by blocks of the fp32 part (1024 rows each: four rows per lane, 16-byte loads).
struct SpikeStore {
    const double *Wf, *Vf;     // fp64 parts, column-major K x m1 per chain
    const float *Wf32, *Vf32;  // fp32 parts, column-major K x (m - m1) per chain (null when m1 == m)
    int m, m1;
};

template <bool TWISTED>
__global__ __launch_bounds__(256) void k_spike_correct(int K, SpikeStore sp, const ChainDesc *chains, int nchains, const double *xb,
                                                       const double *xt, double *x, int mode, int nb64)
{
    extern __shared__ double tip[];
    int p = blockIdx.y, which = blockIdx.z;
    if (mode == 2) { which = blockIdx.y; p = which == 0 ? 0 : nchains - 1; }
    else if (mode == 1 && ((p == 0 && (TWISTED || which == 0)) || (p == nchains - 1 && (TWISTED || which == 1)))) return;
    const ChainDesc cd = chains[p];
    const bool up = TWISTED && cd.vdir < 0;
    if (TWISTED) which = 0;
    const int m = sp.m, m1 = sp.m1;
    const double *src = (which == 0 && !up) ? xb + (int64_t)p * K : xt + (int64_t)(p + 2) * K;
    for (int c = threadIdx.x; c < K; c += blockDim.x) tip[c] = up ? src[K - 1 - c] : src[c];
    __syncthreads();
    // window row w (0 = at the interface... for V windows the rows are stored top to bottom, the interface at the END): the
    // stored row index r in [0, m) maps to chain-local row (which == 0 ? r : nrows - m + r); the fp64 part holds the m1 rows
    // NEXT TO THE INTERFACE, i.e. stored rows [0, m1) of a W window and [m - m1, m) of a V window
    const bool f32 = (int)blockIdx.x >= nb64;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int r, nr;           // first stored row of this lane, rows it owns
    if (!f32) {
        const int q = 2 * (blockIdx.x * blockDim.x + threadIdx.x);            // row pair inside the fp64 part
        if (q >= m1) return;
        nr = q + 1 < m1 ? 2 : 1;
        r = which == 0 ? q : (m - m1) + q;
        const double *S = (which == 0 ? sp.Wf : sp.Vf) + (int64_t)p * K * m1 + q;
        if (nr == 2 && (m1 & 1) == 0) {
#pragma unroll 8
            for (int c = 0; c < K; ++c) {
                const d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(S + (int64_t)c * m1));
                acc[0] = fma(v.x, tip[c], acc[0]);
                acc[1] = fma(v.y, tip[c], acc[1]);
            }
        } else {
            for (int c = 0; c < K; ++c) {
                acc[0] = fma(S[(int64_t)c * m1], tip[c], acc[0]);
                if (nr == 2) acc[1] = fma(S[(int64_t)c * m1 + 1], tip[c], acc[1]);
            }
        }
    } else {
        const int m2 = m - m1;                                                // a multiple of 64 (setup)
        const int q = 4 * ((blockIdx.x - nb64) * blockDim.x + threadIdx.x);   // row quad inside the fp32 part
        if (q >= m2) return;
        nr = 4;
        r = which == 0 ? m1 + q : q;
        typedef float f4 __attribute__((ext_vector_type(4)));
        const float *S = (which == 0 ? sp.Wf32 : sp.Vf32) + (int64_t)p * K * m2 + q;
#pragma unroll 8
        for (int c = 0; c < K; ++c) {
            const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(S + (int64_t)c * m2));
            const double t = tip[c];
            acc[0] = fma((double)v.x, t, acc[0]);
            acc[1] = fma((double)v.y, t, acc[1]);
            acc[2] = fma((double)v.z, t, acc[2]);
            acc[3] = fma((double)v.w, t, acc[3]);
        }
    }
    if (TWISTED) {   // one window per chain: nothing to race with
        const int64_t row = cd.vec0 + (int64_t)cd.vdir * r;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nr) x[row + (int64_t)cd.vdir * e] -= acc[e];
        return;
    }
    const int64_t row = cd.row0 + (which == 0 ? r : cd.nrows - m + r);
    // when 2m > nrows the two windows overlap: the two contributions to a row must not race
    if (2 * m > cd.nrows) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nr) atomicAdd(x + row + e, -acc[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nr) x[row + e] -= acc[e];
    }
}

hipError_t launch_spike_correct(int K, int m, const ChainDesc *chains, int nchains, const double *Wf, const double *Vf,
                                const double *xb, const double *xt, double *x, hipStream_t st, int mode, bool twisted, int m1,
                                const float *Wf32, const float *Vf32, int nthreads)
{
    if (nchains <= 0 || K <= 0 || m <= 0) return hipSuccess;
    if (m1 <= 0 || m1 > m || Wf32 == nullptr) m1 = m;
    SpikeStore sp{Wf, Vf, Wf32, Vf32, m, m1};
    // workgroup size: 2 rows (fp64 part) / 4 rows (fp32 part) per lane, so nt lanes cover 2 nt / 4 nt rows.  With few chains
    // (strong scaling: 182 chains, windows of 1216 rows) 256-lane workgroups give 3 unequal workgroups per chain -- 546 on 256
    // CUs; 128 lanes give 5 (option correct_threads)
    const int nt = (nthreads == 64 || nthreads == 128 || nthreads == 256) ? nthreads : 128;
    const int nb64 = (m1 + 2 * nt - 1) / (2 * nt), nb32 = (m - m1 + 4 * nt - 1) / (4 * nt);
    if (twisted) {
        const dim3 grid = mode == 2 ? dim3(nb64 + nb32, 2, 1) : dim3(nb64 + nb32, nchains, 1);
        hipLaunchKernelGGL(k_spike_correct<true>, grid, dim3(nt), (size_t)K * sizeof(double), st, K, sp, chains, nchains, xb, xt, x, mode, nb64);
        return hipGetLastError();
    }
    const dim3 grid = mode == 2 ? dim3(nb64 + nb32, 2, 1) : dim3(nb64 + nb32, nchains, 2);
    hipLaunchKernelGGL(k_spike_correct<false>, grid, dim3(nt), (size_t)K * sizeof(double), st, K, sp, chains, nchains, xb, xt, x, mode, nb64);
    return hipGetLastError();
}

// full fp64 window (column-major K x m per chain) -> [fp64 K x m1 | fp32 K x (m - m1)]; near = 0: the interface is at
// stored row 0 (W windows), near = 1: at stored row m - 1 (V windows).  max32[0] = max |entry| that went to fp32.
__global__ __launch_bounds__(256) void k_spike_split(int K, int m, int m1, int near_end, const double *full, double *p64, float *p32,
                                                     double *max32)
{
    const int64_t col = blockIdx.y;   // chain * K + column
    const double *src = full + col * m;
    const int m2 = m - m1;
    double mx = 0.0;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < m; r += gridDim.x * blockDim.x) {
        const double v = src[r];
        const int dist = near_end ? m - 1 - r : r;          // rows from the interface
        if (dist < m1) p64[col * m1 + (near_end ? r - m2 : r)] = v;
        else { p32[col * m2 + (near_end ? r : r - m1)] = (float)v; mx = fmax(mx, fabs(v)); }
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o));
    if ((threadIdx.x & 63) == 0 && mx > __hip_atomic_load(max32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_max_pos(max32, mx);
}

hipError_t launch_spike_split(int K, int m, int m1, int nchains, int near_end, const double *full, double *p64, float *p32,
                              double *max32, hipStream_t st)
{
    if (nchains <= 0 || K <= 0 || m <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_spike_split, dim3((m + 255) / 256 < 4 ? (m + 255) / 256 : 4, nchains * K), dim3(256), 0, st, K, m, m1, near_end,
                       full, p64, p32, max32);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Twisted factorisation, setup pieces (spike_internal.h: ChainDesc).
// ------------------------------------------------------------------------------------------
// the band in factor space: chain-local row r of a vdir = -1 chain is vector-space row vec0 - r, and its diagonal d (column
// offset d - K) is the original's diagonal 2K - d (column offset K - d = -(d - K))
__global__ __launch_bounds__(256) void k_band_flip(const double *src, int64_t lds, int K, const ChainDesc *chains, double *dst,
                                                   int64_t ldd)
{
    const ChainDesc cd = chains[blockIdx.y];
    const int d = blockIdx.z;
    const int sd = cd.vdir > 0 ? d : 2 * K - d;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < cd.nrows; r += gridDim.x * blockDim.x)
        dst[(int64_t)d * ldd + cd.row0 + r] = src[(int64_t)sd * lds + cd.vec0 + (int64_t)cd.vdir * r];
}

hipError_t launch_band_flip(const double *src, int64_t lds, int K, const ChainDesc *chains, int nchains, int max_rows, double *dst,
                            int64_t ldd, hipStream_t st)
{
    if (nchains <= 0) return hipSuccess;
    int gx = (max_rows + 255) / 256;
    if (gx > 16) gx = 16;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_band_flip, dim3(gx, nchains, 2 * K + 1), dim3(256), 0, st, src, lds, K, chains, dst, ldd);
    return hipGetLastError();
}

// K <= 32 (diagonal-major LU scratch): one workgroup per chain, thread j = column j.  With i = np - K + a the chain-local row:
//   L_bb[a][b] = lu[(K + b - a) ld + row0 + i]  (b < a),   U_bb[a][b] = lu[(K + b - a) ld + row0 + i]  (b >= a)
//   B[a][b]    = band[(2K + b - a) ldb + row0 + i]  (b <= a)
// Tb[:, j] = D^-1 L_bb^-1 B[:, j]  (forward substitution, then 1/diag);  Gb[:, j] = column j of the inverse of D^-1 U_bb
__global__ __launch_bounds__(64) void k_seam_small(const double *lu, int64_t ld, int K, const double *band, int64_t ldb,
                                                   const ChainDesc *chains, double *Tb, double *Gb)
{
    __shared__ double F[32][33];
    const int p = blockIdx.x, j = threadIdx.x;
    const ChainDesc cd = chains[p];
    if (!has_bot(cd)) return;   // (cleared by the host)
    const int64_t r0 = cd.row0 + cd.nrows - K;
    for (int t = threadIdx.x; t < K * K; t += blockDim.x) {
        const int a = t / K, b = t % K;
        F[a][b] = lu[(int64_t)(K + b - a) * ld + r0 + a];
    }
    __syncthreads();
    if (j >= K) return;
    double z[32];
#pragma unroll
    for (int a = 0; a < 32; ++a) {
        if (a < K) {
            double v = (j <= a) ? band_at(cd, band, ldb, K, cd.nrows - K + a, 2 * K + j - a) : 0.0;
#pragma unroll
            for (int b = 0; b < 32; ++b)
                if (b < a) v = fma(-F[a][b], z[b], v);
            z[a] = v;
        }
    }
#pragma unroll
    for (int a = 0; a < 32; ++a)
        if (a < K) Tb[((int64_t)p * K + a) * K + j] = z[a] / F[a][a];
#pragma unroll
    for (int a = 31; a >= 0; --a) {
        if (a < K) {
            double v = (a == j) ? 1.0 : 0.0;
#pragma unroll
            for (int b = 0; b < 32; ++b)
                if (b > a && b < K) v = fma(-(F[a][b] / F[a][a]), z[b], v);
            z[a] = v;
        }
    }
#pragma unroll
    for (int a = 0; a < 32; ++a)
        if (a < K) Gb[((int64_t)p * K + a) * K + j] = z[a];
}

hipError_t launch_seam_small(const double *lu, int64_t ld, int K, const double *band, int64_t ldb, const ChainDesc *chains,
                             int nchains, double *Tb, double *Gb, hipStream_t st)
{
    if (nchains <= 0 || K < 1 || K > 32) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_seam_small, dim3(nchains), dim3(64), 0, st, lu, ld, K, band, ldb, chains, Tb, Gb);
    return hipGetLastError();
}

// W[t] = Tb[2t] J Gb[2t+1], V[t] = Tb[2t+1] J Gb[2t]: (T J G)[i][j] = sum_k T[i][K-1-k] G[k][j].  One workgroup per
// (pair, which, 16-row strip); a setup-time product of K^3 multiply-adds per matrix (0.5 GFLOP at the headline size).
__global__ __launch_bounds__(256) void k_seam_products(int K, const double *Tb, const double *Gb, double *W, double *V)
{
    extern __shared__ double Ts[];   // 16 rows of T, reversed columns: Ts[i][k] = T[i0 + i][K-1-k]
    const int t = blockIdx.x, which = blockIdx.y, i0 = blockIdx.z * 16;
    const int64_t kk = (int64_t)K * K;
    const double *T = Tb + (int64_t)(2 * t + which) * kk, *G = Gb + (int64_t)(2 * t + 1 - which) * kk;
    double *out = (which == 0 ? W : V) + (int64_t)t * kk;
    for (int q = threadIdx.x; q < 16 * K; q += blockDim.x) {
        const int i = q / K, k = q % K;
        Ts[q] = (i0 + i < K) ? T[(int64_t)(i0 + i) * K + (K - 1 - k)] : 0.0;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < K; j += blockDim.x) {
        double acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0;
        for (int k = 0; k < K; ++k) {
            const double g = G[(int64_t)k * K + j];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = fma(Ts[i * K + k], g, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i0 + i < K) out[(int64_t)(i0 + i) * K + j] = acc[i];
    }
}

hipError_t launch_seam_products(int K, int npairs, const double *Tb, const double *Gb, double *W, double *V, hipStream_t st)
{
    if (npairs <= 0 || K <= 0) return hipSuccess;
    const int nt = K >= 256 ? 256 : (K >= 128 ? 128 : 64);
    hipLaunchKernelGGL(k_seam_products, dim3(npairs, 2, (K + 15) / 16), dim3(nt), (size_t)16 * K * sizeof(double), st, K, Tb, Gb, W, V);
    return hipGetLastError();
}

__global__ void k_flip_kk(int K, const double *in, int first, int stride, double *out)
{
    const int i = blockIdx.x;
    const double *a = in + (int64_t)(first + i * stride) * K * K;
    double *o = out + (int64_t)i * K * K;
    for (int t = threadIdx.x; t < K * K; t += blockDim.x) {
        const int r = t / K, c = t % K;
        o[t] = a[(int64_t)(K - 1 - r) * K + (K - 1 - c)];
    }
}

hipError_t launch_flip_kk(int K, int count, const double *in, int first, int stride, double *out, hipStream_t st)
{
    if (count <= 0 || K <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_flip_kk, dim3(count), dim3(256), 0, st, K, in, first, stride, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Narrow bands (K <= 8), one rank, coupled, stored spikes: the whole coupling step in two small launches.  The general path
// (k_iface_apply: one workgroup per interface, three barrier-separated K x K mat-vecs; then k_spike_correct: a 256-thread
// workgroup per chain end for its ~64-200 rows) costs 30-60 us for a few MB at 8192-16384 chains -- a fifth of a K = 1
// apply.  k_tips_small saves the K values at both ends of every chain of the swept vector (the corrections below overwrite
// them), k_couple_small has every chain solve its two interface systems itself and correct its first and last m rows.
// ------------------------------------------------------------------------------------------
__global__ void k_tips_small(int nchains, int K, const ChainDesc *chains, const double *y, double *tipT, double *tipB)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchains * K) return;
    const int p = t / K, a = t % K;
    const ChainDesc cd = chains[p];
    tipT[t] = y[cd.row0 + a];
    tipB[t] = y[cd.row0 + cd.nrows - K + a];
}

__global__ __launch_bounds__(64) void k_couple_small(int nchains, int K, int m, const ChainDesc *chains, const double *tipT,
                                                     const double *tipB, const double *WT, const double *ST, const double *VT,
                                                     const double *Wf, const double *Vf, double *y)
{
#pragma clang fp contract(off)
    __shared__ double gb[8], gt[8], v1[8], xt[8], xbp[8], xtn[8];
    const int p = blockIdx.x, a = threadIdx.x;
    const ChainDesc cd = chains[p];
    const int kk = K * K;
    // interface i (between chains i and i+1): x_t = S^-1 (g_t - W g_b), x_b = g_b - V x_t; matrices stored transposed
    for (int half = 0; half < 2; ++half) {
        const int i = half == 0 ? p - 1 : p;
        if (i < 0 || i + 1 >= nchains) { if (a < K) (half == 0 ? xbp : xtn)[a] = 0.0; __syncthreads(); continue; }
        if (a < K) { gb[a] = tipB[(int64_t)i * K + a]; gt[a] = tipT[(int64_t)(i + 1) * K + a]; }
        __syncthreads();
        if (a < K) {
            double s = 0.0;
            for (int c = 0; c < K; ++c) s += WT[(int64_t)i * kk + c * K + a] * gb[c];
            v1[a] = gt[a] - s;
        }
        __syncthreads();
        if (a < K) {
            double s = 0.0;
            for (int c = 0; c < K; ++c) s += ST[(int64_t)i * kk + c * K + a] * v1[c];
            xt[a] = s;
            if (half == 1) xtn[a] = s;           // the bottom of chain p takes x_t of the interface below it
        }
        __syncthreads();
        if (half == 0 && a < K) {                // the top of chain p takes x_b of the interface above it
            double s = 0.0;
            for (int c = 0; c < K; ++c) s += VT[(int64_t)i * kk + c * K + a] * xt[c];
            xbp[a] = gb[a] - s;
        }
        __syncthreads();
    }
    for (int r = a; r < m; r += 64) {
        if (p > 0) {
            double s = 0.0;
            for (int c = 0; c < K; ++c) s += Wf[((int64_t)p * K + c) * m + r] * xbp[c];
            y[cd.row0 + r] -= s;
        }
        if (p + 1 < nchains) {
            double s = 0.0;
            for (int c = 0; c < K; ++c) s += Vf[((int64_t)p * K + c) * m + r] * xtn[c];
            y[cd.row0 + cd.nrows - m + r] -= s;
        }
    }
}

// K = 1 (round 3): the same step with everything a lane needs requested up front.  k_couple_small walks through four
// dependent memory round trips per chain (tips -> 1 x 1 matrices -> spike entries -> y) behind eight workgroup barriers;
// for a tridiagonal system every operand's address is known at launch, so one lane = one (chain end, window row) loads its
// two tips, the three scalars of its interface, its spike entry and its y entry at once, solves the 1 x 1 interface system
// redundantly and corrects its row: one round trip.  One wave per (chain, end), four of them per workgroup.
__global__ __launch_bounds__(256) void k_couple_k1(int nchains, int m, const ChainDesc *chains, const double *tipT, const double *tipB,
                                                   const double *WT, const double *ST, const double *VT, const double *Wf,
                                                   const double *Vf, double *y)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // one wave per (chain, end)
    const int p = q >> 1, end = q & 1;                   // end 0: top window (interface p-1 | p), 1: bottom window (p | p+1)
    if (p >= nchains) return;
    const int i = end == 0 ? p - 1 : p;                  // interface between chains i and i+1
    if (i < 0 || i + 1 >= nchains) return;
    const ChainDesc cd = chains[p];
    const double gb = tipB[i], gt = tipT[i + 1], w = WT[i], sinv = ST[i], v = VT[i];
    const double xt = sinv * (gt - w * gb);              // x_t = S^-1 (g_t - W g_b)
    const double tip = end == 0 ? gb - v * xt : xt;      // the top window takes x_b of the interface above, the bottom one x_t below
    const double *S = (end == 0 ? Wf : Vf) + (int64_t)p * m;
    const int64_t row0 = end == 0 ? cd.row0 : cd.row0 + cd.nrows - m;
    for (int r = lane; r < m; r += 64) y[row0 + r] -= S[r] * tip;
}

// K = 2, 3 behind the fused scan (which stores the chain-end values): the same one-round-trip step.  One wave per (chain,
// end); every lane loads the 2 K tips and the three K x K matrices of its interface (same addresses in all lanes: one
// request each), solves the interface system redundantly (K = 2: 12 multiply-adds) and corrects its window rows.  Replaces
// k_iface_apply + k_spike_correct (two launches, 13-20 us at 2048-4096 chains) on one rank.
template <int KK>
__global__ __launch_bounds__(256) void k_couple_kn(int nchains, int m, const ChainDesc *chains, const double *tipT, const double *tipB,
                                                   const double *WT, const double *ST, const double *VT, const double *Wf,
                                                   const double *Vf, double *y)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // one wave per (chain, end)
    const int p = q >> 1, end = q & 1;
    if (p >= nchains) return;
    const int i = end == 0 ? p - 1 : p;                  // interface between chains i and i+1
    if (i < 0 || i + 1 >= nchains) return;
    const ChainDesc cd = chains[p];
    double gb[KK], gt[KK], v1[KK], xt[KK], tip[KK], W[KK * KK], Si[KK * KK], V[KK * KK];
#pragma unroll
    for (int a = 0; a < KK; ++a) { gb[a] = tipB[(int64_t)i * KK + a]; gt[a] = tipT[(int64_t)(i + 1) * KK + a]; }
#pragma unroll
    for (int t = 0; t < KK * KK; ++t) { W[t] = WT[(int64_t)i * KK * KK + t]; Si[t] = ST[(int64_t)i * KK * KK + t]; V[t] = VT[(int64_t)i * KK * KK + t]; }
    // the window rows of this lane (the first 128 rows of the window: all of it in the usual case) are requested together
    // with the interface operands -- one memory round trip for the whole step; addresses clamped, loads unconditional
    const double *S = (end == 0 ? Wf : Vf) + (int64_t)p * KK * m;
    const int64_t row0 = end == 0 ? cd.row0 : cd.row0 + cd.nrows - m;
    constexpr int PRE = 2;
    double sv[PRE][KK], yv[PRE];
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int r = lane + 64 * u, rc = r < m ? r : 0;
#pragma unroll
        for (int c = 0; c < KK; ++c) sv[u][c] = S[(int64_t)c * m + rc];
        yv[u] = y[row0 + rc];
    }
    // x_t = S^-1 (g_t - W g_b), x_b = g_b - V x_t; matrices stored transposed (XT[c*K + a] = X[a][c])
#pragma unroll
    for (int a = 0; a < KK; ++a) {
        double s = gt[a];
#pragma unroll
        for (int c = 0; c < KK; ++c) s -= W[c * KK + a] * gb[c];
        v1[a] = s;
    }
#pragma unroll
    for (int a = 0; a < KK; ++a) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < KK; ++c) s += Si[c * KK + a] * v1[c];
        xt[a] = s;
    }
#pragma unroll
    for (int a = 0; a < KK; ++a) {
        double s = gb[a];
#pragma unroll
        for (int c = 0; c < KK; ++c) s -= V[c * KK + a] * xt[c];
        tip[a] = end == 0 ? s : xt[a];                   // the top window takes x_b of the interface above, the bottom one x_t below
    }
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int r = lane + 64 * u;
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < KK; ++c) s += sv[u][c] * tip[c];
        if (r < m) y[row0 + r] = yv[u] - s;
    }
    for (int r = lane + 64 * PRE; r < m; r += 64) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < KK; ++c) s += S[(int64_t)c * m + r] * tip[c];
        y[row0 + r] -= s;
    }
}

hipError_t launch_couple_small(int nchains, int K, int m, const ChainDesc *chains, double *tips, const double *WT, const double *ST,
                               const double *VT, const double *Wf, const double *Vf, double *y, hipStream_t st, bool tips_ready)
{
    if (nchains <= 0 || K < 1 || K > 8) return hipErrorInvalidValue;
    double *tipT = tips, *tipB = tips + (size_t)nchains * K;
    if (!tips_ready) hipLaunchKernelGGL(k_tips_small, dim3((nchains * K + 255) / 256), dim3(256), 0, st, nchains, K, chains, y, tipT, tipB);
    const dim3 gw((2 * nchains + 3) / 4);
    if (K == 1) hipLaunchKernelGGL(k_couple_k1, gw, dim3(256), 0, st, nchains, m, chains, tipT, tipB, WT, ST, VT, Wf, Vf, y);
    else if (K == 2) hipLaunchKernelGGL((k_couple_kn<2>), gw, dim3(256), 0, st, nchains, m, chains, tipT, tipB, WT, ST, VT, Wf, Vf, y);
    else if (K == 3) hipLaunchKernelGGL((k_couple_kn<3>), gw, dim3(256), 0, st, nchains, m, chains, tipT, tipB, WT, ST, VT, Wf, Vf, y);
    else hipLaunchKernelGGL(k_couple_small, dim3(nchains), dim3(64), 0, st, nchains, K, m, chains, tipT, tipB, WT, ST, VT, Wf, Vf, y);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// read-bandwidth ceiling: the same access shape as a sweep's tile stream (16 B per lane, 1 KiB per wave
// instruction, non-temporal, every byte read once) with nothing else in the way.  bench.py reports it beside
// the spec peak so that roofline.frac can be read against what this chip delivers for a pure read stream.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_read_bw(const d2 *src, int64_t n2, double *sink)
{
    // every block streams one contiguous region front to back (as a chain does), 8 x 16-byte loads in flight per lane
    const int64_t per = (n2 / gridDim.x) & ~(int64_t)2047;
    const d2 *p = src + blockIdx.x * per + threadIdx.x;
    d2 a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = d2{0, 0};
    for (int64_t i = 0; i + 2048 <= per; i += 2048) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += __builtin_nontemporal_load(p + i + k * 256);
    }
    double t = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += a[k].x + a[k].y;
    if (t == 1.2345e300) sink[0] = t;  // keeps the loads alive, never true for finite data
}

hipError_t launch_read_bw(const double *src, int64_t ndoubles, double *sink, hipStream_t st)
{
    hipLaunchKernelGGL(k_read_bw, dim3(256 * 8), dim3(256), 0, st, reinterpret_cast<const d2 *>(src), ndoubles / 2, sink);
    return hipGetLastError();
}

}  // namespace spike
