// conv_x3.hip - the implicit-GEMM convolutions of conv.hip on the f16 matrix pipe with
// 3-term split products (see common.hip.h, "x3" mode).  Same tiling idea (positions of the
// (t, f') plane flattened with one virtual zero column per row, halo-shared f taps, two
// time planes, normalise-on-load, InstanceNorm partials in the epilogue), re-balanced for a
// pipe that is ~5x faster:
//   * tile = 256 positions x all output channels; 8 waves x 32 positions (64-channel kernels) or 4 waves x 32
//     (128-channel sub-pixel kernel), each wave all output channels
//   * a stage = one time plane of one 32-channel chunk: activations are normalised, split
//     into fp16 hi/lo and written to LDS ONCE, then read by 3 taps x COUT/16 x 3 products
//   * stage s+1's global loads are issued into registers before stage s's MFMAs (T14-style
//     issue-early / write-late), so HBM/L2 latency hides under the matrix work.
#include "kernels.h"

namespace X3_NS {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef CX_EARLY
#define CX_EARLY 0            // 1 = weave the next stage's activation loads into the staging phase (CX_WRITE_PREFETCH):
                              // measured 3 % SLOWER on the dense conv (6.05 vs 5.87 ms, same-session A/B), kept as a probe
#endif
#define CX_STRIDE 48          // halfs per LDS activation row (32 used): 96 B = 6 x 16 B slots, the row pitch at which
                              // the B-operand ds_read_b128 (16 consecutive rows x 4 lane groups) is bank-conflict free
                              // for every base row (80 B was 2-way; brute-forced over the b128 lane-group map)

__device__ __forceinline__ f32x4 norm_prelu4x(f32x4 v, f32x4 sc, f32x4 sh, f32x4 al) {
    f32x4 y = v * sc + sh;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = y[e] >= 0.f ? y[e] : al[e] * y[e];
    return y;
}

// kernarg pointer tables are indexed with compile-time constants only (a runtime index would
// make the compiler spill the whole ConvArgs struct to scratch)
__device__ __forceinline__ const float* sel4(const float* const (&p)[4], int i) {
    return i == 0 ? p[0] : (i == 1 ? p[1] : (i == 2 ? p[2] : p[3]));
}

// ---------------------------------------------------------------------------------
// The kernel body is written as macros over a fixed set of local names (setup / prefetch /
// write / MFMA / epilogue) so that alternative schedules can reuse it.  Measured alternatives
// (DESIGN.md section 7): an 8-wave "ping-pong" block whose two 4-wave groups are held in
// anti-phase by the barriers (one group's MFMAs over the other's staging VALU) was 9% SLOWER than
// two independent 4-wave blocks per CU: on this machine a wave's VALU instructions take issue
// slots from the MFMA stream of the wave it shares a SIMD with, so MFMA time and VALU time add
// whichever way they are interleaved, and the forced rendezvous only adds barrier idle time.
// What pays is removing instructions: see CX_WRITE.
// ---------------------------------------------------------------------------------
#define CX_DECLS                                                                                         \
    constexpr int NTHR = 64 * NWV;                            /* threads per block */                      \
    constexpr int RSTEP = NTHR / 8;                           /* staged rows per load step (8 threads per row) */ \
    constexpr int CX_ROWS = 16 * NPB * NWV;                   /* staged rows = MFMA positions (NWV waves x NPB x 16) */ \
    constexpr int CX_TILE = CX_ROWS - 2;                      /* outputs per tile: the last two MFMA positions lack \
                                                                 their right halo and are discarded, which makes  \
                                                                 the staged tile exactly NACT rows per thread */   \
    constexpr int CB = COUT / 16;                                                                        \
    constexpr int TAPS = NT * 3;                                                                         \
    constexpr int NACT = CX_ROWS * 8 / NTHR;                 /* float4 loads per thread per stage */     \
    constexpr int W16 = 3 * CB * 2 * 64;                      /* 16-byte units of weights per stage */    \
    constexpr int NW = W16 / NTHR;                                                                       \
    constexpr int ACT = CX_ROWS * CX_STRIDE;                  /* halfs per activation plane (hi or lo) */ \
    constexpr int SMEM = 2 * ACT + W16 * 8;                   /* [act hi | act lo | weights]: one base register, constant offsets */

// per-thread staging map: row p = (tid >> 3) + RSTEP e, channel quad qd = tid & 7 of the 32-chunk.
// off1[e] = byte offset inside the clip of this thread's 16 B of the t plane (first row when the
// row is zero padding); inv1 / inv0 = per-lane padding bits of the t / t - dil plane.
#define CX_SETUP(VALID)                                                                                  \
    const int Fp = a.F + 1;                                                                              \
    const int q0 = tile * CX_TILE;                                                                       \
    const int qd = tid & 7;                                                                              \
    const int lds_col = 8 * (qd & 3) + 4 * (qd >> 2);       /* chain slot order: [4g..4g+3 | 16+4g..] */ \
    _Float16* const wrow = sm + (tid >> 3) * CX_STRIDE + lds_col;           /* staging writes */           \
    const _Float16* const brow = sm + (16 * NPB * wv + c) * CX_STRIDE + 8 * g;   /* B-operand reads */    \
    const _Float16* const alane = sm + 2 * ACT + lane * 8;                       /* A-operand reads */    \
    unsigned off1[NACT];                                                                                 \
    unsigned inv1 = 0, inv0 = 0;                                                                         \
    const bool revt_ = TRAINV && a.revt != 0;              /* training dgrad: the plane is walked backwards in time */ \
    const unsigned dFb = (unsigned)(a.dil * a.F) * 256u;                                                 \
    const unsigned dFs = revt_ ? dFb : 0u - dFb;          /* byte step from frame t to the logical frame t - dil */ \
    const float osc_ = (TRAINV && a.oscale != nullptr) ? *a.oscale : 1.0f;                               \
    /* (t, f) of row e by one division and 32-row steps (every F + 1 here is > 32) */                    \
    const int qfirst_ = q0 - 1 + (tid >> 3);                                                             \
    int tt_ = (qfirst_ < 0 ? 0 : qfirst_) / Fp, ff_ = qfirst_ - tt_ * Fp;    /* q = -1 -> (0, -1): padding */ \
    _Pragma("unroll") for (int e = 0; e < NACT; ++e) {                                                   \
        bool ok1 = false, ok0 = false;                                                                   \
        off1[e] = qd * 16;                                                                               \
        if ((VALID) && ff_ >= 0 && ff_ < a.F && tt_ < a.T) {                                             \
            ok1 = true;                                                                                  \
            off1[e] = (unsigned)((revt_ ? a.T - 1 - tt_ : tt_) * a.F + ff_) * 256u + qd * 16;            \
            if (NT == 2 && tt_ >= a.dil) ok0 = true;                                                     \
        }                                                                                                \
        if (!ok1) inv1 |= 1u << e;                                                                       \
        if (!ok0) inv0 |= 1u << e;                                                                       \
        ff_ += RSTEP;                                                                                    \
        if (ff_ >= Fp) { ff_ -= Fp; ++tt_; }                                                             \
    }                                                                                                    \
    const size_t clip_bytes = (size_t)a.T * a.F * 256;                                                   \
    f32x4 acc[CB][NPB];               /* start from the bias: its load latency hides in the prologue */ \
    _Pragma("unroll") for (int cb = 0; cb < CB; ++cb) {                                                  \
        const f32x4 bias_ = ldg4(a.bias + 16 * cb + 4 * g);                                              \
        _Pragma("unroll") for (int tb = 0; tb < NPB; ++tb) acc[cb][tb] = bias_;                          \
    }                                                                                                    \
    const int nst = a.nslots * 2 * NT;                                                                   \
    f32x4 pre[NACT];                                                                                     \
    f32x4 sc = splat4(1.f), sh = splat4(0.f), al = splat4(1.f);

// issue stage S's activation loads into registers (consumed by CX_WRITE(S) one phase later):
// uniform 64-bit base (slot, clip, channel half) + per-lane 32-bit offset, no 64-bit VALU math
#define CX_PREFETCH(S)                                                                                   \
    do {                                                                                                 \
        const int chunk_ = (S) / NT, kt_ = (S) - chunk_ * NT;                                            \
        const int slot_ = chunk_ >> 1, half_ = chunk_ & 1;                                               \
        const char* src_ = reinterpret_cast<const char*>(sel4(a.in, slot_)) + b * clip_bytes + half_ * 128; \
        _Pragma("unroll") for (int e = 0; e < NACT; ++e) {                                               \
            /* t - dil plane: same row dil*F positions earlier where it exists, else any in-bounds row */ \
            const unsigned o_ = (NT == 2 && kt_ == 0) ? off1[e] + ((inv0 >> e) & 1u ? 0u : dFs) : off1[e]; \
            pre[e] = *reinterpret_cast<const f32x4*>(src_ + o_);     /* unconditional; masked at write */ \
        }                                                                                                \
        const float* nsc_ = ((a.img_mask >> slot_) & 1u) ? nullptr : sel4(a.nscale, slot_);              \
        if (nsc_ != nullptr) {                                                                           \
            sc = ldg4(nsc_ + b * 64 + half_ * 32 + qd * 4);                                              \
            sh = ldg4(sel4(a.nshift, slot_) + b * 64 + half_ * 32 + qd * 4);                             \
            al = ldg4(sel4(a.nalpha, slot_) + half_ * 32 + qd * 4);   /* no arithmetic on loaded data here: \
                                                                           it would wait for every load above */ \
        } else {                                          /* un-normalised slot: identity (exact) */      \
            sc = splat4(1.f); sh = splat4(0.f); al = splat4(1.f);                                        \
        }                                                                                                \
    } while (0)

// the stage's 24 KB weight image (L2-resident) goes global -> registers -> LDS.  CX_WEARLY = 1: the fetch is issued
// BEFORE the barrier that ends the previous stage's MFMA phase (the MFMA operand registers are dead by then), so its
// latency overlaps the barrier wait and the staging VALU instead of being waited for at the end of the staging phase.
#ifndef CX_WEARLY
#define CX_WEARLY 1
#endif
#ifndef CX_NOWB
#define CX_NOWB 0            // 1 = TIMING-ONLY ablation: the slot images are not written (later layers read garbage)
#endif
#define CX_WFETCH(S)                                                                                     \
    do {                                                                                                 \
        const int chunkf_ = (S) / NT, ktf_ = (S) - chunkf_ * NT;                                         \
        const u32x4* wsrc_ = reinterpret_cast<const u32x4*>(w16) + ((long)chunkf_ * TAPS + ktf_ * 3) * (CB * 128); \
        _Pragma("unroll") for (int i = 0; i < NW; ++i) wpre[i] = wsrc_[tid + NTHR * i];                   \
    } while (0)
#if CX_WEARLY
#define CX_WFETCH_IN_WRITE(S)
#else
#define CX_WFETCH_IN_WRITE(S) CX_WFETCH(S);
#endif
// normalise + PReLU + fp16 hi/lo split of the prefetched stage S, written to this tile's LDS buffers.
// PReLU(y) = y + (alpha - 1) min(y, 0): one v_min + half a packed FMA per value.
// (the stage's 24 KB weight image is L2-resident: its loads are issued at the top and land while
// the activation VALU work runs, so they never occupy registers during the MFMA phase)
#define CX_WRITE(S)                                                                                      \
    do {                                                                                                 \
        const int chunkw_ = (S) / NT, ktw_ = (S) - chunkw_ * NT;                                         \
        CX_WFETCH_IN_WRITE(S)                                                                            \
        const unsigned inv_ = (NT == 2 && ktw_ == 0) ? inv0 : inv1;                                      \
        const int slotw_ = chunkw_ >> 1;                                                                 \
        if ((a.img_mask >> slotw_) & 1u) {              /* pre-split image slot (block-uniform): two LDS stores */ \
            _Pragma("unroll") for (int e = 0; e < NACT; ++e) {                                           \
                u32x4 v = __builtin_bit_cast(u32x4, pre[e]);                                             \
                if (inv_ & (1u << e)) v = u32x4{0u, 0u, 0u, 0u};                                         \
                *reinterpret_cast<unsigned long long*>(wrow + RSTEP * e * CX_STRIDE) =                   \
                    (unsigned long long)v[0] | ((unsigned long long)v[1] << 32);                         \
                *reinterpret_cast<unsigned long long*>(wrow + RSTEP * e * CX_STRIDE + ACT) =             \
                    (unsigned long long)v[2] | ((unsigned long long)v[3] << 32);                         \
            }                                                                                            \
        } else {                                                                                         \
            const f32x4 am1 = al - splat4(1.f);                                                          \
            /* the newest slot's t-plane stage also stores the image for the block's later layers */     \
            char* const wb_ = (!CX_NOWB && a.img_out != nullptr && slotw_ == a.nslots - 1 && ktw_ == NT - 1) \
                                  ? reinterpret_cast<char*>(a.img_out) + b * clip_bytes + (chunkw_ & 1) * 128 : nullptr; \
            _Pragma("unroll") for (int e = 0; e < NACT; ++e) {                                           \
                f32x4 v = pre[e];                                                                        \
                v = v * sc + sh;                          /* branch-free: identity slots carry (1, 0, 0) */ \
                f32x4 mn;                                                                                \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) mn[r] = fminf(v[r], 0.f);                  \
                v = mn * am1 + v;                                                                        \
                if (inv_ & (1u << e)) v = splat4(0.f);   /* zero padding (select, no branch) */          \
                f16x4 hi, lo;                                                                            \
                split4(v, hi, lo);                                                                       \
                *reinterpret_cast<f16x4*>(wrow + RSTEP * e * CX_STRIDE) = hi;                            \
                *reinterpret_cast<f16x4*>(wrow + RSTEP * e * CX_STRIDE + ACT) = lo;                      \
                if (wb_ != nullptr) {                     /* rows 1 .. CX_TILE are this tile's own positions */ \
                    const int p_ = (tid >> 3) + RSTEP * e;                                               \
                    if (p_ >= 1 && p_ <= CX_TILE && !(inv_ & (1u << e))) {                               \
                        f16x8 hl = __builtin_shufflevector(hi, lo, 0, 1, 2, 3, 4, 5, 6, 7);              \
                        *reinterpret_cast<f16x8*>(wb_ + off1[e]) = hl;                                   \
                    }                                                                                    \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < NW; ++i)                                                   \
            *reinterpret_cast<u32x4*>(sm + 2 * ACT + (tid + NTHR * i) * 8) = wpre[i];                     \
    } while (0)

// PROBE (CX_EARLY=1, off by default): CX_WRITE(S) with stage S+1's prefetch woven in: as soon as row group e of
// stage S has been normalised, split and stored, its registers take stage S+1's load, so the next stage's HBM
// requests are in flight during the rest of the staging VALU work, the barrier AND the MFMAs, instead of only during
// the MFMAs.  Motivation: the single-product probe showed that dropping 2/3 of the MFMAs only
// buys 30 % on this kernel and that the remainder equals its HBM time at 5 TB/s.  Result: 3 % slower - the loads
// issued mid-staging make every later `pre[e]` consumer wait behind younger requests (in-order vmcnt) and delay the
// weight-image wait at the end of the staging phase; the one-stage-ahead form stays.
#define CX_WRITE_PREFETCH(S, HAS_NEXT)                                                                   \
    do {                                                                                                 \
        const int chunkw_ = (S) / NT, ktw_ = (S) - chunkw_ * NT;                                         \
        const u32x4* wsrc_ = reinterpret_cast<const u32x4*>(w16) + ((long)chunkw_ * TAPS + ktw_ * 3) * (CB * 128); \
        u32x4 wpre[NW];                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NW; ++i) wpre[i] = wsrc_[tid + NTHR * i];                   \
        const unsigned inv_ = (NT == 2 && ktw_ == 0) ? inv0 : inv1;                                      \
        const f32x4 am1 = al - splat4(1.f);                                                              \
        const int chunkn_ = ((S) + 1) / NT, ktn_ = ((S) + 1) - chunkn_ * NT;                             \
        const int slotn_ = chunkn_ >> 1, halfn_ = chunkn_ & 1;                                           \
        const char* srcn_ = reinterpret_cast<const char*>(sel4(a.in, (HAS_NEXT) ? slotn_ : 0)) + b * clip_bytes + halfn_ * 128; \
        _Pragma("unroll") for (int e = 0; e < NACT; ++e) {                                               \
            f32x4 v = pre[e];                                                                            \
            if (HAS_NEXT) {                                                                              \
                const unsigned o_ = (NT == 2 && ktn_ == 0) ? off1[e] + ((inv0 >> e) & 1u ? 0u : dFs) : off1[e]; \
                pre[e] = *reinterpret_cast<const f32x4*>(srcn_ + o_);                                    \
            }                                                                                            \
            v = v * sc + sh;                                                                             \
            f32x4 mn;                                                                                    \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) mn[r] = fminf(v[r], 0.f);                      \
            v = mn * am1 + v;                                                                            \
            if (inv_ & (1u << e)) v = splat4(0.f);                                                       \
            f16x4 hi, lo;                                                                                \
            split4(v, hi, lo);                                                                           \
            *reinterpret_cast<f16x4*>(wrow + RSTEP * e * CX_STRIDE) = hi;                                \
            *reinterpret_cast<f16x4*>(wrow + RSTEP * e * CX_STRIDE + ACT) = lo;                          \
        }                                                                                                \
        if (HAS_NEXT) {                                                                                  \
            const float* nsc_ = sel4(a.nscale, slotn_);                                                  \
            if (nsc_ != nullptr) {                                                                       \
                sc = ldg4(nsc_ + b * 64 + halfn_ * 32 + qd * 4);                                         \
                sh = ldg4(sel4(a.nshift, slotn_) + b * 64 + halfn_ * 32 + qd * 4);                       \
                al = ldg4(sel4(a.nalpha, slotn_) + halfn_ * 32 + qd * 4);                                \
            } else {                                                                                     \
                sc = splat4(1.f); sh = splat4(0.f); al = splat4(1.f);                                    \
            }                                                                                            \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < NW; ++i)                                                   \
            *reinterpret_cast<u32x4*>(sm + 2 * ACT + (tid + NTHR * i) * 8) = wpre[i];                     \
    } while (0)

// 3 taps x CB output blocks x NPB position blocks x 3 split products from this tile's LDS buffers, software-
// pipelined by hand: the A fragments of group j+1 (and, at the end of a tap, the B fragments of the next tap)
// are requested BEFORE group j's 3 * NPB MFMAs are issued, and sched_barriers keep the compiler from sinking
// the reads back to their first use (left alone it emits read -> s_waitcnt lgkmcnt(0) -> 4..8 MFMAs, exposing
// the LDS latency 24 times per stage).  Worth 4 % at two blocks per CU, 5 % at one.
#define CX_MFMA()                                                                                        \
    do {                                                                                                 \
        f16x8 bh[NPB], bl[NPB], ah, alo;                                                                 \
        _Pragma("unroll") for (int tb = 0; tb < NPB; ++tb) {                                             \
            bh[tb] = *reinterpret_cast<const f16x8*>(brow + (16 * tb) * CX_STRIDE);                      \
            bl[tb] = *reinterpret_cast<const f16x8*>(brow + (16 * tb) * CX_STRIDE + ACT);                \
        }                                                                                                \
        ah = *reinterpret_cast<const f16x8*>(alane);                                                     \
        alo = *reinterpret_cast<const f16x8*>(alane + 512);                                              \
        _Pragma("unroll") for (int j = 0; j < 3 * CB; ++j) {                                             \
            const int kf = j / CB, cb = j - kf * CB;                                                     \
            const bool last_cb = cb == CB - 1 && kf < 2;                                                 \
            f16x8 ahn = ah, aln = alo, bln[NPB];                                                         \
            if (j + 1 < 3 * CB) {                                                                        \
                ahn = *reinterpret_cast<const f16x8*>(alane + (j + 1) * 1024);                           \
                aln = *reinterpret_cast<const f16x8*>(alane + (j + 1) * 1024 + 512);                     \
            }                                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            _Pragma("unroll") for (int tb = 0; tb < NPB; ++tb) acc[cb][tb] = mfma32l(ah, bl[tb], acc[cb][tb]); \
            if (last_cb) {                                                                               \
                __builtin_amdgcn_sched_barrier(0);                                                       \
                _Pragma("unroll") for (int tb = 0; tb < NPB; ++tb)                                       \
                    bln[tb] = *reinterpret_cast<const f16x8*>(brow + (16 * tb + kf + 1) * CX_STRIDE + ACT); \
                __builtin_amdgcn_sched_barrier(0);                                                       \
            }                                                                                            \
            _Pragma("unroll") for (int tb = 0; tb < NPB; ++tb) acc[cb][tb] = mfma32h(ah, bh[tb], acc[cb][tb]);  \
            _Pragma("unroll") for (int tb = 0; tb < NPB; ++tb) acc[cb][tb] = mfma32l(alo, bh[tb], acc[cb][tb]); \
            if (last_cb) {                                                                               \
                __builtin_amdgcn_sched_barrier(0);                                                       \
                _Pragma("unroll") for (int tb = 0; tb < NPB; ++tb) {                                     \
                    bh[tb] = *reinterpret_cast<const f16x8*>(brow + (16 * tb + kf + 1) * CX_STRIDE);     \
                    bl[tb] = bln[tb];                                                                    \
                }                                                                                        \
            }                                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            ah = ahn;                                                                                    \
            alo = aln;                                                                                   \
        }                                                                                                \
    } while (0)

// bias, store, per-wave InstanceNorm partial sums into red[wv] (same contract as conv3_kernel)
#define CX_EPILOGUE(VALID)                                                                               \
    do {                                                                                                 \
        bool ok[NPB];                                                                                    \
        long obase[NPB];                                                                                 \
        const int qe_ = q0 + 16 * NPB * wv + c;                                                          \
        int t = qe_ / Fp, f = qe_ - t * Fp;                                                              \
        _Pragma("unroll") for (int tb = 0; tb < NPB; ++tb) {                                             \
            if (tb > 0) { f += 16; if (f >= Fp) { f -= Fp; ++t; } }                                      \
            ok[tb] = (VALID) && (t < a.T) && (f < a.F) && (16 * NPB * wv + 16 * tb + c < CX_TILE);       \
            if (a.mode == 1) {                                                                           \
                ok[tb] = ok[tb] && ((f & 1) == 0);                                                       \
                const int F2 = (a.F + 1) >> 1;                                                           \
                obase[tb] = ((long)(b * a.T + t) * F2 + (f >> 1)) * 64;                                  \
            } else if (a.mode == 2) {                                                                    \
                obase[tb] = ((long)(b * a.T + t) * (2 * a.F) + 2 * f) * 64;                              \
            } else {                                                                                     \
                obase[tb] = ((long)(b * a.T + (revt_ ? a.T - 1 - t : t)) * a.F + f) * 64;                \
            }                                                                                            \
        }                                                                                                \
        _Pragma("unroll") for (int cb = 0; cb < CB; ++cb) {                                              \
            f32x4 s1 = splat4(0.f), s2 = splat4(0.f);                                                    \
            _Pragma("unroll") for (int tb = 0; tb < NPB; ++tb) {                                         \
                const f32x4 v = acc[cb][tb];                                                             \
                if (ok[tb]) {                                                                            \
                    long off = obase[tb] + 16 * (cb & 3) + 4 * g;                                        \
                    if (a.mode == 2) off += (cb >> 2) * 64;                                              \
                    if (TRAINV && a.accum) stg4(a.out + off, ldg4(a.out + off) + v * splat4(osc_));      \
                    else stg4(a.out + off, v);                                                           \
                    s1 += v;                                                                             \
                    s2 += v * v;                                                                         \
                }                                                                                        \
            }                                                                                            \
            if (a.partials) {                                                                            \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                          \
                    const float t1 = red_c_sum(s1[r]), t2 = red_c_sum(s2[r]);                            \
                    if (c == 0) {                                                                        \
                        red[wv][16 * cb + 4 * g + r][0] = t1;                                            \
                        red[wv][16 * cb + 4 * g + r][1] = t2;                                            \
                    }                                                                                    \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
    } while (0)

// XCD-aware block order: the dispatcher deals consecutive workgroups round-robin to the 8
// XCDs (private L2 each).  Re-map so XCD x walks a CONTIGUOUS range of (clip, tile): the
// rows a tile reads as its t-dil plane were read moments earlier as the t plane of a
// neighbouring tile on the SAME XCD -> second read is an L2 hit, not a second HBM fetch.
__device__ __forceinline__ int xcd_contiguous_block() {
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qn = nwg >> 3, rn = nwg & 7;
    return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
}

// TRAINV = true: the training-step instantiation that honours ConvArgs::revt / accum / oscale (dense-block data
// gradient, train.hip); the inference instantiations compile those paths out
template <int NT, int COUT, int NPB, int NWV = 4, bool TRAINV = false>
__global__ __launch_bounds__(64 * NWV) void conv3x_kernel(ConvArgs a, const _Float16* __restrict__ w16) {
    CX_DECLS
    __shared__ __attribute__((aligned(16))) _Float16 sm[SMEM];
    __shared__ float red[NWV][COUT][2];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4, wv = tid >> 6;
    const int logical = xcd_contiguous_block();
    const int b = logical / a.ntiles, tile = logical - b * a.ntiles;
    CX_SETUP(true)

    CX_PREFETCH(0);
    u32x4 wpre[NW];
#if CX_WEARLY == 2
    CX_WFETCH(0);
#endif
#pragma unroll 1
    for (int s = 0; s < nst; ++s) {
#if CX_WEARLY == 1
        CX_WFETCH(s);                                     // in flight across the barrier and the staging VALU
#endif
        __syncthreads();                                  // stage s-1 fully consumed
#if CX_EARLY
        if (s + 1 < nst) CX_WRITE_PREFETCH(s, true);      // stage s+1's loads issued row group by row group
        else CX_WRITE_PREFETCH(s, false);
        __syncthreads();
#else
        CX_WRITE(s);
        __syncthreads();
        if (s + 1 < nst) CX_PREFETCH(s + 1);              // in flight during the MFMAs below
#if CX_WEARLY == 2
        // the NEXT stage's weight image too: a whole MFMA phase ahead of its LDS store (NW more registers live across
        // the MFMAs: needs the 3-waves-per-SIMD register budget of the 128-row tile, -DCX_NPB64=2 -DCX_NWV64=4)
        if (s + 1 < nst) CX_WFETCH(s + 1);
#endif
#endif
        CX_MFMA();
    }
    CX_EPILOGUE(true);
    if (a.partials) {
        __syncthreads();
        for (int i = tid; i < COUT * 2; i += NTHR) {
            const int co = i >> 1, wh = i & 1;
            float t = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < NWV; ++w8) t += red[w8][co][wh];
            a.partials[(((long)b * a.ntiles + tile) * COUT + co) * 2 + wh] = t;
        }
    }
}

}  // namespace X3_NS
using namespace X3_NS;

// the dense / 1x3 convs stage 256-row tiles (254 outputs); the 128-channel sub-pixel conv 128-row tiles (126 outputs)
// 64-channel kernels: 8 waves x 2 position blocks = the same 256-row tile and the same 76 KB of LDS as 4 waves x 4,
// but four waves per SIMD instead of two (120 VGPRs): 3.5 % faster although every A fragment is now read by
// twice as many waves.  (4 x 4: -DCX_NPB64=4 -DCX_NWV64=4.)
#ifndef CX_NPB64
#define CX_NPB64 2            // position blocks per wave
#endif
#ifndef CX_NWV64
#define CX_NWV64 8            // waves per block (tile = 16 * NPB * NWV rows)
#endif
#ifndef CX_NPB128
#define CX_NPB128 2           // 128-channel sub-pixel conv: 4 waves x 2 position blocks = 128-row tile (8 x 1 needs
                              // 82 KB of LDS with its 8 KB of partial sums -> one block per CU, 8 % slower)
#endif
#ifndef CX_NWV128
#define CX_NWV128 4
#endif
#ifndef X3_SINGLE
int conv3x_ntiles(int T, int F, int cout) {
    const int tile = (cout == 128 ? 16 * CX_NPB128 * CX_NWV128 : 16 * CX_NPB64 * CX_NWV64) - 2;
    return (T * (F + 1) + tile - 1) / tile;
}

void launch_conv3_x3_dgrad(LaunchCtx ctx, const ConvArgs& a, const void* w16, int B) {
    dim3 grid(a.ntiles * B);
    LAUNCH(ctx, "dense_train_bwd", (conv3x_kernel<2, 64, CX_NPB64, CX_NWV64, true><<<grid, 64 * CX_NWV64, 0, ctx.stream>>>(
                                       a, reinterpret_cast<const _Float16*>(w16))));
}
#endif

void launch_conv3_x3(LaunchCtx ctx, const ConvArgs& a, const void* w16, int B, int time_taps, int cout) {
    dim3 grid(a.ntiles * B);                              // 1-D: see the XCD re-map in the kernel
    const _Float16* w = reinterpret_cast<const _Float16*>(w16);
    if (time_taps == 2 && cout == 64)
        LAUNCH(ctx, "conv_dense", (conv3x_kernel<2, 64, CX_NPB64, CX_NWV64><<<grid, 64 * CX_NWV64, 0, ctx.stream>>>(a, w)));
    else if (time_taps == 1 && cout == 64)
        LAUNCH(ctx, "conv_1x3", (conv3x_kernel<1, 64, CX_NPB64, CX_NWV64><<<grid, 64 * CX_NWV64, 0, ctx.stream>>>(a, w)));
    else
        LAUNCH(ctx, "conv_subpixel", (conv3x_kernel<1, 128, CX_NPB128, CX_NWV128><<<grid, 64 * CX_NWV128, 0, ctx.stream>>>(a, w)));
}
