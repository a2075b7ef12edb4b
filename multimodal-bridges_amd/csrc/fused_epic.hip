// K-E fused: the whole MultiModalEPiC forward -- and the whole generation loop -- for one jet
// inside one workgroup, activations never leaving the register file.
// Reference: encoders/multimodal_epic.py:158-194, encoders/epic.py:71-107,145-159,187-208,
// encoders/embedder.py:63-83, model/solvers.py:21-25,59-81, model/multimodal_bridge_matching.py:227-259.
//
// Layout (gfx950, wave64).  One 256-thread workgroup = one jet of N <= 128 particles; wave w owns
// particles [32w, 32w+32).  Activations are kept TRANSPOSED (features x particles) in the C/D
// layout of the 32x32 MFMAs: a 32-feature tile is one f32x16 per lane, lane&31 = particle,
// register r / lane half h = feature (r&3) + 8(r>>2) + 4h.  A layer Y^T = W X^T then needs NO data
// movement between layers: registers 8j..8j+7 of an input tile ARE the B operand of one 16-deep k-block
// of v_mfma_f32_32x32x16_bf16 (after the exact three-plane bf16 split of fused_common.h: mma_x, which
// forms every fp32 product from six bf16 partial products), and the A operand is the weight fragment,
// pre-split and pre-packed on the host so that three dwordx4 per lane feed the block and a wave
// reads 3 KiB contiguous.  Per-jet inputs of a layer (time embedding x mask, global vector, bias)
// are NOT run through the MFMA: they are a per-jet matvec c,d done once per layer by the 256
// threads and enter as the accumulator's initial value  m[p]*c[o] + d[o]  (25 % fewer MFMA flops
// than the concatenated formulation, same result up to fp32 summation order).
// Masked mean/sum pooling is a DPP reduction over the 32 particle lanes + an LDS combine of the 4
// waves.  The per-jet MLPs (M = 1 row) are VALU matvecs over transposed weights.
//
// Occupancy.  The per-jet chain pool -> fc_glob1 -> fc_glob2 -> d-vector is serial and latency
// bound (L2 weight fetches + barriers), and nothing of the SAME jet can overlap it.  So the kernel
// is built to run TWO workgroups (two jets) per CU: the register budget is held to 256 per wave
// (unified VGPR+AGPR file, 512 per SIMD lane) by parking the fused encoder's skip/input tiles
// (cat[h1,h2], read once per layer and by the heads) in LDS -- 64 KiB per jet, each lane reading back
// only what it wrote (no barrier, conflict-free b128) -- and by folding the residual into the
// accumulator's initial value.  One jet's matvec latency then hides under the other jet's MFMAs.

#ifndef MMB_FE_WAVES
#define MMB_FE_WAVES 4
#endif
#if MMB_FE_WAVES == 5     // second compilation of this file (fused_epic_w5.hip): five waves = up to 160 particles per jet,
                          // forward and generation entry points only, under names of their own
#define mmb_fused_epic_forward mmb_fused_epic_forward_w5
#define mmb_fused_epic_generate mmb_fused_epic_generate_w5
#define mmb_fused_uni_forward mmb_fused_uni_forward_w5
#define mmb_fused_uni_generate mmb_fused_uni_generate_w5
#define mmb_fused_epic_train_forward mmb_fused_epic_train_forward_w5
#define mmb_fused_uni_train_forward mmb_fused_uni_train_forward_w5
#endif
#include "fused_common.h"
#include <stdlib.h>

namespace mmb {

// Developer-only phase timing (build with MMB_EXTRA_FLAGS=-DMMB_FE_PROF, benchmarks/prof_fused.py):
// thread 0 of block 0 accumulates core-clock cycles per phase category.
#ifdef MMB_FE_PROF
__device__ unsigned long long fe_prof[16];
#define FE_PROF(L, cat)                                                        \
  do {                                                                         \
    if (threadIdx.x == 0 && blockIdx.x == 0) {                                 \
      const unsigned long long now_ = clock64();                               \
      atomicAdd(&fe_prof[cat], now_ - (L).tlast);                              \
      (L).tlast = now_;                                                        \
    }                                                                          \
  } while (0)
#else
#define FE_PROF(L, cat) do { } while (0)
#endif
enum { PF_EMBED = 0, PF_POOL, PF_GLOB1, PF_GLOB2, PF_DVEC, PF_LOC1, PF_LOC2, PF_EPI, PF_OUT, PF_PROJ_MMA, PF_HEAD, PF_STEP, PF_X1, PF_X2, PF_X3, PF_N };

// ------------------------------------------------------------------ one EPiC encoder ----
// What every stage of a forward needs besides its tiles
// ACT: this wave's tile takes part in the per-particle work (TilePlan).  A compile-time property of the context: an idle
// tile runs its own, small instantiation of the forward (same barriers, its share of the per-jet chain, no GEMMs), so
// that the working tiles' instruction stream is the one it was before tiles could be idle.
template <bool ACT>
struct FCtxT {
  static constexpr bool active = ACT;
  const float* P;          // packed parameters
  WRsrc R;                 // ... as a buffer resource
  Lds& L;
  float m;                 // this lane's particle mask
  int tid, lane, wave, half, lane16;
  const float* ctab;       // (n_c_layers,128) time vectors W_t temb of the CURRENT grid point (CT kernels)
};

// An idle tile contributes exact zeros to the pooled sums
__device__ __forceinline__ void zero_partials(float* partial, int wave, int lane) {
  partial[wave * 128 + lane] = 0.0f;
  partial[wave * 128 + 64 + lane] = 0.0f;
}

// Issue priority of this wave against the other jet's wave on the same SIMD (arbitration is by priority, then age).
// The per-jet chain is a string of short dependent steps between barriers: when the partner sits in a GEMM -- MFMAs
// back to back with vector work in every gap -- an equal-priority chain wave only gets the leftover issue slots and
// the jet's three other waves wait at the next barrier for it.  Chain: high priority; GEMM stream: low.
#ifndef MMB_PRIO_CHAIN
#define MMB_PRIO_CHAIN 2
#endif
__device__ __forceinline__ void prio_chain() { __builtin_amdgcn_s_setprio(MMB_PRIO_CHAIN); }
__device__ __forceinline__ void prio_gemm() { __builtin_amdgcn_s_setprio(0); }

// __syncthreads() that the machine scheduler does not move loads across either: the weight prefetches of the chain are
// buffer loads the scheduler considers free to hoist over s_barrier -- into the pooling butterfly, where their
// registers do not exist yet (they went to scratch as they arrived: load, vmcnt(0), scratch store).
__device__ __forceinline__ void sync_pin() {
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
}

// One float4 per lane of wave `w` moves a per-layer vector (time vector from the table, biases) from global memory into
// LDS: requested at the stage's entry, written in the interval before the barrier that publishes it, so that the
// accumulator initialisation reads LDS instead of waiting on L2 in front of a GEMM (round 2: flat loads + vmcnt(0)).
struct Stage {
  float4 v;
};
// No branch and no select on the loaded value: lanes past the vector re-read its last float4 (a select against zero made
// the compiler wait for the load -- vmcnt(0), i.e. for every weight prefetch issued just before it -- at once).
__device__ __forceinline__ void stage_issue(Stage& s, const float* src, int n, int lane) {
  const int o = 4 * lane < n ? 4 * lane : n - 4;
  s.v = *reinterpret_cast<const float4*>(src + o);
}
// up to three vectors, one per wave (pointer chosen by wave index, ONE load instruction: two conditional loads into the
// same registers made the second one wait for the first -- vmcnt(0) again)
__device__ __forceinline__ void stage_issue3(Stage& s, const float* a, const float* b, const float* c3, int n, int wave,
                                             int lane) {
  const float* src = wave == 0 ? a : (wave == 1 ? b : c3);
  stage_issue(s, src, n, lane);
}
__device__ __forceinline__ void stage_commit(const Stage& s, float* dst, int n, int lane) {
  if (4 * lane < n) *reinterpret_cast<float4*>(dst + 4 * lane) = s.v;
}

// offsets of the LDS copy of the descriptors: wave-uniform, used as scalar offsets of buffer loads
__device__ __forceinline__ int desc_wx(const Lds& L, int i) { return __builtin_amdgcn_readfirstlane(L.desc[i].wx); }
// one field of descriptor i, fetched where it is used (four whole descriptors per layer held in SGPRs were spilling
// into VGPR lanes, and offsets that lost their SGPR turned buffer loads into waterfall loops)
enum { F_WX = 0, F_WT = 1, F_WG = 2, F_B = 3 };
template <int F>
__device__ __forceinline__ int dfield(const Lds& L, int i) {
  return __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(L.desc)[4 * i + F]);
}
// N consecutive descriptors in ONE LDS round trip (N ds_read_b128 issued together), kept in VGPRs; fld<F>(d[i]) moves a
// field to an SGPR where it is used.  (One ds_read_b32 + wait per field was nine dependent LDS latencies per layer.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int N>
struct DescV {
  i32x4 d[N];
};
template <int N>
__device__ __forceinline__ DescV<N> load_descs(const Lds& L, int i) {
  DescV<N> r;
  ffor<N>([&](auto nc) { constexpr int n = decltype(nc)::v; r.d[n] = reinterpret_cast<const i32x4*>(L.desc)[i + n]; });
  return r;
}
template <int F>
__device__ __forceinline__ int fld(const i32x4& d) { return __builtin_amdgcn_readfirstlane(d[F]); }

// x: HT tiles (in/out), skip: HT tiles.  Global vector in L.gcur.  `li` walks the descriptors (LDS copy).
// RD: depth of the weight-fragment ring (fused_common.h: mma_r).  CT: the time vectors come from the per-grid-point
// table (generation); otherwise they are matvecs on the time embedding (single forward, training).
template <int HT, int ET, int G, int RD, bool CT>
struct Epic {
  static constexpr int H = 32 * HT;
  // The ring is carried through the per-jet chain only where the chain leaves the registers for it (fc_glob1 of a
  // 128-wide encoder holds 144 weights per lane): otherwise the layer primes it after fc_glob1.
  static constexpr bool CARRY = HT <= 2;
  typedef MV<2 * H + G, H> MVA;      // fc_glob1
  typedef MV<H, G> MVB;              // fc_glob2
  typedef MV<G, H> MVC;              // global-vector block of fc_loc1 (+ its bias)
  typedef MV<ET, H> MVT;             // time block of a 32*HT-wide layer

  template <class CMV, class TR>
  static __device__ __forceinline__ void save_pooled(TR& tr, Lds& L, int tid) {
    if constexpr (TR::on) {
      float* dst = tr.row(2 * H);
      if (dst != nullptr)
        for (int k = tid; k < 2 * H; k += FE_THREADS) dst[k] = L.pooled[CMV::vidx(k)];
    }
  }

  // EPiCLayer (epic.py:187-208) + the encoder-level skip add (:89-90).  skipf(IF<t>) yields skip tile t.
  // next_wx: first fragment of the GEMM that follows this layer (< 0: that consumer primes the ring itself).
  // Training slots: pooled, gh, g_new, g_out, sign bits of lh's pre-activation, lh, sign bits of y1 (the
  // leaky-relu input behind l_out), l_out.
  // Five intervals separated by barriers; every global load of the chain is requested in the first one:
  //   I0 pool partials | I1 pooled vector | I2 fc_glob1 (+ time vector, staged vectors) | I3 fc_glob2 | I4 d-vector, g
  template <typename SkipFn, class TR, class CX>
  static __device__ __forceinline__ void layer(f32x16 (&x)[HT], SkipFn&& skipf, bool use_skip, CX& c, int& li, int& ci,
                                               int next_wx, WRing<RD>& ring, TR& tr) {
    Lds& L = c.L;
    const int tid = c.tid, lane = c.lane, wave = c.wave, half = c.half;
    prio_chain();
    const DescV<4> dv = load_descs<4>(L, li);     // fc_glob1, fc_glob2, fc_loc1, fc_loc2
    li += 4;
    FE_PROF(L, PF_EPI);
#ifndef MMB_WHATIF_NOCHAIN        // developer timing build (wrong results on purpose): what the forward costs without the per-jet chain
    typename MVA::Pre wa;
    typename MVT::Pre wt;
    Stage st;
    if constexpr (CARRY) MVA::prefetch(wa, c.R, fld<F_WT>(dv.d[0]), fld<F_B>(dv.d[0]), tid);      // a narrow encoder has the registers now
    {
      const float* b2 = c.P + fld<F_B>(dv.d[3]);
      stage_issue3(st, CT ? c.ctab + (int64_t)ci * 128 : b2, b2, b2, H, wave, lane);
    }
    if constexpr (!CT) MVT::template prefetch<false>(wt, c.R, fld<F_WT>(dv.d[2]), 0, tid);
    ++ci;
    FE_PROF(L, PF_X1);
    if constexpr (CX::active) pool_partials<HT>(x, c.m, L.partial, wave, lane);
    else zero_partials(L.partial, wave, lane);
    FE_PROF(L, PF_X2);
    typename MVB::Pre wb;
    typename MVC::Pre wc;
    if constexpr (CARRY) {
      MVB::prefetch(wb, c.R, fld<F_WT>(dv.d[1]), fld<F_B>(dv.d[1]), tid);
      MVC::prefetch(wc, c.R, fld<F_WG>(dv.d[2]), fld<F_B>(dv.d[2]), tid);
    }
    sync_pin();                                                             // 1: wave partials
    // a wide encoder only now: requested next to the pooling butterfly, the 64 registers of the first batch went to
    // scratch as they arrived (load, vmcnt(0), scratch store: worse than no prefetch)
    if constexpr (!CARRY) MVA::prefetch(wa, c.R, fld<F_WT>(dv.d[0]), fld<F_B>(dv.d[0]), tid);
    FE_PROF(L, PF_POOL);
    pooled_combine<H, MVA>(L.pooled, L.partial, L.cnt, L.gcur, G, tid);
    sync_pin();                                                             // 2: pooled vector
    FE_PROF(L, PF_X3);
    save_pooled<MVA>(tr, L, tid);
    MVA::template run<MMB_ACT_LRELU>(wa, c.R, fld<F_WT>(dv.d[0]), L.pooled, L.gh, nullptr, tid);
    if constexpr (CT) {
      if (wave == 0) stage_commit(st, L.cvec, H, lane);
    } else {
      MVT::template run<MMB_ACT_NONE>(wt, c.R, fld<F_WT>(dv.d[2]), L.temb, L.cvec, nullptr, tid);
    }
    if (wave == 1) stage_commit(st, L.bvec, H, lane);
    if constexpr (!CARRY) {                // fc_glob1's 144 weights per lane are gone: now the rest fits
      MVB::prefetch(wb, c.R, fld<F_WT>(dv.d[1]), fld<F_B>(dv.d[1]), tid);
      MVC::prefetch(wc, c.R, fld<F_WG>(dv.d[2]), fld<F_B>(dv.d[2]), tid);
      ring_prime(ring, c.R, fld<F_WX>(dv.d[2]), c.lane16);
    }
    sync_pin();                                                             // 3: gh, time vector, second bias
    if constexpr (TR::on) tr.vec(L.gh, H, tid);
    FE_PROF(L, PF_GLOB1);
    MVB::template run<MMB_ACT_NONE>(wb, c.R, fld<F_WT>(dv.d[1]), L.gh, L.gnew, L.gcur, tid);   // g' = g + fc_glob2(.)
    sync_pin();                                                             // 4: g'
    if constexpr (TR::on) tr.vec(L.gnew, G, tid);
    FE_PROF(L, PF_GLOB2);
    MVC::template run<MMB_ACT_NONE>(wc, c.R, fld<F_WG>(dv.d[2]), L.gnew, L.dvec, nullptr, tid);
    if constexpr (TR::on) {
      const bool dr = tr.drop_p > 0.0f;
      if (dr) ++tr.off;
      const SaveSlot so = tr.next();
      if (tid < G) {
        float v = lrelu_f(L.gnew[tid]);
        if (dr) {
          const uint64_t idx = (uint64_t)tr.b * G + tid;
          uint32_t r[4];
          philox4(tr.seed, tr.off, idx >> 2, 0xD0D0u, r);
          v = tr.drop(v, r[idx & 3]);
        }
        v += use_skip ? L.gskip[tid] : 0.0f;
        L.gcur[tid] = v;
        const int64_t i = (int64_t)tr.b * so.ld + tid;
        if (i < so.n) so.p[i] = v;
      }
    } else {
      if (tid < G) L.gcur[tid] = lrelu_f(L.gnew[tid]) + (use_skip ? L.gskip[tid] : 0.0f);
    }
    sync_pin();                                                             // 5: d-vector, g
#else
    ++ci;
    if constexpr (!CARRY) ring_prime(ring, c.R, fld<F_WX>(dv.d[2]), c.lane16);
#endif
    FE_PROF(L, PF_DVEC);
    prio_gemm();
    if constexpr (CX::active) {
    f32x16 lh[HT];
    acc_init<HT>(lh, c.m, L.cvec, L.dvec, half);
    mma_r<HT, HT, RD, true>(lh, [&](auto fc) -> const f32x16& { return x[decltype(fc)::v]; }, c.R, fld<F_WX>(dv.d[2]), fld<F_WX>(dv.d[3]), c.lane16, ring);
    if constexpr (TR::on) tr.bits(sign_bits<HT>(lh), tid);
    act_tiles<HT, MMB_ACT_LRELU>(lh);
    if constexpr (TR::on) tr.template tiles<HT>(lh, half);
    FE_PROF(L, PF_LOC1);
    // x <- (x + b2) + W2 lh : the residual rides in the accumulator, x's registers ARE the accumulator
    ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v;
      ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
        const float4 bv = *reinterpret_cast<const float4*>(L.bvec + 32 * t + 8 * q + 4 * half);
        x[t][4 * q + 0] += bv.x; x[t][4 * q + 1] += bv.y; x[t][4 * q + 2] += bv.z; x[t][4 * q + 3] += bv.w; }); });
    mma_r<HT, HT, RD, CARRY>(x, [&](auto fc) -> const f32x16& { return lh[decltype(fc)::v]; }, c.R, fld<F_WX>(dv.d[3]), next_wx, c.lane16, ring);
    FE_PROF(L, PF_LOC2);
    bool dr = false;
    if constexpr (TR::on) {
      dr = tr.drop_p > 0.0f;
      if (dr) ++tr.off;
    }
    uint64_t keep = 0;
    if constexpr (TR::on) {
      tr.bits(sign_bits<HT>(x), tid);
      if (dr) keep = dropout_keep_bits<HT>(tr.drop_p, tr.seed, tr.off, (uint64_t)tr.b * tr.N + tr.p, half);
    }
    ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v;
      ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; x[t][r] = lrelu_f(x[t][r]); });
      if constexpr (TR::on) {
        if (dr) {
          const uint32_t kw = (uint32_t)(keep >> (32 * (t >> 1)));
          ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v;
            x[t][r] = ((kw >> (16 * (t & 1) + r)) & 1u) ? x[t][r] / (1.0f - tr.drop_p) : 0.0f; });
        }
      }
      if (use_skip) {
        const f32x16 sk = skipf(IF<t>{});
        ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; x[t][r] += sk[r]; });
      } });
    if constexpr (TR::on) tr.template tiles<HT>(x, half);
    } else {
      if constexpr (TR::on) { tr.si += 4; if (tr.drop_p > 0.0f) ++tr.off; }       // the slots and the Philox call an idle tile skips
    }
    FE_PROF(L, PF_EPI);
  }

  // output layer: wn Linear on cat[t, x, g] (epic.py:93-96); OTO output tiles, OW = true output width
  // Training slot: out (only when OW is a whole number of tiles).
  // PFN: the GEMM after this one takes its first fragments from this one's tail (false: it primes the ring itself)
  template <int OTO, int OW, bool PFN, class TR, class CX>
  static __device__ __forceinline__ void output(f32x16 (&out)[OTO], const f32x16 (&x)[HT], CX& c, int& li, int& ci,
                                                int next_wx, WRing<RD>& ring, TR& tr) {
    Lds& L = c.L;
    const int tid = c.tid, lane = c.lane, wave = c.wave, half = c.half;
    typedef MV<G, OW> MVO;               // the packed per-jet blocks have the TRUE output width (8 logits: UniModalEPiC)
    typedef MV<ET, OW> MVTO;
    prio_chain();
    const DescV<1> dv = load_descs<1>(L, li);
    li += 1;
    typename MVO::Pre wo;
    MVO::prefetch(wo, c.R, fld<F_WG>(dv.d[0]), fld<F_B>(dv.d[0]), tid);
    typename MVTO::Pre wt;
    Stage st;
    if constexpr (CT) stage_issue(st, c.ctab + (int64_t)ci * 128, 32 * OTO, lane);
    else MVTO::template prefetch<false>(wt, c.R, fld<F_WT>(dv.d[0]), 0, tid);
    ++ci;
    if constexpr (!CARRY) ring_prime(ring, c.R, fld<F_WX>(dv.d[0]), c.lane16);
    sync_pin();                        // every wave is past the accumulator initialisation of the stage before
    MVO::template run<MMB_ACT_NONE>(wo, c.R, fld<F_WG>(dv.d[0]), L.gcur, L.dvec, nullptr, tid);
    if constexpr (CT) {
      if (wave == 0) stage_commit(st, L.cvec, 32 * OTO, lane);
    } else {
      MVTO::template run<MMB_ACT_NONE>(wt, c.R, fld<F_WT>(dv.d[0]), L.temb, L.cvec, nullptr, tid);
    }
    sync_pin();
    prio_gemm();
    if constexpr (CX::active) {
    acc_init<OTO>(out, c.m, L.cvec, L.dvec, half);
    mma_r<HT, OTO, RD, PFN>(out, [&](auto fc) -> const f32x16& { return x[decltype(fc)::v]; }, c.R, fld<F_WX>(dv.d[0]), next_wx, c.lane16, ring);
    if constexpr (TR::on && OW == 32 * OTO) tr.template tiles<OTO>(out, half);
    } else {
      if constexpr (TR::on && OW == 32 * OTO) tr.si += 1;
    }
    FE_PROF(L, PF_OUT);
  }

  // EPiCProjection (epic.py:145-159) from KIN padded input tiles delivered by xin(IF<t>)
  // Training slots: p0 pre, p0 out, p2 pre, p2 out, pooled, q0 pre, q0 out, q2 pre, q2 out.
  template <int KIN, typename InFn, class TR, class CX>
  static __device__ __forceinline__ void project(f32x16 (&x)[HT], InFn&& xin, CX& c, int& li, int& ci, int next_wx,
                                                 WRing<RD>& ring, TR& tr) {
    Lds& L = c.L;
    const int tid = c.tid, lane = c.lane, wave = c.wave, half = c.half;
    typedef MV<2 * H + ET, H> MVQ0;
    prio_chain();
    const DescV<4> dv = load_descs<4>(L, li);     // mlp_local.0, mlp_local.2, mlp_global.0, mlp_global.2
    li += 4;
    typename MVT::Pre wt;
    Stage st;
    {
      const float* b0 = c.P + fld<F_B>(dv.d[0]);
      stage_issue3(st, CT ? c.ctab + (int64_t)ci * 128 : b0, b0, c.P + fld<F_B>(dv.d[1]), H, wave, lane);
    }
    if constexpr (!CT) MVT::template prefetch<false>(wt, c.R, fld<F_WT>(dv.d[0]), 0, tid);
    ++ci;
    sync_pin();                        // every wave is past the accumulator initialisation of the stage before
    if constexpr (CT) {
      if (wave == 0) stage_commit(st, L.cvec, H, lane);
    } else {
      MVT::template run<MMB_ACT_NONE>(wt, c.R, fld<F_WT>(dv.d[0]), L.temb, L.cvec, nullptr, tid);
    }
    if (wave == 1) stage_commit(st, L.dvec, H, lane);
    if (wave == 2) stage_commit(st, L.bvec, H, lane);
    sync_pin();
    prio_gemm();
    if constexpr (CX::active) {
    f32x16 h[HT];
    acc_init<HT>(h, c.m, L.cvec, L.dvec, half);
    mma_r<KIN, HT, RD, true>(h, xin, c.R, fld<F_WX>(dv.d[0]), fld<F_WX>(dv.d[1]), c.lane16, ring);
    if constexpr (TR::on) tr.template tiles<HT>(h, half);
    act_tiles<HT, MMB_ACT_GELU>(h);
    if constexpr (TR::on) tr.template tiles<HT>(h, half);
    acc_init<HT>(x, 0.0f, nullptr, L.bvec, half);
    mma_r<HT, HT, RD, CARRY>(x, [&](auto fc) -> const f32x16& { return h[decltype(fc)::v]; }, c.R, fld<F_WX>(dv.d[1]), next_wx, c.lane16, ring);
    if constexpr (TR::on) tr.template tiles<HT>(x, half);
    act_tiles<HT, MMB_ACT_GELU>(x);
    if constexpr (TR::on) tr.template tiles<HT>(x, half);
    } else {
      if constexpr (TR::on) tr.si += 4;
    }
    FE_PROF(L, PF_PROJ_MMA);
    prio_chain();
    typename MVQ0::Pre wq0;
    if constexpr (CARRY) MVQ0::prefetch(wq0, c.R, fld<F_WT>(dv.d[2]), fld<F_B>(dv.d[2]), tid);
    if constexpr (CX::active) pool_partials<HT>(x, c.m, L.partial, wave, lane);
    else zero_partials(L.partial, wave, lane);
    typename MV<H, G>::Pre wq2;
    if constexpr (CARRY) MV<H, G>::prefetch(wq2, c.R, fld<F_WT>(dv.d[3]), fld<F_B>(dv.d[3]), tid);
    sync_pin();
    if constexpr (!CARRY) MVQ0::prefetch(wq0, c.R, fld<F_WT>(dv.d[2]), fld<F_B>(dv.d[2]), tid);
    pooled_combine<H, MVQ0>(L.pooled, L.partial, L.cnt, L.temb, ET, tid);          // context = time embedding
    sync_pin();
    save_pooled<MVQ0>(tr, L, tid);
    float* pre_q0 = nullptr;
    if constexpr (TR::on) pre_q0 = tr.row(H);
    MVQ0::template run<MMB_ACT_GELU>(wq0, c.R, fld<F_WT>(dv.d[2]), L.pooled, L.gh, nullptr, tid, pre_q0);
    if constexpr (!CARRY) MV<H, G>::prefetch(wq2, c.R, fld<F_WT>(dv.d[3]), fld<F_B>(dv.d[3]), tid);
    sync_pin();
    float* pre_q2 = nullptr;
    if constexpr (TR::on) { tr.vec(L.gh, H, tid); pre_q2 = tr.row(G); }
    MV<H, G>::template run<MMB_ACT_GELU>(wq2, c.R, fld<F_WT>(dv.d[3]), L.gh, L.gcur, nullptr, tid, pre_q2);
    sync_pin();
    if constexpr (TR::on) tr.vec(L.gcur, G, tid);
    FE_PROF(L, PF_GLOB1);
  }
};

// arguments of the single-launch generation loops
struct GenArgs {
  int steps;
  const float* times;        // (steps) grid points (host linspace, fp32)
  const float* temb_steps;   // (steps, ET) time embedding per grid point (identical for every jet)
  const float* ctab_steps;   // (steps, n_ctab, 128) precomputed W_t temb of every time-consuming layer
  int n_ctab;
  float dt;
  int thermo_kind;
  float thermo_c;            // fp32(-V*gamma)
  const float* delta_n;      // (steps,B,N,V) injected Poisson counts or NULL -> Philox
  uint64_t seed, offset;
  int em;                    // 1: Euler-Maruyama  x += dt v + sigma sqrt(t(1-t)) dW  (solvers.py:27-33, no sqrt(dt))
  float em_sigma;
  const float* delta_w;      // (steps,B,N,DC) injected Wiener increments or NULL -> Philox/Box-Muller
  int stagger;               // core-clock cycles the second workgroup of a CU waits before its first forward (0: none)
  int tiles;                 // bit 0: tiles of padded slots only stay idle (TilePlan); bit 1: tiles steered to SIMDs
  const int32_t* jet_order;  // (n_jets) the jets of this launch, workgroup i starts with jet_order[i]; NULL: 0, 1, 2, ...
  int n_jets;                // entries of jet_order (B: a permutation of all jets; fewer: the listed jets only)
};

// Two jets share a CU and run the SAME program: started together they stay in lock-step -- both in their GEMMs
// (one matrix pipe per SIMD: half speed each), then both in their per-jet chains (pipe idle).  The workgroup that
// finds itself in the second wave slot of its SIMD (HW_ID.wave_id of its first wave) starts `stagger` cycles late, so
// that one jet's chain falls under the other's GEMMs; only workgroups of the first dispatch round wait (later ones
// inherit the offset: they start when their predecessor ends).
__device__ __forceinline__ void stagger_start(int cycles, int tid, int* flag) {
  if (cycles <= 0 || blockIdx.x >= 512) return;
  if (tid == 0) *flag = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xF);      // HW_REG_HW_ID: wave_id[3:0]
  __syncthreads();
  if ((*flag & 1) != 0) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(16);
  }
  __syncthreads();
}


struct FusedArgs {
  const float* ctab;       // generation: (n_c_layers,128) time vectors W_t temb of the CURRENT grid point, or NULL
  const float* P;          // packed parameters
  const LDesc* D;          // layer descriptors
  int nb_branch, nb_fused_exec, use_skip;
  int emb_cw, emb_cb, emb_table;      // offsets into P
  int B, N;
  const float* maskf;      // (B,N)
  const int32_t* counts;   // (B)
  int nb_branch1;          // EPiC layers of the discrete branch (nb_branch: the continuous one); they may differ
  int n_desc;              // descriptors behind D (cached in LDS by every workgroup)
  int tiles;               // training kernels: bit 0 idle tiles of padded slots, bit 1 tile-to-SIMD steering (TilePlan)
  const int32_t* jet_order;  // training kernels: (B) launch order of the jets (mmb_jet_schedule) or NULL
};

// ring depth of a width set: 4 fragments in flight where every GEMM of the model has a multiple of 4 steps (tiles of
// 64+ features), else 2
template <class DM>
constexpr int ring_depth_multi() { return DM::HBT >= 2 ? 4 : 2; }
template <class DM>
constexpr int ring_depth_uni() { return (DM::HT >= 2 && (DM::EDT * DM::HT) % 2 == 0) ? 4 : 2; }

// embedded inputs of one particle as padded 32-feature tiles (embedder.py:74-80 + apply_mask): the continuous
// features (br = 0) or the truncated token embedding (br = 1, quirk Q1); each is built right before the branch that
// consumes it, so that neither occupies 16 registers while the other branch runs
template <class DM, int BR>
__device__ __forceinline__ void embed_tile(f32x16& t, const float (&xp)[4], int kp, float m, const Lds& L, int half) {
  ffor<16>([&](auto rc) {
    constexpr int r = decltype(rc)::v;
    const int f = (r & 3) + 8 * (r >> 2) + 4 * half;
    float v = 0.0f;
    if constexpr (BR == 0) {
      if (f < DM::EC) {
        float s = 0.0f;
#pragma unroll
        for (int dd = 0; dd < DM::DC; ++dd) s += xp[dd] * L.cw[f * 4 + dd];
        v = (s + L.cb[f]) * m;
      }
    } else {
      if (f < DM::ED) v = truncf(L.table[kp * 32 + f] * m);           // quirk Q1
    }
    t[r] = v;
  });
}

// One GELU head (multimodal_epic.py:188-189): Linear(cat[t, f_i, h_i]) -> GELU -> Linear.  in(IF<t>) yields the
// 2*HBT input tiles; `o` receives the output tile; i0: descriptor of the first Linear.  Training slots:
// pre-activation, hidden.
template <class DM, int RD, bool CT, typename InFn, class TR, class CX>
__device__ __forceinline__ void head(f32x16& o, InFn&& in, CX& c, int i0, int cslot,
                                     int next_wx, WRing<RD>& ring, TR& tr) {
  constexpr int HBT = DM::HBT, HFT = DM::HFT;
  Lds& L = c.L;
  const int tid = c.tid, lane = c.lane, wave = c.wave, half = c.half;
  prio_chain();
  const DescV<2> dv = load_descs<2>(L, i0);
  typedef MV<DM::ET, DM::HB> MVT;
  typename MVT::Pre wt;
  Stage st;
  {
    // (the 32 floats of the second bias are followed by more parameters: reading HB of them is harmless)
    const float* b0 = c.P + fld<F_B>(dv.d[0]);
    stage_issue3(st, CT ? c.ctab + (int64_t)cslot * 128 : b0, b0, c.P + fld<F_B>(dv.d[1]), DM::HB, wave, lane);
  }
  if constexpr (!CT) MVT::template prefetch<false>(wt, c.R, fld<F_WT>(dv.d[0]), 0, tid);
  sync_pin();                          // every wave is past the accumulator initialisation of the stage before
  if constexpr (CT) {
    if (wave == 0) stage_commit(st, L.cvec, DM::HB, lane);
  } else {
    MVT::template run<MMB_ACT_NONE>(wt, c.R, fld<F_WT>(dv.d[0]), L.temb, L.cvec, nullptr, tid);
  }
  if (wave == 1) stage_commit(st, L.dvec, DM::HB, lane);
  if (wave == 2) stage_commit(st, L.bvec, 32, lane);
  sync_pin();
  prio_gemm();
  if constexpr (CX::active) {
  f32x16 h[HBT];
  acc_init<HBT>(h, c.m, L.cvec, L.dvec, half);
  mma_r<HFT, HBT, RD, true>(h, in, c.R, fld<F_WX>(dv.d[0]), fld<F_WX>(dv.d[1]), c.lane16, ring);
  if constexpr (TR::on) tr.template tiles<HBT>(h, half);
  act_tiles<HBT, MMB_ACT_GELU>(h);
  if constexpr (TR::on) tr.template tiles<HBT>(h, half);
  f32x16 oo[1];
  acc_init<1>(oo, 0.0f, nullptr, L.bvec, half);
  mma_r<HBT, 1, RD, true>(oo, [&](auto fc) -> const f32x16& { return h[decltype(fc)::v]; }, c.R, fld<F_WX>(dv.d[1]), next_wx, c.lane16, ring);
  o = oo[0];
  } else {
    if constexpr (TR::on) tr.si += 2;
  }
}

// Whole MultiModalEPiC forward for this jet.  which: 0 continuous head, 1 discrete head, 2 both.
// hc: tile whose features 0..DC-1 are the drift; hd: tile whose features 0..V-1 are the logits.
// The ring enters primed with the first fragments of the first GEMM (descriptor 0) and leaves primed the same way:
// the last GEMM of a forward prefetches the first one of the next.
// Training (TR = Train): slots in execution order -- per branch [projection 9, layers 8 each, output 1],
// the concatenated global vector 1, fused encoder [layers 8 each, output 1], heads [pre, hidden] x 2.
template <class DM, bool CT, bool ACT = true, class TR>
__device__ __forceinline__ void forward_jet(const FusedArgs& A, Lds& L, float4* S, const float (&xp)[4], int kp, float m,
                                            int which, f32x16& hc, f32x16& hd, int tid,
                                            WRing<ring_depth_multi<DM>()>& ring, TR& tr) {
  constexpr int HBT = DM::HBT, HFT = DM::HFT, RD = ring_depth_multi<DM>();
  typedef Epic<HBT, DM::ET, DM::GB, RD, CT> EB;
  typedef Epic<HFT, DM::ET, DM::GF, RD, CT> EF;
  const int lane = tid & 63, half = lane >> 5;
  FCtxT<ACT> c{A.P, weight_rsrc(A.P), L, m, tid, lane, tid >> 6, half, lane * 16, A.ctab};
  int li = 0, ci = 0;
  FE_PROF(L, PF_STEP);
  // S (LDS, HFT tiles) = cat[h1, h2]: fused-encoder input, its skip connection AND the heads' input
  // ---- two branch encoders (multimodal_epic.py:173-175)
  ffor<2>([&](auto bc) {
    constexpr int br = decltype(bc)::v;
    f32x16 x[HBT], out[HBT], tin;
    if constexpr (ACT) embed_tile<DM, br>(tin, xp, kp, m, L, half);
    FE_PROF(L, PF_EMBED);
    const int nbb = br == 0 ? A.nb_branch : A.nb_branch1;
    // descriptors: projection 4, layers 4 each, output 1
    const int wx_after_proj = desc_wx(L, nbb > 0 ? li + 4 + 2 : li + 4);
    EB::template project<1>(x, [&](auto) -> const f32x16& { return tin; }, c, li, ci, wx_after_proj, ring, tr);
    if (tid < DM::GB) L.gskip[tid] = L.gcur[tid];        // read in the fourth interval of a layer: barriers in between
    // the skip tiles wait in the upper half of S (free until this branch's own output goes there, or -- branch 0 --
    // until branch 1's does): 16*HBT registers less across the layers
    if constexpr (ACT) ffor<HBT>([&](auto t) { park_tile(S, HBT + decltype(t)::v, x[decltype(t)::v], tid); });
    for (int i = 0; i < nbb; ++i) {
      const int nx = desc_wx(L, i + 1 < nbb ? li + 4 + 2 : li + 4);
      EB::layer(x, [&](auto t) -> f32x16 { return fetch_tile(S, HBT + decltype(t)::v, tid); }, A.use_skip != 0, c, li, ci, nx,
                ring, tr);
    }
    // after this branch's output layer: the other branch's projection, or the fused encoder
    int nx;
    if constexpr (br == 0) nx = desc_wx(L, li + 1);
    else nx = desc_wx(L, A.nb_fused_exec > 0 ? li + 1 + 2 : li + 1);
    EB::template output<HBT, DM::HB, (br == 0 || EF::CARRY)>(out, x, c, li, ci, nx, ring, tr);
    if constexpr (ACT) ffor<HBT>([&](auto t) { park_tile(S, br * HBT + decltype(t)::v, out[decltype(t)::v], tid); });
    if (tid < DM::GB) L.gcat[br * DM::GB + tid] = L.gcur[tid];
  });
  // ---- fused encoder, no projection (K2: nb_fused_exec = constructed - 1).  gcat / gcur / gskip are written and read
  // by wave 0 here (LDS operations of one wave are in order); every other reader sits behind a barrier of the next stage
  if (tid < DM::GF) { L.gcur[tid] = L.gcat[tid]; L.gskip[tid] = L.gcat[tid]; }
  if constexpr (TR::on) tr.vec(L.gcat, DM::GF, tid);
  f32x16 x[HFT], f[HFT];
  if constexpr (ACT) ffor<HFT>([&](auto t) { x[decltype(t)::v] = fetch_tile(S, decltype(t)::v, tid); });
  for (int i = 0; i < A.nb_fused_exec; ++i) {
    const int nx = desc_wx(L, i + 1 < A.nb_fused_exec ? li + 4 + 2 : li + 4);
    EF::layer(x, [&](auto t) -> f32x16 { return fetch_tile(S, decltype(t)::v, tid); }, A.use_skip != 0, c, li, ci, nx, ring, tr);
  }
  // ---- heads (multimodal_epic.py:187-189): cat[t, f_i, h_i] -> GELU -> out.  Descriptors: output 1, c0, c2, d0, d2
  const int ic0 = li + 1, id0 = li + 3;
  EF::template output<HFT, 2 * DM::HB, true>(f, x, c, li, ci, desc_wx(L, which != 1 ? ic0 : id0), ring, tr);
  FE_PROF(L, PF_EPI);
  const int ci_heads = ci;                               // table slots: [.. , continuous head, discrete head]
  if (which != 1)
    head<DM, RD, CT>(hc, [&](auto fc) -> f32x16 {
      constexpr int t = decltype(fc)::v;
      if constexpr (t < HBT) return f[t]; else return fetch_tile(S, t - HBT, tid); }, c, ic0, ci_heads,
      desc_wx(L, which == 2 ? id0 : 0), ring, tr);
  if (which != 0)
    head<DM, RD, CT>(hd, [&](auto fc) -> f32x16 {
      constexpr int t = decltype(fc)::v;
      if constexpr (t < HBT) return f[HBT + t]; else return fetch_tile(S, t, tid); }, c, id0, ci_heads + 1,
      desc_wx(L, 0), ring, tr);
  FE_PROF(L, PF_HEAD);
}

// layer descriptors -> LDS (every stage reads its offsets from there: an LDS broadcast instead of an L2 round trip
// at the head of every layer)
__device__ __forceinline__ void load_desc(const FusedArgs& A, Lds& L, int tid) {
  const int* src = reinterpret_cast<const int*>(A.D);
  int* dst = reinterpret_cast<int*>(L.desc);
  for (int i = tid; i < 4 * A.n_desc; i += FE_THREADS) dst[i] = src[i];
}

template <class DM>
__device__ __forceinline__ void load_constants(const FusedArgs& A, Lds& L, int tid) {
  load_desc(A, L, tid);
  for (int i = tid; i < DM::EC * 4; i += FE_THREADS) {
    const int f = i >> 2, d = i & 3;
    L.cw[i] = d < DM::DC ? A.P[A.emb_cw + f * DM::DC + d] : 0.0f;
  }
  for (int i = tid; i < DM::EC; i += FE_THREADS) L.cb[i] = A.P[A.emb_cb + i];
  for (int i = tid; i < DM::V * 32; i += FE_THREADS) {
    const int v = i >> 5, e = i & 31;
    L.table[i] = e < DM::ED ? A.P[A.emb_table + v * DM::ED + e] : 0.0f;
  }
}

// gather the 8 (V) logits of this lane's particle: features 0..3 sit in lane half 0, 4..7 in half 1
template <class DM>
__device__ __forceinline__ void gather_logits(const f32x16& hd, float (&lg)[DM::V], int lane) {
  static_assert(DM::V == 8, "fused path is instantiated for vocab_size 8");
  const int half = lane >> 5;
  float mine[4] = {hd[0], hd[1], hd[2], hd[3]};
  float other[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) other[i] = __shfl_xor(mine[i], 32, 64);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lg[i] = half == 0 ? mine[i] : other[i];
    lg[4 + i] = half == 0 ? other[i] : mine[i];
  }
}

// ---- single forward: heads to HBM (validation / parity / training-forward of eval mode) ----
template <class DM, int NWT = FE_WAVES>        // (NWT: a name of its own per wave count, the two compilations link together)
__global__ void __launch_bounds__(FE_THREADS, FE_WAVES == 4 ? 2 : 1)
fused_forward_kernel(FusedArgs A, const float* __restrict__ x, const int64_t* __restrict__ k,
                     const float* __restrict__ temb, int64_t temb_stride, int which, float* __restrict__ head_c,
                     float* __restrict__ head_d) {
  __shared__ Lds L;
  __shared__ float4 S[DM::HFT * 4 * FE_THREADS];
  __shared__ float padv[12];               // heads of the representative padded slot: drift 0..2, logits 4..11 (TilePlan)
  __shared__ short order_s[32 * FE_WAVES];
  const int lane = threadIdx.x & 63, wave = steer_wave((A.tiles & 2) != 0, threadIdx.x >> 6, lane, L.red);
  const int tid = 64 * wave + lane, half = lane >> 5;
  load_constants<DM>(A, L, tid);
  __syncthreads();
  WRing<ring_depth_multi<DM>()> ring;
  ring_prime(ring, weight_rsrc(A.P), desc_wx(L, 0), lane * 16);
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    __syncthreads();
    if (tid < DM::ET) L.temb[tid] = temb[(int64_t)b * temb_stride + tid];
    if (tid == 0) L.cnt = (float)A.counts[b];
    const int slot = 32 * wave + (lane & 31);
    const bool valid = slot < A.N;
    const int p = jet_order((A.tiles & 1) != 0, A.maskf + (int64_t)b * A.N, A.N, slot, wave, lane, L.red, order_s);
    const int64_t gp = (int64_t)b * A.N + (valid ? p : 0);
    const float m = valid ? A.maskf[gp] : 0.0f;
    float xp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DM::DC; ++d) xp[d] = valid ? x[gp * DM::DC + d] : 0.0f;
    const int kp = valid ? (int)k[gp] : 0;
    // tiles without a real particle stay idle and take their heads from one representative padded slot, as in the
    // generation loop (every slot of the result is what it was: the callers see padded slots too)
    const TilePlan tp = plan_tiles((A.tiles & 1) != 0, true, m, valid, wave, lane, L.red);
    __syncthreads();
    f32x16 hc, hd;
    NoTrain nt;
    float lg[DM::V];
    float v3[3] = {0.0f, 0.0f, 0.0f};
    if (tp.active) {
      forward_jet<DM, false, true>(A, L, S, xp, kp, m, which, hc, hd, tid, ring, nt);
      if (which != 1) { v3[0] = hc[0]; v3[1] = hc[1]; v3[2] = hc[2]; }
      if (which != 0) gather_logits<DM>(hd, lg, lane);
      if (tp.idle_any) {
        if (tp.rep) {
          if (which != 1) { padv[0] = v3[0]; padv[1] = v3[1]; padv[2] = v3[2]; }
          if (which != 0) {
#pragma unroll
            for (int v = 0; v < DM::V; ++v) padv[4 + v] = lg[v];
          }
        }
        __syncthreads();
      }
    } else {
      forward_jet<DM, false, false>(A, L, S, xp, kp, m, which, hc, hd, tid, ring, nt);
      if (tp.idle_any) {
        __syncthreads();
        v3[0] = padv[0]; v3[1] = padv[1]; v3[2] = padv[2];
#pragma unroll
        for (int v = 0; v < DM::V; ++v) lg[v] = padv[4 + v];
      }
    }
    if (valid && which != 1 && half == 0) {
#pragma unroll
      for (int d = 0; d < DM::DC; ++d) head_c[gp * DM::DC + d] = v3[d];       // features 0..2 = regs 0..2, half 0
    }
    if (valid && which != 0 && half == 0) {
#pragma unroll
      for (int v = 0; v < DM::V; ++v) head_d[gp * DM::V + v] = lg[v];
    }
  }
}

// ---- training forward: both heads + every activation of the tape to HBM ---------------------
template <class DM, int NWT = FE_WAVES>        // (NWT: a name of its own per wave count, the two compilations link together)
__global__ void __launch_bounds__(FE_THREADS, FE_WAVES == 4 ? 2 : 1)
fused_train_forward_kernel(FusedArgs A, const float* __restrict__ x, const int64_t* __restrict__ k,
                           const float* __restrict__ temb, int64_t temb_stride, const SaveSlot* __restrict__ slots,
                           int n_slots, float drop_p, uint64_t seed, uint64_t offset0, float* __restrict__ head_c,
                           float* __restrict__ head_d) {
  __shared__ Lds L;
  __shared__ float4 S[DM::HFT * 4 * FE_THREADS];
  __shared__ short order_s[32 * FE_WAVES];
  // tile t of a jet is the same particles in this kernel and in the reverse one (jet_order is a function of the mask);
  // which hardware wave computes it is this kernel's own business (steer_wave)
  const int lane = threadIdx.x & 63, wave = steer_wave((A.tiles & 2) != 0, threadIdx.x >> 6, lane, L.red);
  const int tid = 64 * wave + lane, half = lane >> 5;
  load_constants<DM>(A, L, tid);
  __syncthreads();
  WRing<ring_depth_multi<DM>()> ring;
  ring_prime(ring, weight_rsrc(A.P), desc_wx(L, 0), lane * 16);
  for (int bi = blockIdx.x; bi < A.B; bi += gridDim.x) {
    const int b = A.jet_order ? A.jet_order[bi] : bi;
    __syncthreads();
    if (tid < DM::ET) L.temb[tid] = temb[(int64_t)b * temb_stride + tid];
    if (tid == 0) L.cnt = (float)A.counts[b];
    const int slot = 32 * wave + (lane & 31);
    const bool valid = slot < A.N;
    const int p = jet_order((A.tiles & 1) != 0, A.maskf + (int64_t)b * A.N, A.N, slot, wave, lane, L.red, order_s);
    const int64_t gp = (int64_t)b * A.N + (valid ? p : 0);
    const float m = valid ? A.maskf[gp] : 0.0f;
    float xp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DM::DC; ++d) xp[d] = valid ? x[gp * DM::DC + d] : 0.0f;
    const int kp = valid ? (int)k[gp] : 0;
    const TilePlan tp = plan_tiles((A.tiles & 1) != 0, false, m, valid, wave, lane, L.red);
    __syncthreads();
    Train tr{slots, n_slots, 0, b, A.N, p, drop_p, seed, offset0};      // rows and dropout keys by the true slot
    tr.dead = (A.tiles & 16) != 0 && m == 0.0f;        // padded slots of working tiles: like the idle tiles' rows, unwritten
    f32x16 hc, hd;
    if (tp.active) forward_jet<DM, false, true>(A, L, S, xp, kp, m, 2, hc, hd, tid, ring, tr);
    else forward_jet<DM, false, false>(A, L, S, xp, kp, m, 2, hc, hd, tid, ring, tr);
    // the heads of an idle tile: padded slots, which the masked losses multiply by zero -- zeros, not stale memory
    if (valid && half == 0) {
#pragma unroll
      for (int d = 0; d < DM::DC; ++d) head_c[gp * DM::DC + d] = tp.active ? hc[d] : 0.0f;
    }
    float lg[DM::V];
    gather_logits<DM>(hd, lg, lane);
    if (valid && half == 0) {
#pragma unroll
      for (int v = 0; v < DM::V; ++v) head_d[gp * DM::V + v] = tp.active ? lg[v] : 0.0f;
    }
  }
}

// ---- the whole generation loop (multimodal_bridge_matching.py:227-259) --------------------
__device__ __forceinline__ float thermo_w_f(int kind, float c, float t0, float t1) { return thermo_w(kind, c, t0, t1); }

// telegraph rates (bridges.py:137-163) + tau-leap jump / final argmax (solvers.py:59-81, matching.py:253-255)
// on the V=8 logits of one particle; shared by the multimodal and the discrete-only loops.
__device__ __forceinline__ int jump_step(float (&r)[8], int kp, float tt, const GenArgs& Gn, bool last, bool valid,
                                         int B, int N, int b, int p, int s, int64_t gp) {
  constexpr int V = 8;
  const float w = thermo_w(Gn.thermo_kind, Gn.thermo_c, tt, 1.0f);
  const float Bc = (w * (float)V) / (1.0f - w);
  float mx = -INFINITY;
#pragma unroll
  for (int v = 0; v < V; ++v) mx = fmaxf(mx, r[v]);
  float sum = 0.0f;
#pragma unroll
  for (int v = 0; v < V; ++v) { r[v] = expf(r[v] - mx); sum += r[v]; }
  float qk = 0.0f;
#pragma unroll
  for (int v = 0; v < V; ++v) { r[v] = r[v] / sum; qk = (v == kp) ? r[v] : qk; }
#pragma unroll
  for (int v = 0; v < V; ++v) r[v] = 1.0f + Bc * r[v] + w * qk;
  if (last) {
    int best = 0;
    float bv = r[0];
#pragma unroll
    for (int v = 1; v < V; ++v) if (r[v] > bv) { bv = r[v]; best = v; }
    return best;
  }
  if (Gn.delta_n != nullptr) {
    int knew = kp;
    if (valid) {
      const float* dn = Gn.delta_n + (((int64_t)s * B + b) * N + p) * V;
      float tot = 0.0f, net = 0.0f;
#pragma unroll
      for (int v = 0; v < V; ++v) { const float d = dn[v]; tot += d; net += d * (float)(v - kp); }
      const int64_t ok = ((int64_t)tot <= 1) ? 1 : 0;
      int64_t moved = (int64_t)kp + (int64_t)net * ok;
      moved %= V;
      if (moved < 0) moved += V;
      knew = (int)moved;
    }
    return knew;
  }
  uint32_t q[4];
  philox4(Gn.seed, Gn.offset + (uint64_t)s, (uint64_t)gp, 0xC001u, q);
  const float u = u01(q[0]);
  float Lam = 0.0f;
#pragma unroll
  for (int v = 0; v < V; ++v) Lam += r[v] * Gn.dt;
  const float e = expf(-Lam);
  float acc = 0.0f;
  int hit = -1;
#pragma unroll
  for (int v = 0; v < V; ++v) { acc += r[v] * Gn.dt * e; if (hit < 0 && u < acc) hit = v; }
  return hit >= 0 ? hit : kp;
}

// CH = 2 (four-wave build, 128 < N <= 160): the jets whose real particles fit four tiles -- fewer than 128 of them:
// 99.97 % of AOJ's 150-slot clouds -- run here, two to a CU, instead of in the five-wave build at one per CU.  Their
// slots beyond the four tiles are padded slots without a wave of their own: every lane of wave 0 carries one of them
// as a shadow (ordered slot + 128), state in LDS, moved by the representative's drift and logits like the slots of an
// idle tile.  The few longer jets are left to the five-wave launch of the same call (bit 2 of Gn.tiles: a split call).
template <class DM, int NWT = FE_WAVES, int CH = 1>       // (NWT: a name of its own per wave count, the two compilations link together)
__global__ void __launch_bounds__(FE_THREADS, FE_WAVES == 4 ? 2 : 1)
fused_generate_kernel(FusedArgs A, GenArgs Gn, float* __restrict__ x, int64_t* __restrict__ k) {
  __shared__ Lds L;
  __shared__ float4 S[DM::HFT * 4 * FE_THREADS];
  __shared__ float park_s[4][FE_THREADS];  // x (3) and k of this lane's particle while a forward runs
  __shared__ float padv[12];               // drift (0..2) and logits (4..11) of the representative padded slot
  // (two workgroups of this kernel fill a CU's 160 KB of LDS to the last KB: the slot map in bytes, and the shadow
  // slots -- at most 32, all carried by wave 0, both lane halves holding the same values -- in 512 B)
  __shared__ unsigned char order_s[32 * FE_WAVES * CH];
  __shared__ float park2_s[CH > 1 ? 4 : 1][CH > 1 ? 32 : 1];             // state of the shadow slots, by lane & 31
  stagger_start(Gn.stagger, threadIdx.x, L.red);
  const int lane = threadIdx.x & 63, wave = steer_wave((Gn.tiles & 2) != 0, threadIdx.x >> 6, lane, L.red);
  const int tid = 64 * wave + lane, half = lane >> 5;
  constexpr int V = DM::V;
#ifdef MMB_FE_PROF
  if (tid == 0) L.tlast = clock64();
#endif
  load_constants<DM>(A, L, tid);
  __syncthreads();
  WRing<ring_depth_multi<DM>()> ring;
  ring_prime(ring, weight_rsrc(A.P), desc_wx(L, 0), lane * 16);
  for (int bi = blockIdx.x; bi < Gn.n_jets; bi += gridDim.x) {
    const int b = Gn.jet_order ? Gn.jet_order[bi] : bi;
    // a split call: jets of 128 and more particles need a fifth tile (or the representative padded slot in it)
    if ((Gn.tiles & 4) && (A.counts[b] >= 128) != (FE_WAVES == 5)) continue;
    __syncthreads();
    if (tid == 0) L.cnt = (float)A.counts[b];
    const int slot = 32 * wave + (lane & 31);
    const bool valid = slot < A.N;
    int pp[CH];
    jet_order_n<CH, unsigned char>((Gn.tiles & 1) != 0, A.maskf + (int64_t)b * A.N, A.N, wave, lane, L.red, order_s, pp);
    const int p = pp[0];
    const int p2 = pp[CH - 1];
    const bool valid2 = CH > 1 && wave == 0 && slot + 32 * FE_WAVES < A.N;
    const int l2 = lane & 31;
    const int64_t gp2 = (int64_t)b * A.N + (valid2 ? p2 : 0);
    if constexpr (CH > 1) {
      if (valid2) {
#pragma unroll
        for (int d = 0; d < DM::DC; ++d) park2_s[d][l2] = x[gp2 * DM::DC + d];
        park2_s[3][l2] = __int_as_float((int)k[gp2]);
      }
    }
    const int64_t gp = (int64_t)b * A.N + (valid ? p : 0);
    const float m = valid ? A.maskf[gp] : 0.0f;
    float xp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DM::DC; ++d) xp[d] = valid ? x[gp * DM::DC + d] : 0.0f;
    int kp = valid ? (int)k[gp] : 0;
    const TilePlan tp = plan_tiles((Gn.tiles & 1) != 0, true, m, valid, wave, lane, L.red);
    const bool pub = tp.idle_any || CH > 1;                  // somebody needs the representative's heads
    // The whole grid loop once per kind of tile (same barriers in both): a working tile runs the loop it ran before
    // tiles could be idle, with that loop's register allocation; the idle one is a short loop of its own.
    auto grid_loop = [&](auto actc) __attribute__((always_inline)) {
    constexpr bool ACT = decltype(actc)::v != 0;
    for (int s = 0; s < Gn.steps; ++s) {
      __syncthreads();
      if (tid < DM::ET) L.temb[tid] = Gn.temb_steps[(int64_t)s * DM::ET + tid];
      __syncthreads();
      A.ctab = Gn.ctab_steps + (int64_t)s * Gn.n_ctab * 128;
      const float tt = Gn.times[s];
      f32x16 hc, hd;
      // The particle state is only needed at the two ends of a forward: it waits in LDS meanwhile
      // (each lane reads back its own slot), which keeps it out of the register allocator's way.
      auto park = [&]() {
#pragma unroll
        for (int d = 0; d < DM::DC; ++d) park_s[d][tid] = xp[d];
        park_s[3][tid] = __int_as_float(kp);
      };
      auto unpark = [&]() {
#pragma unroll
        for (int d = 0; d < DM::DC; ++d) xp[d] = park_s[d][tid];
        kp = __float_as_int(park_s[3][tid]);
      };
      // Euler step on the drift head (solvers.py:21-25); padded slots move too (K3)
      park();
      NoTrain nt;
      forward_jet<DM, true, ACT>(A, L, S, xp, kp, m, 0, hc, hd, tid, ring, nt);
      unpark();
      {
        float v[3] = {0.0f, 0.0f, 0.0f};
        if constexpr (ACT) {
          v[0] = hc[0]; v[1] = hc[1]; v[2] = hc[2];
          if (pub) {                                     // (workgroup-uniform; an idle tile exists only when it is set)
            if (tp.rep) { padv[0] = v[0]; padv[1] = v[1]; padv[2] = v[2]; }
            __syncthreads();
          }
        } else if (pub) {                                // (an idle tile without a valid slot needs nothing)
          __syncthreads();
          v[0] = padv[0]; v[1] = padv[1]; v[2] = padv[2];
        }
        const float sig = Gn.em ? Gn.em_sigma * sqrtf(tt * (1.0f - tt)) : 0.0f;
        auto wiener = [&](float (&w)[4], bool ok, int pt, int64_t gpt) {      // both lane halves draw the same increments
          w[0] = w[1] = w[2] = w[3] = 0.0f;
          if (!Gn.em) return;
          if (Gn.delta_w != nullptr) {
            if (ok) {
              const float* dw = Gn.delta_w + (((int64_t)s * A.B + b) * A.N + pt) * DM::DC;
#pragma unroll
              for (int d = 0; d < DM::DC; ++d) w[d] = dw[d];
            }
          } else {
            uint32_t q[4];
            philox4(Gn.seed, Gn.offset + (uint64_t)s, (uint64_t)gpt, 0xE001u, q);
            // hardware v_sin/v_cos (argument in revolutions): distribution-level draws, and no
            // private-memory argument reduction inside the generation loop
            const float r0 = sqrtf(-2.0f * __logf(u01_open0(q[0]))), r1 = sqrtf(-2.0f * __logf(u01_open0(q[2])));
            w[0] = r0 * __builtin_amdgcn_cosf(u01(q[1]));
            w[1] = r0 * __builtin_amdgcn_sinf(u01(q[1]));
            w[2] = r1 * __builtin_amdgcn_cosf(u01(q[3]));
          }
        };
        float w[4];
        wiener(w, valid, p, gp);
#pragma unroll
        for (int d = 0; d < DM::DC; ++d) {
          const float o = __shfl_xor(v[d], 32, 64);
          const float drift = half == 0 ? v[d] : o;
          xp[d] += Gn.em ? (Gn.dt * drift + sig * w[d]) : Gn.dt * drift;
        }
        if constexpr (CH > 1) {                          // the shadow slot: a padded slot, the representative's drift
          if (valid2) {
            float w2[4];
            wiener(w2, true, p2, gp2);
#pragma unroll
            for (int d = 0; d < DM::DC; ++d)
              park2_s[d][l2] += Gn.em ? (Gn.dt * padv[d] + sig * w2[d]) : Gn.dt * padv[d];
          }
        }
      }
      __syncthreads();
      // tau-leap on the logits head, which sees the UPDATED x at the same t (solvers.py:59-81)
      park();
      forward_jet<DM, true, ACT>(A, L, S, xp, kp, m, 1, hc, hd, tid, ring, nt);
      unpark();
      float r[V];
      if constexpr (ACT) {
        gather_logits<DM>(hd, r, lane);
        if (pub) {
          if (tp.rep) {
#pragma unroll
            for (int v = 0; v < V; ++v) padv[4 + v] = r[v];
          }
          __syncthreads();
        }
      } else {
        if (pub) __syncthreads();
#pragma unroll
        for (int v = 0; v < V; ++v) r[v] = padv[4 + v];
      }
      kp = jump_step(r, kp, tt, Gn, s == Gn.steps - 1, valid, A.B, A.N, b, p, s, gp);
      if constexpr (CH > 1) {
        if (valid2) {
          float r2[V];
#pragma unroll
          for (int v = 0; v < V; ++v) r2[v] = padv[4 + v];
          const int k2 = jump_step(r2, __float_as_int(park2_s[3][l2]), tt, Gn, s == Gn.steps - 1, true, A.B, A.N, b, p2, s, gp2);
          park2_s[3][l2] = __int_as_float(k2);
        }
      }
    }
    };
    if (tp.active) grid_loop(IF<1>{});
    else grid_loop(IF<0>{});
    if (valid && half == 0) {
#pragma unroll
      for (int d = 0; d < DM::DC; ++d) x[gp * DM::DC + d] = xp[d];
      k[gp] = kp;
    }
    if constexpr (CH > 1) {
      if (valid2 && half == 0) {
#pragma unroll
        for (int d = 0; d < DM::DC; ++d) x[gp2 * DM::DC + d] = park2_s[d][l2];
        k[gp2] = __float_as_int(park2_s[3][l2]);
      }
    }
  }
}

// ==================================================================== UniModalEPiC ====
// Discrete-only model (multimodal_epic.py:44-64): ONE EPiC encoder with projection on
// cat[t, t, trunc(k_emb*mask)] (the time embedding enters twice: multimodal_epic.py:47-54 +
// epic.py:155 -- both column blocks are summed into one time vector on the host), whose output
// layer emits the V logits directly (no heads).

// Training (TR = Train) slots: projection 9, layers 8 each (the V logits of the output layer go to the caller).
template <class DM, bool CT, bool ACT = true, class TR>
__device__ __forceinline__ void forward_uni(const FusedArgs& A, Lds& L, float4* S, const float* tab, int kp, float m,
                                            f32x16& hd, int tid, WRing<ring_depth_uni<DM>()>& ring, TR& nt) {
  constexpr int HT = DM::HT, RD = ring_depth_uni<DM>();
  typedef Epic<HT, DM::ET, DM::G, RD, CT> E;
  const int lane = tid & 63, half = lane >> 5;
  FCtxT<ACT> c{A.P, weight_rsrc(A.P), L, m, tid, lane, tid >> 6, half, lane * 16, A.ctab};
  int li = 0, ci = 0;
  FE_PROF(L, PF_STEP);
  f32x16 td[DM::EDT];
  if constexpr (ACT) ffor<DM::EDT>([&](auto tc) { constexpr int t = decltype(tc)::v;
    ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
      const float4 e = *reinterpret_cast<const float4*>(tab + kp * DM::EDP + 32 * t + 8 * q + 4 * half);
      td[t][4 * q + 0] = truncf(e.x * m); td[t][4 * q + 1] = truncf(e.y * m);      // quirk Q1
      td[t][4 * q + 2] = truncf(e.z * m); td[t][4 * q + 3] = truncf(e.w * m); }); });
  FE_PROF(L, PF_EMBED);
  f32x16 x[HT];
  const int nb = A.nb_branch;
  const int wx_after_proj = desc_wx(L, nb > 0 ? 4 + 2 : 4);
  E::template project<DM::EDT>(x, [&](auto fc) -> const f32x16& { return td[decltype(fc)::v]; }, c, li, ci, wx_after_proj,
                               ring, nt);
  if (tid < DM::G) L.gskip[tid] = L.gcur[tid];
  if constexpr (ACT) ffor<HT>([&](auto t) { park_tile(S, decltype(t)::v, x[decltype(t)::v], tid); });
  for (int i = 0; i < nb; ++i) {
    const int nx = desc_wx(L, i + 1 < nb ? li + 4 + 2 : li + 4);
    E::layer(x, [&](auto t) -> f32x16 { return fetch_tile(S, decltype(t)::v, tid); }, A.use_skip != 0, c, li, ci, nx, ring, nt);
  }
  f32x16 o[1];
  E::template output<1, DM::V, true>(o, x, c, li, ci, desc_wx(L, 0), ring, nt);
  hd = o[0];
}

template <class DM>
__device__ __forceinline__ void load_table_uni(const FusedArgs& A, float* tab, int tid) {
  for (int i = tid; i < DM::V * DM::EDP; i += FE_THREADS) {
    const int v = i / DM::EDP, e = i % DM::EDP;
    tab[i] = e < DM::ED ? A.P[A.emb_table + v * DM::ED + e] : 0.0f;
  }
}

// gather V=8 logits of this lane's particle from an output tile (features 0..3 half 0, 4..7 half 1)
__device__ __forceinline__ void gather8(const f32x16& hd, float (&lg)[8], int lane) {
  const int half = lane >> 5;
  float mine[4] = {hd[0], hd[1], hd[2], hd[3]};
  float other[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) other[i] = __shfl_xor(mine[i], 32, 64);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lg[i] = half == 0 ? mine[i] : other[i];
    lg[4 + i] = half == 0 ? other[i] : mine[i];
  }
}

template <class DM, int NWT = FE_WAVES>        // (NWT: a name of its own per wave count, the two compilations link together)
__global__ void __launch_bounds__(FE_THREADS, FE_WAVES == 4 ? 2 : 1)
fused_uni_forward_kernel(FusedArgs A, const int64_t* __restrict__ k, const float* __restrict__ temb,
                         int64_t temb_stride, float* __restrict__ head_d) {
  static_assert(DM::V == 8, "fused path is instantiated for vocab_size 8");
  __shared__ Lds L;
  __shared__ float4 S[DM::HT * 4 * FE_THREADS];
  __shared__ __attribute__((aligned(16))) float tab[DM::V * DM::EDP];
  __shared__ float padv[8];                // logits of the representative padded slot (TilePlan)
  __shared__ short order_s[32 * FE_WAVES];
  const int lane = threadIdx.x & 63, wave = steer_wave((A.tiles & 2) != 0, threadIdx.x >> 6, lane, L.red);
  const int tid = 64 * wave + lane, half = lane >> 5;
  load_table_uni<DM>(A, tab, tid);
  load_desc(A, L, tid);
  __syncthreads();
  WRing<ring_depth_uni<DM>()> ring;
  ring_prime(ring, weight_rsrc(A.P), desc_wx(L, 0), lane * 16);
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    __syncthreads();
    if (tid < DM::ET) L.temb[tid] = temb[(int64_t)b * temb_stride + tid];
    if (tid == 0) L.cnt = (float)A.counts[b];
    const int slot = 32 * wave + (lane & 31);
    const bool valid = slot < A.N;
    const int p = jet_order((A.tiles & 1) != 0, A.maskf + (int64_t)b * A.N, A.N, slot, wave, lane, L.red, order_s);
    const int64_t gp = (int64_t)b * A.N + (valid ? p : 0);
    const float m = valid ? A.maskf[gp] : 0.0f;
    const int kp = valid ? (int)k[gp] : 0;
    const TilePlan tp = plan_tiles((A.tiles & 1) != 0, true, m, valid, wave, lane, L.red);      // (as fused_forward_kernel)
    __syncthreads();
    f32x16 hd;
    NoTrain nt;
    float lg[8];
    if (tp.active) {
      forward_uni<DM, false, true>(A, L, S, tab, kp, m, hd, tid, ring, nt);
      gather8(hd, lg, lane);
      if (tp.idle_any) {
        if (tp.rep) {
#pragma unroll
          for (int v = 0; v < 8; ++v) padv[v] = lg[v];
        }
        __syncthreads();
      }
    } else {
      forward_uni<DM, false, false>(A, L, S, tab, kp, m, hd, tid, ring, nt);
      if (tp.idle_any) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 8; ++v) lg[v] = padv[v];
      }
    }
    if (valid && half == 0) {
#pragma unroll
      for (int v = 0; v < 8; ++v) head_d[gp * 8 + v] = lg[v];
    }
  }
}

// training forward of the discrete-only model: logits + every activation of the tape to HBM
template <class DM, int NWT = FE_WAVES>        // (NWT: a name of its own per wave count, the two compilations link together)
__global__ void __launch_bounds__(FE_THREADS, FE_WAVES == 4 ? 2 : 1)
fused_uni_train_forward_kernel(FusedArgs A, const int64_t* __restrict__ k, const float* __restrict__ temb,
                               int64_t temb_stride, const SaveSlot* __restrict__ slots, int n_slots, float drop_p,
                               uint64_t seed, uint64_t offset0, float* __restrict__ head_d) {
  static_assert(DM::V == 8, "fused path is instantiated for vocab_size 8");
  __shared__ Lds L;
  __shared__ float4 S[DM::HT * 4 * FE_THREADS];
  __shared__ __attribute__((aligned(16))) float tab[DM::V * DM::EDP];
  __shared__ short order_s[32 * FE_WAVES];
  const int lane = threadIdx.x & 63, wave = steer_wave((A.tiles & 2) != 0, threadIdx.x >> 6, lane, L.red);
  const int tid = 64 * wave + lane, half = lane >> 5;
  load_table_uni<DM>(A, tab, tid);
  load_desc(A, L, tid);
  __syncthreads();
  WRing<ring_depth_uni<DM>()> ring;
  ring_prime(ring, weight_rsrc(A.P), desc_wx(L, 0), lane * 16);
  for (int bi = blockIdx.x; bi < A.B; bi += gridDim.x) {
    const int b = A.jet_order ? A.jet_order[bi] : bi;
    __syncthreads();
    if (tid < DM::ET) L.temb[tid] = temb[(int64_t)b * temb_stride + tid];
    if (tid == 0) L.cnt = (float)A.counts[b];
    const int slot = 32 * wave + (lane & 31);
    const bool valid = slot < A.N;
    const int p = jet_order((A.tiles & 1) != 0, A.maskf + (int64_t)b * A.N, A.N, slot, wave, lane, L.red, order_s);
    const int64_t gp = (int64_t)b * A.N + (valid ? p : 0);
    const float m = valid ? A.maskf[gp] : 0.0f;
    const int kp = valid ? (int)k[gp] : 0;
    const TilePlan tp = plan_tiles((A.tiles & 1) != 0, false, m, valid, wave, lane, L.red);
    __syncthreads();
    Train tr{slots, n_slots, 0, b, A.N, p, drop_p, seed, offset0};      // rows and dropout keys by the true slot
    tr.dead = (A.tiles & 16) != 0 && m == 0.0f;        // padded slots of working tiles: like the idle tiles' rows, unwritten
    f32x16 hd;
    if (tp.active) forward_uni<DM, false, true>(A, L, S, tab, kp, m, hd, tid, ring, tr);
    else forward_uni<DM, false, false>(A, L, S, tab, kp, m, hd, tid, ring, tr);
    float lg[8];
    gather8(hd, lg, lane);
    if (valid && half == 0) {
#pragma unroll
      for (int v = 0; v < 8; ++v) head_d[gp * 8 + v] = tp.active ? lg[v] : 0.0f;      // (idle tile: padded slots, masked loss)
    }
  }
}

// (CH = 2: 128 < N <= 160 on four waves with shadow slots, as fused_generate_kernel)
template <class DM, int NWT = FE_WAVES, int CH = 1>       // (NWT: a name of its own per wave count, the two compilations link together)
__global__ void __launch_bounds__(FE_THREADS, FE_WAVES == 4 ? 2 : 1)
fused_uni_generate_kernel(FusedArgs A, GenArgs Gn, int64_t* __restrict__ k) {
  __shared__ Lds L;
  __shared__ float4 S[DM::HT * 4 * FE_THREADS];
  __shared__ __attribute__((aligned(16))) float tab[DM::V * DM::EDP];
  __shared__ float padv[8];                // logits of the representative padded slot (TilePlan)
  __shared__ unsigned char order_s[32 * FE_WAVES * CH];
  __shared__ int park2_k[CH > 1 ? 32 : 1]; // k of the shadow slots, by lane & 31
  stagger_start(Gn.stagger, threadIdx.x, L.red);
  const int lane = threadIdx.x & 63, wave = steer_wave((Gn.tiles & 2) != 0, threadIdx.x >> 6, lane, L.red);
  const int tid = 64 * wave + lane, half = lane >> 5;
#ifdef MMB_FE_PROF
  if (tid == 0) L.tlast = clock64();
#endif
  load_table_uni<DM>(A, tab, tid);
  load_desc(A, L, tid);
  __syncthreads();
  WRing<ring_depth_uni<DM>()> ring;
  ring_prime(ring, weight_rsrc(A.P), desc_wx(L, 0), lane * 16);
  for (int bi = blockIdx.x; bi < Gn.n_jets; bi += gridDim.x) {
    const int b = Gn.jet_order ? Gn.jet_order[bi] : bi;
    if ((Gn.tiles & 4) && (A.counts[b] >= 128) != (FE_WAVES == 5)) continue;      // a split call: see fused_generate_kernel
    __syncthreads();
    if (tid == 0) L.cnt = (float)A.counts[b];
    const int slot = 32 * wave + (lane & 31);
    const bool valid = slot < A.N;
    int pp[CH];
    jet_order_n<CH, unsigned char>((Gn.tiles & 1) != 0, A.maskf + (int64_t)b * A.N, A.N, wave, lane, L.red, order_s, pp);
    const int p = pp[0];
    const int p2 = pp[CH - 1], l2 = lane & 31;
    const bool valid2 = CH > 1 && wave == 0 && slot + 32 * FE_WAVES < A.N;
    const int64_t gp2 = (int64_t)b * A.N + (valid2 ? p2 : 0);
    if constexpr (CH > 1) {
      if (valid2) park2_k[l2] = (int)k[gp2];
    }
    const int64_t gp = (int64_t)b * A.N + (valid ? p : 0);
    const float m = valid ? A.maskf[gp] : 0.0f;
    int kp = valid ? (int)k[gp] : 0;
    const TilePlan tp = plan_tiles((Gn.tiles & 1) != 0, true, m, valid, wave, lane, L.red);
    const bool pub = tp.idle_any || CH > 1;
    auto grid_loop = [&](auto actc) __attribute__((always_inline)) {      // (the whole loop per kind of tile: see fused_generate_kernel)
    constexpr bool ACT = decltype(actc)::v != 0;
    for (int s = 0; s < Gn.steps; ++s) {
      __syncthreads();
      if (tid < DM::ET) L.temb[tid] = Gn.temb_steps[(int64_t)s * DM::ET + tid];
      __syncthreads();
      A.ctab = Gn.ctab_steps + (int64_t)s * Gn.n_ctab * 128;
      f32x16 hd;
      NoTrain nt;
      forward_uni<DM, true, ACT>(A, L, S, tab, kp, m, hd, tid, ring, nt);
      float r[8];
      if constexpr (ACT) {
        gather8(hd, r, lane);
        if (pub) {
          if (tp.rep) {
#pragma unroll
            for (int v = 0; v < 8; ++v) padv[v] = r[v];
          }
          __syncthreads();
        }
      } else {
        if (pub) __syncthreads();
#pragma unroll
        for (int v = 0; v < 8; ++v) r[v] = padv[v];
      }
      kp = jump_step(r, kp, Gn.times[s], Gn, s == Gn.steps - 1, valid, A.B, A.N, b, p, s, gp);
      if constexpr (CH > 1) {
        if (valid2) {
          float r2[8];
#pragma unroll
          for (int v = 0; v < 8; ++v) r2[v] = padv[v];
          park2_k[l2] = jump_step(r2, park2_k[l2], Gn.times[s], Gn, s == Gn.steps - 1, true, A.B, A.N, b, p2, s, gp2);
        }
      }
    }
    };
    if (tp.active) grid_loop(IF<1>{});
    else grid_loop(IF<0>{});
    if (valid && half == 0) k[gp] = kp;
    if constexpr (CH > 1) {
      if (valid2 && half == 0) k[gp2] = park2_k[l2];
    }
  }
}



// MMB_STAGGER=<cycles> (developer switch; default: see stagger_default) -- start offset of the second jet of a CU
static int stagger_cycles(int dflt) {
  const char* e = getenv("MMB_STAGGER");
  return e ? atoi(e) : dflt;
}
// MMB_GEN_TILES=<bits> (developer switch; default 3): bit 0 idle tiles of padded slots, bit 1 tile-to-SIMD steering
static int gen_tiles() {
  const char* e = getenv("MMB_GEN_TILES");
  return e ? atoi(e) : 3;
}
// MMB_TRAIN_TILES=<bits> (developer switch; default 3): the same two bits for the training kernels (forward and
// reverse read it per call: both must run under the same setting, they share the tile order)
// MMB_GEN_SPLIT=0 (developer switch): 128 < N <= 160 runs every jet in the five-wave build (no shadow slots)
static int gen_split() {
  const char* e = getenv("MMB_GEN_SPLIT");
  return e ? atoi(e) : 1;
}
#if MMB_FE_WAVES == 4
// the library's side stream of the current device (created at first use) and the two events that fence it
struct SideStream { hipStream_t s = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr; };
static SideStream* side_stream() {
  static SideStream sides[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  SideStream& sd = sides[dev];
  if (sd.s == nullptr &&
      !(hipStreamCreateWithFlags(&sd.s, hipStreamNonBlocking) == hipSuccess &&
        hipEventCreateWithFlags(&sd.e0, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&sd.e1, hipEventDisableTiming) == hipSuccess)) {
    sd.s = nullptr;
    return nullptr;
  }
  return &sd;
}
#endif
// MMB_FWD_TILES=<bits> (developer switch; default 3): the same two bits for the inference forward kernels
static int fwd_tiles() {
  const char* e = getenv("MMB_FWD_TILES");
  return e ? atoi(e) : 3;
}
static int train_tiles() {
  const char* e = getenv("MMB_TRAIN_TILES");
  int tiles = e ? atoi(e) : 3;
  // bit 4: the padded slots of WORKING tiles leave their tape rows unwritten too (the idle tiles' contract, per lane:
  // nothing multiplies those rows by anything but exact zeros; MMB_FWD_DEAD_LANES=0: developer switch)
  const char* d = getenv("MMB_FWD_DEAD_LANES");
  if ((tiles & 1) && (!d || atoi(d) != 0)) tiles |= 16;
  return tiles;
}
static int check_fused(const char* name, int B, int N, const void* P, const void* D, const void* maskf,
                       const void* counts, int n_desc) {
  MMB_REQUIRE(B > 0 && N > 0 && N <= 32 * FE_WAVES, "%s: needs 0 < N <= %d (got B=%d N=%d)", name, 32 * FE_WAVES, B, N);
  MMB_REQUIRE(P && D && maskf && counts, "%s: NULL argument", name);
  MMB_REQUIRE(n_desc > 0 && n_desc <= MAX_DESC, "%s: %d layer descriptors, the kernels cache at most %d (fewer blocks, "
              "or the layered path)", name, n_desc, MAX_DESC);
  return MMB_OK;
}
// descriptors of a packed model (mmb_hip/fused.py: _pack_multi / _pack_uni): projection 4, EPiC layer 4, output 1, heads 4
static int n_desc_multi(int nb0, int nb1, int nbf) { return (4 + 4 * nb0 + 1) + (4 + 4 * nb1 + 1) + (4 * nbf + 1) + 4; }
static int n_desc_uni(int nb) { return 4 + 4 * nb + 1; }

}  // namespace mmb

using namespace mmb;

#if MMB_FE_WAVES == 4
// the same entry points compiled for five waves per jet (fused_epic_w5.hip): 128 < N <= 160
extern "C" int mmb_fused_epic_forward_w5(int, int, int, const float*, const int32_t*, int, int, int, int, int, int, int,
                                         const float*, const int32_t*, const float*, const int64_t*, const float*, int64_t,
                                         int, float*, float*, void*);
extern "C" int mmb_fused_uni_forward_w5(int, int, int, const float*, const int32_t*, int, int, int, const float*,
                                        const int32_t*, const int64_t*, const float*, int64_t, float*, void*);
extern "C" int mmb_fused_uni_generate_w5(int, int, int, const float*, const int32_t*, int, int, int, const float*,
                                         const int32_t*, int64_t*, int, const float*, const float*, const float*, int, float,
                                         int, float, const float*, uint64_t, uint64_t, const int32_t*, const int32_t*, int, void*);
extern "C" int mmb_fused_epic_generate_w5(int, int, int, const float*, const int32_t*, int, int, int, int, int, int, int,
                                          const float*, const int32_t*, float*, int64_t*, int, const float*, const float*,
                                          const float*, int, float, int, float, const float*, uint64_t, uint64_t, int, float,
                                          const float*, const int32_t*, const int32_t*, int, void*);
extern "C" int mmb_fused_epic_train_forward_w5(int, int, int, const float*, const int32_t*, int, int, int, int, int, int, int,
                                               const float*, const int32_t*, const float*, const int64_t*, const float*,
                                               int64_t, const void*, int, float, uint64_t, uint64_t, float*, float*,
                                               const int32_t*, void*);
extern "C" int mmb_fused_uni_train_forward_w5(int, int, int, const float*, const int32_t*, int, int, int, const float*,
                                              const int32_t*, const int64_t*, const float*, int64_t, const void*, int, float,
                                              uint64_t, uint64_t, float*, const int32_t*, void*);
#endif

// variant: 0 = S3 dims (Et32 Ec16 Ed16 H64/64/128 G16/16/32), 1 = "mid" dims (Et16 Ec8 Ed8 H32/32/64 G8/8/16)
extern "C" int mmb_fused_epic_forward(int variant, int B, int N, const float* packed, const int32_t* desc,
                                      int nb_branch, int nb_branch1, int nb_fused_exec, int use_skip, int off_cw, int off_cb,
                                      int off_table, const float* maskf, const int32_t* counts, const float* x,
                                      const int64_t* k, const float* temb, int64_t temb_stride, int which,
                                      float* head_c, float* head_d, void* stream) {
#if MMB_FE_WAVES == 4
  if (N > 128) return mmb_fused_epic_forward_w5(variant, B, N, packed, desc, nb_branch, nb_branch1, nb_fused_exec, use_skip, off_cw, off_cb,
                                                off_table, maskf, counts, x, k, temb, temb_stride, which, head_c, head_d, stream);
#endif
  int rc = check_fused("mmb_fused_epic_forward", B, N, packed, desc, maskf, counts, n_desc_multi(nb_branch, nb_branch1, nb_fused_exec));
  if (rc) return rc;
  MMB_REQUIRE(variant >= 0 && variant < N_MULTI_VARIANTS, "mmb_fused_epic_forward: unknown variant %d", variant);
  MMB_REQUIRE(x && k && temb, "mmb_fused_epic_forward: NULL state");
  MMB_REQUIRE(which >= 0 && which <= 2, "mmb_fused_epic_forward: which=%d", which);
  if (which != 1) MMB_REQUIRE(head_c, "mmb_fused_epic_forward: head_c is NULL");
  if (which != 0) MMB_REQUIRE(head_d, "mmb_fused_epic_forward: head_d is NULL");
  FusedArgs A{nullptr, packed, reinterpret_cast<const LDesc*>(desc), nb_branch, nb_fused_exec, use_skip, off_cw, off_cb,
              off_table, B, N, maskf, counts, nb_branch1, n_desc_multi(nb_branch, nb_branch1, nb_fused_exec), fwd_tiles()};
  const int grid = B < 4096 ? B : 4096;
  switch (variant) {
    case 0: hipLaunchKernelGGL(fused_forward_kernel<DimsS3>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, x, k, temb,
                       temb_stride, which, head_c, head_d); break;
    case 1: hipLaunchKernelGGL(fused_forward_kernel<DimsMid>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, x, k, temb,
                       temb_stride, which, head_c, head_d); break;
    case 2: hipLaunchKernelGGL(fused_forward_kernel<DimsWide>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, x, k, temb,
                       temb_stride, which, head_c, head_d); break;
    default: hipLaunchKernelGGL(fused_forward_kernel<DimsNarrow>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, x, k, temb,
                       temb_stride, which, head_c, head_d); break;
  }
  return check_launch("mmb_fused_epic_forward");
}

extern "C" int mmb_fused_epic_train_forward(int variant, int B, int N, const float* packed, const int32_t* desc,
                                            int nb_branch, int nb_branch1, int nb_fused_exec, int use_skip, int off_cw, int off_cb,
                                            int off_table, const float* maskf, const int32_t* counts, const float* x,
                                            const int64_t* k, const float* temb, int64_t temb_stride,
                                            const void* slots, int n_slots, float drop_p, uint64_t seed,
                                            uint64_t offset0, float* head_c, float* head_d, const int32_t* jet_order,
                                            void* stream) {
#if MMB_FE_WAVES == 4
  if (N > 128) return mmb_fused_epic_train_forward_w5(variant, B, N, packed, desc, nb_branch, nb_branch1, nb_fused_exec, use_skip,
                                                      off_cw, off_cb, off_table, maskf, counts, x, k, temb, temb_stride, slots,
                                                      n_slots, drop_p, seed, offset0, head_c, head_d, jet_order, stream);
#endif
  int rc = check_fused("mmb_fused_epic_train_forward", B, N, packed, desc, maskf, counts, n_desc_multi(nb_branch, nb_branch1, nb_fused_exec));
  if (rc) return rc;
  MMB_REQUIRE(variant >= 0 && variant < N_MULTI_VARIANTS, "mmb_fused_epic_train_forward: unknown variant %d", variant);
  MMB_REQUIRE(x && k && temb && head_c && head_d && slots, "mmb_fused_epic_train_forward: NULL argument");
  MMB_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "mmb_fused_epic_train_forward: dropout p=%f outside [0,1)", drop_p);
  const int expect = 2 * 10 + 8 * (nb_branch + nb_branch1) + 1 + 8 * nb_fused_exec + 1 + 4;
  MMB_REQUIRE(n_slots == expect, "mmb_fused_epic_train_forward: %d save slots given, the architecture has %d", n_slots,
              expect);
  FusedArgs A{nullptr, packed, reinterpret_cast<const LDesc*>(desc), nb_branch, nb_fused_exec, use_skip, off_cw, off_cb,
              off_table, B, N, maskf, counts, nb_branch1, n_desc_multi(nb_branch, nb_branch1, nb_fused_exec), train_tiles(),
              jet_order};
  const SaveSlot* tab = reinterpret_cast<const SaveSlot*>(slots);
  const int grid = B < 4096 ? B : 4096;
  switch (variant) {
    case 0: hipLaunchKernelGGL(fused_train_forward_kernel<DimsS3>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, x, k,
                       temb, temb_stride, tab, n_slots, drop_p, seed, offset0, head_c, head_d); break;
    case 1: hipLaunchKernelGGL(fused_train_forward_kernel<DimsMid>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, x, k,
                       temb, temb_stride, tab, n_slots, drop_p, seed, offset0, head_c, head_d); break;
    case 2: hipLaunchKernelGGL(fused_train_forward_kernel<DimsWide>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, x, k,
                       temb, temb_stride, tab, n_slots, drop_p, seed, offset0, head_c, head_d); break;
    default: hipLaunchKernelGGL(fused_train_forward_kernel<DimsNarrow>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, x, k,
                       temb, temb_stride, tab, n_slots, drop_p, seed, offset0, head_c, head_d); break;
  }
  return check_launch("mmb_fused_epic_train_forward");
}

// variant: 0 = S2 widths (Et128 Ed128 H128 G16), 1 = Et16 Ed16 H32 G8
extern "C" int mmb_fused_uni_forward(int variant, int B, int N, const float* packed, const int32_t* desc,
                                     int num_blocks, int use_skip, int off_table, const float* maskf,
                                     const int32_t* counts, const int64_t* k, const float* temb, int64_t temb_stride,
                                     float* head_d, void* stream) {
#if MMB_FE_WAVES == 4
  if (N > 128) return mmb_fused_uni_forward_w5(variant, B, N, packed, desc, num_blocks, use_skip, off_table, maskf, counts, k, temb,
                                               temb_stride, head_d, stream);
#endif
  int rc = check_fused("mmb_fused_uni_forward", B, N, packed, desc, maskf, counts, n_desc_uni(num_blocks));
  if (rc) return rc;
  MMB_REQUIRE(variant >= 0 && variant < N_UNI_VARIANTS, "mmb_fused_uni_forward: unknown variant %d", variant);
  MMB_REQUIRE(k && temb && head_d, "mmb_fused_uni_forward: NULL state");
  FusedArgs A{nullptr, packed, reinterpret_cast<const LDesc*>(desc), num_blocks, 0, use_skip, 0, 0, off_table, B, N,
              maskf, counts, 0, n_desc_uni(num_blocks), fwd_tiles()};
  const int grid = B < 4096 ? B : 4096;
  switch (variant) {
    case 0: hipLaunchKernelGGL(fused_uni_forward_kernel<UniS2>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, k, temb,
                       temb_stride, head_d); break;
    case 1: hipLaunchKernelGGL(fused_uni_forward_kernel<UniMid>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, k, temb,
                       temb_stride, head_d); break;
    case 2: hipLaunchKernelGGL(fused_uni_forward_kernel<UniH64>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, k, temb,
                       temb_stride, head_d); break;
    default: hipLaunchKernelGGL(fused_uni_forward_kernel<UniH32>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, k, temb,
                       temb_stride, head_d); break;
  }
  return check_launch("mmb_fused_uni_forward");
}

extern "C" int mmb_fused_uni_train_forward(int variant, int B, int N, const float* packed, const int32_t* desc,
                                           int num_blocks, int use_skip, int off_table, const float* maskf,
                                           const int32_t* counts, const int64_t* k, const float* temb,
                                           int64_t temb_stride, const void* slots, int n_slots, float drop_p,
                                           uint64_t seed, uint64_t offset0, float* head_d, const int32_t* jet_order,
                                           void* stream) {
#if MMB_FE_WAVES == 4
  if (N > 128) return mmb_fused_uni_train_forward_w5(variant, B, N, packed, desc, num_blocks, use_skip, off_table, maskf, counts,
                                                     k, temb, temb_stride, slots, n_slots, drop_p, seed, offset0, head_d, jet_order,
                                                     stream);
#endif
  int rc = check_fused("mmb_fused_uni_train_forward", B, N, packed, desc, maskf, counts, n_desc_uni(num_blocks));
  if (rc) return rc;
  MMB_REQUIRE(variant >= 0 && variant < N_UNI_VARIANTS, "mmb_fused_uni_train_forward: unknown variant %d", variant);
  MMB_REQUIRE(k && temb && head_d && slots, "mmb_fused_uni_train_forward: NULL argument");
  MMB_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "mmb_fused_uni_train_forward: dropout p=%f outside [0,1)", drop_p);
  const int expect = 9 + 8 * num_blocks;
  MMB_REQUIRE(n_slots == expect, "mmb_fused_uni_train_forward: %d save slots given, the architecture has %d", n_slots,
              expect);
  FusedArgs A{nullptr, packed, reinterpret_cast<const LDesc*>(desc), num_blocks, 0, use_skip, 0, 0, off_table, B, N,
              maskf, counts, 0, n_desc_uni(num_blocks), train_tiles(), jet_order};
  const SaveSlot* tab = reinterpret_cast<const SaveSlot*>(slots);
  const int grid = B < 4096 ? B : 4096;
  switch (variant) {
    case 0: hipLaunchKernelGGL(fused_uni_train_forward_kernel<UniS2>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, k,
                       temb, temb_stride, tab, n_slots, drop_p, seed, offset0, head_d); break;
    case 1: hipLaunchKernelGGL(fused_uni_train_forward_kernel<UniMid>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, k,
                       temb, temb_stride, tab, n_slots, drop_p, seed, offset0, head_d); break;
    case 2: hipLaunchKernelGGL(fused_uni_train_forward_kernel<UniH64>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, k,
                       temb, temb_stride, tab, n_slots, drop_p, seed, offset0, head_d); break;
    default: hipLaunchKernelGGL(fused_uni_train_forward_kernel<UniH32>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, k,
                       temb, temb_stride, tab, n_slots, drop_p, seed, offset0, head_d); break;
  }
  return check_launch("mmb_fused_uni_train_forward");
}

extern "C" int mmb_fused_uni_generate(int variant, int B, int N, const float* packed, const int32_t* desc,
                                      int num_blocks, int use_skip, int off_table, const float* maskf,
                                      const int32_t* counts, int64_t* k, int steps, const float* times,
                                      const float* temb_steps, const float* ctab_steps, int n_ctab, float dt,
                                      int thermo_kind, float gamma, const float* delta_n, uint64_t seed,
                                      uint64_t offset, const int32_t* jet_order, const int32_t* long_jets, int n_long,
                                      void* stream) {
  int tiles = gen_tiles();
  bool shadows = false;                                  // 128 < N <= 160: a split call, as mmb_fused_epic_generate
#if MMB_FE_WAVES == 4
  if (N > 128) {
    shadows = (tiles & 1) != 0 && gen_split() != 0 && N <= 160;
    if (!shadows)
      return mmb_fused_uni_generate_w5(variant, B, N, packed, desc, num_blocks, use_skip, off_table, maskf, counts, k, steps,
                                       times, temb_steps, ctab_steps, n_ctab, dt, thermo_kind, gamma, delta_n, seed, offset,
                                       jet_order, nullptr, -1, stream);
    tiles |= 4;
  }
  const int n_jets = B;
#else
  int n_jets = B;
  if (variant & 0x100) {                                 // the remainder launch of a split call
    tiles |= 4;
    variant &= 0xFF;
    if (n_long >= 0) { jet_order = long_jets; n_jets = n_long; }
  }
#endif
  int rc = check_fused("mmb_fused_uni_generate", B, shadows ? 128 : N, packed, desc, maskf, counts, n_desc_uni(num_blocks));
  if (rc) return rc;
  MMB_REQUIRE(variant >= 0 && variant < N_UNI_VARIANTS, "mmb_fused_uni_generate: unknown variant %d", variant);
  MMB_REQUIRE(k && times && temb_steps && steps > 0, "mmb_fused_uni_generate: NULL state / steps");
  MMB_REQUIRE(ctab_steps && n_ctab > 0, "mmb_fused_uni_generate: the per-grid-point time-vector table is required");
  MMB_REQUIRE(thermo_kind >= 0 && thermo_kind <= 4, "mmb_fused_uni_generate: bad thermostat");
  FusedArgs A{nullptr, packed, reinterpret_cast<const LDesc*>(desc), num_blocks, 0, use_skip, 0, 0, off_table, B, N,
              maskf, counts, 0, n_desc_uni(num_blocks)};
  GenArgs Gn{steps, times, temb_steps, ctab_steps, n_ctab, dt, thermo_kind, (float)(-(double)8 * (double)gamma), delta_n,
             seed, offset, 0, 0.0f, nullptr, stagger_cycles(0), tiles, jet_order, n_jets};
  const int grid = n_jets < 4096 ? n_jets : 4096;
#define MMB_UNI_LAUNCH(CHv)                                                                                                          \
  switch (variant) {                                                                                                                 \
    case 0: hipLaunchKernelGGL((fused_uni_generate_kernel<UniS2, FE_WAVES, CHv>), dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, k); break;   \
    case 1: hipLaunchKernelGGL((fused_uni_generate_kernel<UniMid, FE_WAVES, CHv>), dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, k); break;  \
    case 2: hipLaunchKernelGGL((fused_uni_generate_kernel<UniH64, FE_WAVES, CHv>), dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, k); break;  \
    default: hipLaunchKernelGGL((fused_uni_generate_kernel<UniH32, FE_WAVES, CHv>), dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, k); break; \
  }
#if MMB_FE_WAVES == 4
  if (shadows && n_long == 0) {
    MMB_UNI_LAUNCH(2)
    return check_launch("mmb_fused_uni_generate");
  }
  if (shadows) {
    MMB_REQUIRE(n_long < 0 || (long_jets != nullptr && n_long <= B), "mmb_fused_uni_generate: long_jets / n_long = %d", n_long);
    SideStream* sd = side_stream();
    MMB_REQUIRE(sd != nullptr, "mmb_fused_uni_generate: side stream");
    MMB_REQUIRE(hipEventRecord(sd->e0, as_stream(stream)) == hipSuccess && hipStreamWaitEvent(sd->s, sd->e0, 0) == hipSuccess,
                "mmb_fused_uni_generate: fork of the side stream");
    rc = mmb_fused_uni_generate_w5(variant | 0x100, B, N, packed, desc, num_blocks, use_skip, off_table, maskf, counts, k, steps,
                                   times, temb_steps, ctab_steps, n_ctab, dt, thermo_kind, gamma, delta_n, seed, offset, jet_order,
                                   long_jets, n_long, sd->s);
    if (rc == MMB_OK) {
      MMB_UNI_LAUNCH(2)
      rc = check_launch("mmb_fused_uni_generate");
    }
    MMB_REQUIRE(hipEventRecord(sd->e1, sd->s) == hipSuccess && hipStreamWaitEvent(as_stream(stream), sd->e1, 0) == hipSuccess,
                "mmb_fused_uni_generate: join of the side stream");
    return rc;
  }
#endif
  MMB_UNI_LAUNCH(1)
#undef MMB_UNI_LAUNCH
  return check_launch("mmb_fused_uni_generate");
}

#if MMB_FE_WAVES == 4
// ---- launch order of the jets of a generation call ------------------------------------------------
// A jet costs what its working tiles cost (TilePlan), and a call has few jets per workgroup slot (1024 jets on 512
// slots).  Two things decide the time of the call: which jets share a CU -- the working tiles of two jets run side by
// side only while together they fit the four SIMDs (steer_wave) -- and how long the last slots run.  The jets are sorted
// by particle count (a counting sort in one workgroup; jets of one count in any order: a jet's result does not depend on
// the workgroup that computes it) and dealt out in rounds of `chunk` = one workgroup per CU: the heaviest chunk, then
// the lightest chunk (workgroups i and i + chunk share CU i in the first two rounds: heaviest with lightest), the
// second heaviest, the second lightest, ...; what is left in the middle follows, heaviest first.  chunk = 0: plain
// longest first.
__device__ __forceinline__ int schedule_pos(int r, int B, int chunk) {
  if (chunk <= 0) return r;
  const int cp = (B / (2 * chunk)) * chunk;              // jets on either end that form whole pairs of rounds
  if (r < cp) return 2 * (r / chunk) * chunk + r % chunk;
  const int rl = B - 1 - r;
  if (rl < cp) return (2 * (rl / chunk) + 1) * chunk + rl % chunk;
  return 2 * cp + (r - cp);
}
__global__ void __launch_bounds__(1024) jet_schedule_kernel(const int32_t* __restrict__ counts, int B, int chunk,
                                                           int32_t* __restrict__ order) {
  __shared__ int bin[256];
  const int tid = threadIdx.x;
  if (tid < 256) bin[tid] = 0;
  __syncthreads();
  for (int i = tid; i < B; i += 1024) atomicAdd(&bin[255 - min(max(counts[i], 0), 255)], 1);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int c = 0; c < 256; ++c) { const int n = bin[c]; bin[c] = run; run += n; }
  }
  __syncthreads();
  for (int i = tid; i < B; i += 1024)
    order[schedule_pos(atomicAdd(&bin[255 - min(max(counts[i], 0), 255)], 1), B, chunk)] = i;
}

extern "C" int mmb_jet_schedule(int B, const int32_t* counts, int32_t* order, void* stream) {
  MMB_REQUIRE(B > 0 && counts && order, "mmb_jet_schedule: needs B > 0 and two buffers of B int32 (got B=%d)", B);
  int chunk = 0;
  if (const char* e = getenv("MMB_GEN_PAIR")) {          // developer switch: the round size (0: plain longest first)
    chunk = atoi(e);
  } else {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&chunk, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      chunk = 0;
  }
  hipLaunchKernelGGL(jet_schedule_kernel, dim3(1), dim3(1024), 0, as_stream(stream), counts, B, chunk, order);
  return check_launch("mmb_jet_schedule");
}

// ==================================================================== MultiModalMLP (toy, S1) ====
// The whole generation loop of the toy hybrid experiment (notebooks/data_toys.ipynb cell 3 inside
// multimodal_bridge_matching.py:227-259) in ONE launch: the layered path needs 16 launches per grid point and is
// launch-bound (25 ms per 100-step generation of 1024 points).  One wave owns 32 points (lane & 31 = point, both lane
// halves carry its state); all weights (57 KB fp32) sit in LDS; activations stay in registers as TRANSPOSED tiles in
// the C/D layout of v_mfma_f32_32x32x2_f32 (lane = point, registers = features), so a layer's output IS the next
// layer's B operand: MFMA step r contracts the two features that register r holds in the two lane halves and the A
// operand is those two columns of the weight matrix (exact fp32: bit-for-bit an fmaf chain).
struct ToyW {
  const float* table;                      // (V, H/4) embedding
  const float *w0, *b0, *w1, *b1, *w2, *b2, *w3, *b3;   // body: (H, 1+DC+H/4), 3 x (H, H)
  const float *wc, *bc, *wd, *bd;          // heads: (DC, H), (V, H)
};

__device__ __forceinline__ float selu_f(float x) {
  return x > 0.0f ? 1.0507009873554804934193349852946f * x
                  : (1.0507009873554804934193349852946f * 1.6732632423543772848170429916717f) * (expf(x) - 1.0f);
}
__device__ __forceinline__ int toy_crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int H, int DC>
__global__ void __launch_bounds__(256)
toy_generate_kernel(ToyW W, GenArgs Gn, int B, int N, const float* __restrict__ maskf, float* __restrict__ x,
                    int64_t* __restrict__ k) {
  constexpr int V = 8, E = H / 4, K0 = 1 + DC + E, HT = H / 32;
  constexpr int P0 = 33, PH = H + 1;                       // LDS pitches (odd: conflict-free column reads)
  static_assert(K0 <= 32 && DC + V <= 32 && H % 32 == 0, "toy kernel shape");
  extern __shared__ float sm[];
  float* w0 = sm;                          // [H][P0]   columns >= K0 are zeros
  float* wl = w0 + H * P0;                 // [3][H][PH]
  float* wh = wl + 3 * H * PH;             // [32][PH]  rows 0..DC-1 drift head, DC..DC+V-1 logits, rest zeros
  float* bs = wh + 32 * PH;                // [4][H] + [32]
  float* tb = bs + 4 * H + 32;             // [V][E]
  const int tid = threadIdx.x;
  for (int i = tid; i < H * P0; i += blockDim.x) {
    const int o = i / P0, c = i - o * P0;
    w0[i] = c < K0 ? W.w0[o * K0 + c] : 0.0f;
  }
  for (int i = tid; i < 3 * H * PH; i += blockDim.x) {
    const int l = i / (H * PH), r = i - l * (H * PH), o = r / PH, c = r - o * PH;
    const float* src = l == 0 ? W.w1 : (l == 1 ? W.w2 : W.w3);
    wl[i] = c < H ? src[o * H + c] : 0.0f;
  }
  for (int i = tid; i < 32 * PH; i += blockDim.x) {
    const int o = i / PH, c = i - o * PH;
    float v = 0.0f;
    if (c < H) v = o < DC ? W.wc[o * H + c] : (o < DC + V ? W.wd[(o - DC) * H + c] : 0.0f);
    wh[i] = v;
  }
  for (int i = tid; i < 4 * H + 32; i += blockDim.x) {
    float v = 0.0f;
    if (i < H) v = W.b0[i];
    else if (i < 2 * H) v = W.b1[i - H];
    else if (i < 3 * H) v = W.b2[i - 2 * H];
    else if (i < 4 * H) v = W.b3[i - 3 * H];
    else { const int o = i - 4 * H; v = o < DC ? W.bc[o] : (o < DC + V ? W.bd[o - DC] : 0.0f); }
    bs[i] = v;
  }
  for (int i = tid; i < V * E; i += blockDim.x) tb[i] = W.table[i];
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  const int64_t P = (int64_t)B * N;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile * 32 < P; tile += (int64_t)gridDim.x * 4) {
    const int64_t gp = tile * 32 + col;
    const bool inb = gp < P;
    const int64_t gl = inb ? gp : P - 1;
    const bool valid = inb && maskf[gl] != 0.0f;
    const int b = (int)(gl / N), p = (int)(gl - (int64_t)b * N);
    float xp[DC];
#pragma unroll
    for (int d = 0; d < DC; ++d) xp[d] = x[gl * DC + d];
    int kp = (int)k[gl];

    // heads of one forward pass at time tt: out[r] = row toy_crow(r, half) of [drift | logits | 0...]
    auto forward = [&](float tt, f32x16& out) __attribute__((always_inline)) {
      f32x16 in;
      ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v;
        const int f = toy_crow(r, half);
        float v = 0.0f;
        if (f == 0) v = tt;
        else if (f > DC && f < K0) v = tb[kp * E + (f - 1 - DC)];
        ffor<DC>([&](auto dc) { constexpr int d = decltype(dc)::v; if (f == 1 + d) v = xp[d]; });
        in[r] = v; });
      f32x16 h[HT], g[HT];
      ffor<HT>([&](auto mc) { constexpr int mt = decltype(mc)::v;
        ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; h[mt][r] = bs[32 * mt + toy_crow(r, half)]; });
        ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v;
          h[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[(32 * mt + col) * P0 + toy_crow(r, half)], in[r], h[mt], 0, 0, 0); });
        ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; h[mt][r] = selu_f(h[mt][r]); }); });
      ffor<3>([&](auto lc) { constexpr int l = decltype(lc)::v;
        const float* wm = wl + l * H * PH;
        ffor<HT>([&](auto mc) { constexpr int mt = decltype(mc)::v;
          ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; g[mt][r] = bs[(l + 1) * H + 32 * mt + toy_crow(r, half)]; });
          ffor<HT>([&](auto kc) { constexpr int kt = decltype(kc)::v;
            ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v;
              g[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wm[(32 * mt + col) * PH + 32 * kt + toy_crow(r, half)], h[kt][r],
                                                          g[mt], 0, 0, 0); }); }); });
        ffor<HT>([&](auto mc) { constexpr int mt = decltype(mc)::v;
          ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; h[mt][r] = l < 2 ? selu_f(g[mt][r]) : g[mt][r]; }); }); });
      ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; out[r] = bs[4 * H + toy_crow(r, half)]; });
      ffor<HT>([&](auto kc) { constexpr int kt = decltype(kc)::v;
        ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v;
          out = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[col * PH + 32 * kt + toy_crow(r, half)], h[kt][r], out, 0, 0, 0); }); });
    };
    // row `row` (compile time) of the head tile, delivered to BOTH lane halves
    auto head_row = [&](const f32x16& out, auto rowc) __attribute__((always_inline)) {
      constexpr int row = decltype(rowc)::v;
      constexpr int hsrc = (row >> 2) & 1, reg = (row & 3) + 4 * (row >> 3);
      const float mine = out[reg], other = __shfl_xor(mine, 32, 64);
      return half == hsrc ? mine : other;
    };

    for (int s = 0; s < Gn.steps; ++s) {
      const float tt = Gn.times[s];
      f32x16 out;
      forward(tt, out);
      {
        float w[DC];
#pragma unroll
        for (int d = 0; d < DC; ++d) w[d] = 0.0f;
        float sig = 0.0f;
        if (Gn.em) {
          sig = Gn.em_sigma * sqrtf(tt * (1.0f - tt));
          if (Gn.delta_w != nullptr) {
            if (valid) {
              const float* dw = Gn.delta_w + (((int64_t)s * B + b) * N + p) * DC;
#pragma unroll
              for (int d = 0; d < DC; ++d) w[d] = dw[d];
            }
          } else {
            uint32_t q[4];
            philox4(Gn.seed, Gn.offset + (uint64_t)s, (uint64_t)gl, 0xE001u, q);
            const float r0 = sqrtf(-2.0f * __logf(u01_open0(q[0]))), r1 = sqrtf(-2.0f * __logf(u01_open0(q[2])));
            const float z[4] = {r0 * __builtin_amdgcn_cosf(u01(q[1])), r0 * __builtin_amdgcn_sinf(u01(q[1])),
                                r1 * __builtin_amdgcn_cosf(u01(q[3])), r1 * __builtin_amdgcn_sinf(u01(q[3]))};
#pragma unroll
            for (int d = 0; d < DC; ++d) w[d] = z[d & 3];
          }
        }
        ffor<DC>([&](auto dc) { constexpr int d = decltype(dc)::v;
          const float drift = head_row(out, IF<d>{});
          xp[d] += Gn.em ? (Gn.dt * drift + sig * w[d]) : Gn.dt * drift; });
      }
      // tau-leap on the logits head, which sees the UPDATED x at the same t (solvers.py:59-81)
      forward(tt, out);
      float r[V];
      ffor<V>([&](auto vc) { constexpr int v = decltype(vc)::v; r[v] = head_row(out, IF<DC + v>{}); });
      kp = jump_step(r, kp, tt, Gn, s == Gn.steps - 1, valid, B, N, b, p, s, gl);
    }
    if (valid && half == 0) {
#pragma unroll
      for (int d = 0; d < DC; ++d) x[gp * DC + d] = xp[d];
      k[gp] = kp;
    }
  }
}

extern "C" int mmb_toy_generate(int B, int N, int H, int Dc, int V, const float* table, const float* const* weights,
                                const float* maskf, float* x, int64_t* k, int steps, const float* times, float dt,
                                int thermo_kind, float gamma, const float* delta_n, uint64_t seed, uint64_t offset,
                                int em, float em_sigma, const float* delta_w, void* stream) {
  MMB_REQUIRE(B > 0 && N > 0 && steps > 0, "mmb_toy_generate: empty shape");
  MMB_REQUIRE(H == 64 && Dc == 2 && V == 8, "mmb_toy_generate: compiled for dim_hidden 64, 2 continuous features, "
              "8 tokens (got %d, %d, %d): use the layered entry points", H, Dc, V);
  MMB_REQUIRE(table && weights && maskf && x && k && times, "mmb_toy_generate: NULL argument");
  for (int i = 0; i < 12; ++i) MMB_REQUIRE(weights[i], "mmb_toy_generate: NULL weight %d", i);
  MMB_REQUIRE(thermo_kind >= 0 && thermo_kind <= 4, "mmb_toy_generate: bad thermostat");
  ToyW W{table, weights[0], weights[1], weights[2], weights[3], weights[4], weights[5], weights[6], weights[7],
         weights[8], weights[9], weights[10], weights[11]};
  GenArgs Gn{steps, times, nullptr, nullptr, 0, dt, thermo_kind, (float)(-(double)V * (double)gamma), delta_n, seed, offset,
             em, em_sigma, delta_w, 0};
  constexpr int HH = 64;
  const size_t lds = (size_t)(HH * 33 + 3 * HH * (HH + 1) + 32 * (HH + 1) + 4 * HH + 32 + 8 * (HH / 4)) * sizeof(float);
  static PerDeviceOnce lds_limit;                        // the runtime keeps the attribute per device
  MMB_REQUIRE(lds_limit.run([&] {
                return hipFuncSetAttribute((const void*)toy_generate_kernel<64, 2>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess; }),
              "mmb_toy_generate: cannot raise the dynamic LDS limit");
  const int64_t tiles = ((int64_t)B * N + 31) / 32;
  const int64_t blocks = (tiles + 3) / 4;
  hipLaunchKernelGGL((toy_generate_kernel<64, 2>), dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), lds,
                     as_stream(stream), W, Gn, B, N, maskf, x, k);
  return check_launch("mmb_toy_generate");
}

#ifdef MMB_FE_PROF
extern "C" int mmb_debug_fused_prof(unsigned long long* out16, int reset) {
  if (reset) {
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(fe_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
  }
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(fe_prof), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

#endif  // MMB_FE_WAVES == 4

extern "C" int mmb_fused_epic_generate(int variant, int B, int N, const float* packed, const int32_t* desc,
                                       int nb_branch, int nb_branch1, int nb_fused_exec, int use_skip, int off_cw, int off_cb,
                                       int off_table, const float* maskf, const int32_t* counts, float* x, int64_t* k,
                                       int steps, const float* times, const float* temb_steps,
                                       const float* ctab_steps, int n_ctab, float dt,
                                       int thermo_kind, float gamma, const float* delta_n, uint64_t seed,
                                       uint64_t offset, int em, float em_sigma, const float* delta_w,
                                       const int32_t* jet_order, const int32_t* long_jets, int n_long, void* stream) {
  int tiles = gen_tiles();
  // 128 < N <= 160 is a split call when tiles may be idle: the jets of fewer than 128 particles (AOJ: all but 3 in
  // 10 000) run on four waves with shadow slots (CH = 2), two to a CU; the five-wave build (bit 8 of its `variant`
  // says so) takes the rest.  Every workgroup of either launch leaves at once when its jet belongs to the other one.
  bool shadows = false;
#if MMB_FE_WAVES == 4
  if (N > 128) {
    shadows = (tiles & 1) != 0 && gen_split() != 0 && N <= 160;
    if (!shadows)
      return mmb_fused_epic_generate_w5(variant, B, N, packed, desc, nb_branch, nb_branch1, nb_fused_exec, use_skip, off_cw, off_cb,
                                        off_table, maskf, counts, x, k, steps, times, temb_steps, ctab_steps, n_ctab, dt, thermo_kind,
                                        gamma, delta_n, seed, offset, em, em_sigma, delta_w, jet_order, nullptr, -1, stream);
    tiles |= 4;
  }
  const int n_jets = B;
#else
  int n_jets = B;
  if (variant & 0x100) {                                 // the remainder launch of a split call
    tiles |= 4;
    variant &= 0xFF;
    if (n_long >= 0) { jet_order = long_jets; n_jets = n_long; }       // the caller knows the long jets: only those
  }
#endif
  int rc = check_fused("mmb_fused_epic_generate", B, shadows ? 128 : N, packed, desc, maskf, counts,
                       n_desc_multi(nb_branch, nb_branch1, nb_fused_exec));
  if (rc) return rc;
  MMB_REQUIRE(variant >= 0 && variant < N_MULTI_VARIANTS, "mmb_fused_epic_generate: unknown variant %d", variant);
  MMB_REQUIRE(x && k && times && temb_steps && steps > 0, "mmb_fused_epic_generate: NULL state / steps");
  MMB_REQUIRE(ctab_steps && n_ctab > 0, "mmb_fused_epic_generate: the per-grid-point time-vector table is required");
  MMB_REQUIRE(thermo_kind >= 0 && thermo_kind <= 4, "mmb_fused_epic_generate: bad thermostat");
  FusedArgs A{nullptr, packed, reinterpret_cast<const LDesc*>(desc), nb_branch, nb_fused_exec, use_skip, off_cw, off_cb,
              off_table, B, N, maskf, counts, nb_branch1, n_desc_multi(nb_branch, nb_branch1, nb_fused_exec)};
  GenArgs Gn{steps, times, temb_steps, ctab_steps, n_ctab, dt, thermo_kind, (float)(-(double)8 * (double)gamma), delta_n, seed, offset,
             em, em_sigma, delta_w, stagger_cycles(0), tiles, jet_order, n_jets};
  const int grid = n_jets < 4096 ? n_jets : 4096;
#define MMB_GEN_LAUNCH(CHv)                                                                                                          \
  switch (variant) {                                                                                                                 \
    case 0: hipLaunchKernelGGL((fused_generate_kernel<DimsS3, FE_WAVES, CHv>), dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, x, k); break;    \
    case 1: hipLaunchKernelGGL((fused_generate_kernel<DimsMid, FE_WAVES, CHv>), dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, x, k); break;   \
    case 2: hipLaunchKernelGGL((fused_generate_kernel<DimsWide, FE_WAVES, CHv>), dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, x, k); break;  \
    default: hipLaunchKernelGGL((fused_generate_kernel<DimsNarrow, FE_WAVES, CHv>), dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, x, k); break; \
  }
#if MMB_FE_WAVES == 4
  if (shadows && n_long == 0) {                            // the caller knows there is no long jet: one launch
    MMB_GEN_LAUNCH(2)
    return check_launch("mmb_fused_epic_generate");
  }
  if (shadows) {
    MMB_REQUIRE(n_long < 0 || (long_jets != nullptr && n_long <= B), "mmb_fused_epic_generate: long_jets / n_long = %d", n_long);
    // The five-wave launch holds a handful of jets, each of which still takes a whole generation loop: behind the
    // four-wave launch on one stream that is a third on top of the call.  It goes first, on a side stream per device
    // that starts behind whatever precedes this call on the caller's stream (event e0) and is joined back (event e1)
    // after the four-wave launch: the two write disjoint jets.  With the list of long jets (long_jets, n_long) it is
    // n_long workgroups; without (n_long < 0) every jet gets a workgroup that leaves at once when its jet is short --
    // but each of those needs an EMPTY CU to start (98 KB of LDS), which costs the four-wave launch ~15 ms.
    SideStream* sdp = side_stream();
    MMB_REQUIRE(sdp != nullptr, "mmb_fused_epic_generate: side stream");
    SideStream& sd = *sdp;
    MMB_REQUIRE(hipEventRecord(sd.e0, as_stream(stream)) == hipSuccess && hipStreamWaitEvent(sd.s, sd.e0, 0) == hipSuccess,
                "mmb_fused_epic_generate: fork of the side stream");
    rc = mmb_fused_epic_generate_w5(variant | 0x100, B, N, packed, desc, nb_branch, nb_branch1, nb_fused_exec, use_skip, off_cw,
                                    off_cb, off_table, maskf, counts, x, k, steps, times, temb_steps, ctab_steps, n_ctab, dt,
                                    thermo_kind, gamma, delta_n, seed, offset, em, em_sigma, delta_w, jet_order, long_jets, n_long,
                                    sd.s);
    if (rc == MMB_OK) {
      MMB_GEN_LAUNCH(2)
      rc = check_launch("mmb_fused_epic_generate");
    }
    MMB_REQUIRE(hipEventRecord(sd.e1, sd.s) == hipSuccess && hipStreamWaitEvent(as_stream(stream), sd.e1, 0) == hipSuccess,
                "mmb_fused_epic_generate: join of the side stream");
    return rc;
  }
#endif
  MMB_GEN_LAUNCH(1)
#undef MMB_GEN_LAUNCH
  return check_launch("mmb_fused_epic_generate");
}
