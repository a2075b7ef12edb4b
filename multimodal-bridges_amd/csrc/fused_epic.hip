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

#include "fused_common.h"

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
enum { PF_EMBED = 0, PF_POOL, PF_GLOB1, PF_GLOB2, PF_DVEC, PF_LOC1, PF_LOC2, PF_EPI, PF_OUT, PF_PROJ_MMA, PF_HEAD, PF_STEP, PF_N };

// ------------------------------------------------------------------ one EPiC encoder ----
// x: HT tiles (in/out), skip: HT tiles.  Global vector in L.gcur.  `li` walks the descriptors.
template <int HT, int ET, int G>
struct Epic {
  static constexpr int H = 32 * HT;

  // pooled = [mean, 0.01*sum] of the wave partials (without the tail) -> the tape's (B, 2H) tensor
  template <class TR>
  static __device__ __forceinline__ void save_pooled(TR& tr, Lds& L, int tid) {
    if constexpr (TR::on) {
      float* dst = tr.row(2 * H);
      const PooledIn<H> pin{L.partial, nullptr, L.cnt};
      if (dst != nullptr)
        for (int k = tid; k < 2 * H; k += FE_THREADS) dst[k] = pin(k);
    }
  }

  // EPiCLayer (epic.py:187-208) + the encoder-level skip add (:89-90).  skipf(IF<t>) yields skip tile t.
  // Training slots: pooled, gh, g_new, g_out, sign bits of lh's pre-activation, lh, sign bits of y1 (the
  // leaky-relu input behind l_out), l_out: the reverse pass takes the two derivatives from 2 x 8 bytes per lane instead of
  // re-reading lh, l_out and the skip tensor.
  template <typename SkipFn, class TR>
  static __device__ __forceinline__ void layer(f32x16 (&x)[HT], SkipFn&& skipf, bool use_skip,
                                               const float* __restrict__ P, const LDesc* __restrict__ D, int& li, Lds& L,
                                               float m, int tid, const float* ctab, int& ci, TR& tr) {
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const WRsrc R = weight_rsrc(P);
    const LDesc g1 = D[li], g2 = D[li + 1], l1 = D[li + 2], l2 = D[li + 3];
    li += 4;
    FE_PROF(L, PF_EPI);
    pool_partials<HT>(x, m, L.partial, wave, lane);
    AFrag<HT> pre1, pre2;
    if constexpr (HT <= 2) mma_prefetch(pre1, R, l1.wx, lane);   // arrives under the per-jet chain
    __syncthreads();
    FE_PROF(L, PF_POOL);
    save_pooled(tr, L, tid);
    WaveMV<2 * H + G, H>::template run<MMB_ACT_LRELU>(P + g1.wt, P + g1.b, PooledIn<H>{L.partial, L.gcur, L.cnt}, L.gh,
                                                      nullptr, tid);
    if constexpr (TR::on) tr.vec(L.gh, H, tid);
    FE_PROF(L, PF_GLOB1);
    matvec<MMB_ACT_NONE, H, G>(P + g2.wt, P + g2.b, L.gh, L.gnew, L.gcur, tid);      // g' = g + fc_glob2(.)
    if constexpr (TR::on) tr.vec(L.gnew, G, tid);
    FE_PROF(L, PF_GLOB2);
    const float* cv = time_vec<ET, H>(ctab, ci, P + l1.wt, L, tid);
    matvec<MMB_ACT_NONE, G, H>(P + l1.wg, P + l1.b, L.gnew, L.dvec, nullptr, tid);
    FE_PROF(L, PF_DVEC);
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
    f32x16 lh[HT];
    if constexpr (HT > 2) mma_prefetch(pre1, R, l1.wx, lane);    // (wide encoder: no registers to spare earlier)
    acc_init<HT>(lh, m, cv, L.dvec, half);
    if constexpr (HT <= 2) mma_prefetch(pre2, R, l2.wx, lane);   // arrives under the first GEMM
    mma_x<HT, HT>(lh, [&](auto fc) -> const f32x16& { return x[decltype(fc)::v]; }, R, l1.wx, lane, pre1);
    if constexpr (TR::on) tr.bits(sign_bits<HT>(lh), tid);
    act_tiles<HT, MMB_ACT_LRELU>(lh);
    if constexpr (HT > 2) mma_prefetch(pre2, R, l2.wx, lane);    // before the stores: vmcnt is one in-order counter
    if constexpr (TR::on) tr.template tiles<HT>(lh, half);
    FE_PROF(L, PF_LOC1);
    // x <- (x + b2) + W2 lh : the residual rides in the accumulator, x's registers ARE the accumulator
    ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v;
      ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
        const float4 bv = *reinterpret_cast<const float4*>(P + l2.b + 32 * t + 8 * q + 4 * half);
        x[t][4 * q + 0] += bv.x; x[t][4 * q + 1] += bv.y; x[t][4 * q + 2] += bv.z; x[t][4 * q + 3] += bv.w; }); });
    mma_x<HT, HT>(x, [&](auto fc) -> const f32x16& { return lh[decltype(fc)::v]; }, R, l2.wx, lane, pre2);
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
    __syncthreads();                                   // gcur settled before the next pooled build
    FE_PROF(L, PF_EPI);
  }

  // output layer: wn Linear on cat[t, x, g] (epic.py:93-96); OTO output tiles, OW = true output width
  // Training slot: out (only when OW is a whole number of tiles).
  template <int OTO, int OW, class TR>
  static __device__ __forceinline__ void output(f32x16 (&out)[OTO], const f32x16 (&x)[HT],
                                                const float* __restrict__ P, const LDesc* __restrict__ D, int& li, Lds& L,
                                                float m, int tid, const float* ctab, int& ci, TR& tr) {
    const int lane = tid & 63, half = lane >> 5;
    const WRsrc R = weight_rsrc(P);
    const LDesc o = D[li];
    li += 1;
    AFrag<OTO> pre;
    mma_prefetch(pre, R, o.wx, lane);
    const float* cv = time_vec<ET, OW>(ctab, ci, P + o.wt, L, tid);
    matvec<MMB_ACT_NONE, G, OW>(P + o.wg, P + o.b, L.gcur, L.dvec, nullptr, tid);
    acc_init<OTO>(out, m, cv, L.dvec, half);
    mma_x<HT, OTO>(out, [&](auto fc) -> const f32x16& { return x[decltype(fc)::v]; }, R, o.wx, lane, pre);
    if constexpr (TR::on && OW == 32 * OTO) tr.template tiles<OTO>(out, half);
    FE_PROF(L, PF_OUT);
  }

  // EPiCProjection (epic.py:145-159) from KIN padded input tiles delivered by xin(IF<t>)
  // Training slots: p0 pre, p0 out, p2 pre, p2 out, pooled, q0 pre, q0 out, q2 pre, q2 out.
  template <int KIN, typename InFn, class TR>
  static __device__ __forceinline__ void project(f32x16 (&x)[HT], InFn&& xin,
                                                 const float* __restrict__ P, const LDesc* __restrict__ D, int& li, Lds& L,
                                                 float m, int tid, const float* ctab, int& ci, TR& tr) {
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const WRsrc R = weight_rsrc(P);
    const LDesc p0 = D[li], p2 = D[li + 1], q0 = D[li + 2], q2 = D[li + 3];
    li += 4;
    AFrag<HT> pre0, pre2;
    mma_prefetch(pre0, R, p0.wx, lane);
    mma_prefetch(pre2, R, p2.wx, lane);
    const float* cv = time_vec<ET, H>(ctab, ci, P + p0.wt, L, tid);
    f32x16 h[HT];
    acc_init<HT>(h, m, cv, P + p0.b, half);
    mma_x<KIN, HT>(h, xin, R, p0.wx, lane, pre0);
    if constexpr (TR::on) tr.template tiles<HT>(h, half);
    act_tiles<HT, MMB_ACT_GELU>(h);
    if constexpr (TR::on) tr.template tiles<HT>(h, half);
    acc_init<HT>(x, 0.0f, nullptr, P + p2.b, half);
    mma_x<HT, HT>(x, [&](auto fc) -> const f32x16& { return h[decltype(fc)::v]; }, R, p2.wx, lane, pre2);
    if constexpr (TR::on) tr.template tiles<HT>(x, half);
    act_tiles<HT, MMB_ACT_GELU>(x);
    if constexpr (TR::on) tr.template tiles<HT>(x, half);
    FE_PROF(L, PF_PROJ_MMA);
    pool_partials<HT>(x, m, L.partial, wave, lane);
    __syncthreads();                                   // context = time embedding (dim_context = Et)
    save_pooled(tr, L, tid);
    float* pre_q0 = nullptr;
    if constexpr (TR::on) pre_q0 = tr.row(H);
    WaveMV<2 * H + ET, H>::template run<MMB_ACT_GELU>(P + q0.wt, P + q0.b, PooledIn<H>{L.partial, L.temb, L.cnt}, L.gh,
                                                      nullptr, tid, pre_q0);
    float* pre_q2 = nullptr;
    if constexpr (TR::on) { tr.vec(L.gh, H, tid); pre_q2 = tr.row(G); }
    matvec<MMB_ACT_GELU, H, G>(P + q2.wt, P + q2.b, L.gh, L.gcur, nullptr, tid, pre_q2);
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
};


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
};

// embedded inputs of one particle as padded 32-feature tiles (embedder.py:74-80 + apply_mask)
template <class DM>
__device__ __forceinline__ void embed_tiles(f32x16& tc, f32x16& td, const float (&xp)[4], int kp, float m, const Lds& L,
                                            int half) {
  ffor<16>([&](auto rc) {
    constexpr int r = decltype(rc)::v;
    const int f = (r & 3) + 8 * (r >> 2) + 4 * half;
    float c = 0.0f, d = 0.0f;
    if (f < DM::EC) {
      float s = 0.0f;
#pragma unroll
      for (int dd = 0; dd < DM::DC; ++dd) s += xp[dd] * L.cw[f * 4 + dd];
      c = (s + L.cb[f]) * m;
    }
    if (f < DM::ED) d = truncf(L.table[kp * 32 + f] * m);           // quirk Q1
    tc[r] = c;
    td[r] = d;
  });
}

// Whole MultiModalEPiC forward for this jet.  which: 0 continuous head, 1 discrete head, 2 both.
// hc: tile whose features 0..DC-1 are the drift; hd: tile whose features 0..V-1 are the logits.
// Training (TR = Train): slots in execution order -- per branch [projection 9, layers 8 each, output 1],
// the concatenated global vector 1, fused encoder [layers 8 each, output 1], heads [pre, hidden] x 2.
template <class DM, class TR>
__device__ __forceinline__ void forward_jet(const FusedArgs& A, Lds& L, float4* S, const float (&xp)[4], int kp, float m,
                                            int which, f32x16& hc, f32x16& hd, int tid, TR& tr) {
  constexpr int HBT = DM::HBT, HFT = DM::HFT;
  const int lane = tid & 63, half = lane >> 5;
  int li = 0, ci = 0;
  f32x16 tc, td;
  FE_PROF(L, PF_STEP);
  embed_tiles<DM>(tc, td, xp, kp, m, L, half);
  FE_PROF(L, PF_EMBED);
  // S (LDS, HFT tiles) = cat[h1, h2]: fused-encoder input, its skip connection AND the heads' input
  // ---- two branch encoders (multimodal_epic.py:173-175)
  ffor<2>([&](auto bc) {
    constexpr int br = decltype(bc)::v;
    f32x16 x[HBT], sk[HBT], out[HBT];
    Epic<HBT, DM::ET, DM::GB>::template project<1>(x, [&](auto) -> const f32x16& { return br == 0 ? tc : td; }, A.P,
                                                   A.D, li, L, m, tid, A.ctab, ci, tr);
    if (tid < DM::GB) L.gskip[tid] = L.gcur[tid];
    ffor<HBT>([&](auto t) { sk[decltype(t)::v] = x[decltype(t)::v]; });
    __syncthreads();
    const int nbb = br == 0 ? A.nb_branch : A.nb_branch1;
    for (int i = 0; i < nbb; ++i)
      Epic<HBT, DM::ET, DM::GB>::layer(x, [&](auto t) -> const f32x16& { return sk[decltype(t)::v]; },
                                       A.use_skip != 0, A.P, A.D, li, L, m, tid, A.ctab, ci, tr);
    Epic<HBT, DM::ET, DM::GB>::template output<HBT, DM::HB>(out, x, A.P, A.D, li, L, m, tid, A.ctab, ci, tr);
    ffor<HBT>([&](auto t) { park_tile(S, br * HBT + decltype(t)::v, out[decltype(t)::v], tid); });
    if (tid < DM::GB) L.gcat[br * DM::GB + tid] = L.gcur[tid];
    __syncthreads();
  });
  // ---- fused encoder, no projection (K2: nb_fused_exec = constructed - 1)
  if (tid < DM::GF) { L.gcur[tid] = L.gcat[tid]; L.gskip[tid] = L.gcat[tid]; }
  if constexpr (TR::on) tr.vec(L.gcat, DM::GF, tid);
  __syncthreads();
  f32x16 x[HFT], f[HFT];
  ffor<HFT>([&](auto t) { x[decltype(t)::v] = fetch_tile(S, decltype(t)::v, tid); });
  for (int i = 0; i < A.nb_fused_exec; ++i)
    Epic<HFT, DM::ET, DM::GF>::layer(x, [&](auto t) -> f32x16 { return fetch_tile(S, decltype(t)::v, tid); },
                                     A.use_skip != 0, A.P, A.D, li, L, m, tid, A.ctab, ci, tr);
  Epic<HFT, DM::ET, DM::GF>::template output<HFT, 2 * DM::HB>(f, x, A.P, A.D, li, L, m, tid, A.ctab, ci, tr);
  // ---- heads (multimodal_epic.py:187-189): cat[t, f_i, h_i] -> GELU -> out
  const LDesc c0 = A.D[li], c2 = A.D[li + 1], d0 = A.D[li + 2], d2 = A.D[li + 3];
  const WRsrc R = weight_rsrc(A.P);
  FE_PROF(L, PF_EPI);
  const int ci_heads = ci;                               // table slots: [.. , continuous head, discrete head]
  if (which != 1) {
    ci = ci_heads;
    const float* cv = time_vec<DM::ET, DM::HB>(A.ctab, ci, A.P + c0.wt, L, tid);
    AFrag<HBT> pre0;
    AFrag<1> pre2;
    mma_prefetch(pre0, R, c0.wx, lane);
    mma_prefetch(pre2, R, c2.wx, lane);
    f32x16 h[HBT];
    acc_init<HBT>(h, m, cv, A.P + c0.b, half);
    mma_x<HFT, HBT>(h, [&](auto fc) -> f32x16 {
      constexpr int t = decltype(fc)::v;
      if constexpr (t < HBT) return f[t]; else return fetch_tile(S, t - HBT, tid); }, R, c0.wx, lane, pre0);
    if constexpr (TR::on) tr.template tiles<HBT>(h, half);
    act_tiles<HBT, MMB_ACT_GELU>(h);
    if constexpr (TR::on) tr.template tiles<HBT>(h, half);
    f32x16 o[1];
    acc_init<1>(o, 0.0f, nullptr, A.P + c2.b, half);
    mma_x<HBT, 1>(o, [&](auto fc) -> const f32x16& { return h[decltype(fc)::v]; }, R, c2.wx, lane, pre2);
    hc = o[0];
  }
  if (which != 0) {
    ci = ci_heads + 1;
    const float* cv = time_vec<DM::ET, DM::HB>(A.ctab, ci, A.P + d0.wt, L, tid);
    AFrag<HBT> pre0;
    AFrag<1> pre2;
    mma_prefetch(pre0, R, d0.wx, lane);
    mma_prefetch(pre2, R, d2.wx, lane);
    f32x16 h[HBT];
    acc_init<HBT>(h, m, cv, A.P + d0.b, half);
    mma_x<HFT, HBT>(h, [&](auto fc) -> f32x16 {
      constexpr int t = decltype(fc)::v;
      if constexpr (t < HBT) return f[HBT + t]; else return fetch_tile(S, t, tid); }, R, d0.wx, lane, pre0);
    if constexpr (TR::on) tr.template tiles<HBT>(h, half);
    act_tiles<HBT, MMB_ACT_GELU>(h);
    if constexpr (TR::on) tr.template tiles<HBT>(h, half);
    f32x16 o[1];
    acc_init<1>(o, 0.0f, nullptr, A.P + d2.b, half);
    mma_x<HBT, 1>(o, [&](auto fc) -> const f32x16& { return h[decltype(fc)::v]; }, R, d2.wx, lane, pre2);
    hd = o[0];
  }
  FE_PROF(L, PF_HEAD);
}

template <class DM>
__device__ __forceinline__ void load_constants(const FusedArgs& A, Lds& L, int tid) {
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
template <class DM>
__global__ void __launch_bounds__(FE_THREADS, 2)
fused_forward_kernel(FusedArgs A, const float* __restrict__ x, const int64_t* __restrict__ k,
                     const float* __restrict__ temb, int64_t temb_stride, int which, float* __restrict__ head_c,
                     float* __restrict__ head_d) {
  __shared__ Lds L;
  __shared__ float4 S[DM::HFT * 4 * FE_THREADS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  load_constants<DM>(A, L, tid);
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    __syncthreads();
    if (tid < DM::ET) L.temb[tid] = temb[(int64_t)b * temb_stride + tid];
    if (tid == 0) L.cnt = (float)A.counts[b];
    const int p = 32 * wave + (lane & 31);
    const bool valid = p < A.N;
    const int64_t gp = (int64_t)b * A.N + (valid ? p : 0);
    const float m = valid ? A.maskf[gp] : 0.0f;
    float xp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DM::DC; ++d) xp[d] = valid ? x[gp * DM::DC + d] : 0.0f;
    const int kp = valid ? (int)k[gp] : 0;
    __syncthreads();
    f32x16 hc, hd;
    NoTrain nt;
    forward_jet<DM>(A, L, S, xp, kp, m, which, hc, hd, tid, nt);
    if (valid && which != 1 && half == 0) {
#pragma unroll
      for (int d = 0; d < DM::DC; ++d) head_c[gp * DM::DC + d] = hc[d];       // features 0..2 = regs 0..2, half 0
    }
    if (valid && which != 0) {
      float lg[DM::V];
      gather_logits<DM>(hd, lg, lane);
      if (half == 0) {
#pragma unroll
        for (int v = 0; v < DM::V; ++v) head_d[gp * DM::V + v] = lg[v];
      }
    }
  }
}

// ---- training forward: both heads + every activation of the tape to HBM ---------------------
template <class DM>
__global__ void __launch_bounds__(FE_THREADS, 2)
fused_train_forward_kernel(FusedArgs A, const float* __restrict__ x, const int64_t* __restrict__ k,
                           const float* __restrict__ temb, int64_t temb_stride, const SaveSlot* __restrict__ slots,
                           int n_slots, float drop_p, uint64_t seed, uint64_t offset0, float* __restrict__ head_c,
                           float* __restrict__ head_d) {
  __shared__ Lds L;
  __shared__ float4 S[DM::HFT * 4 * FE_THREADS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  load_constants<DM>(A, L, tid);
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    __syncthreads();
    if (tid < DM::ET) L.temb[tid] = temb[(int64_t)b * temb_stride + tid];
    if (tid == 0) L.cnt = (float)A.counts[b];
    const int p = 32 * wave + (lane & 31);
    const bool valid = p < A.N;
    const int64_t gp = (int64_t)b * A.N + (valid ? p : 0);
    const float m = valid ? A.maskf[gp] : 0.0f;
    float xp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DM::DC; ++d) xp[d] = valid ? x[gp * DM::DC + d] : 0.0f;
    const int kp = valid ? (int)k[gp] : 0;
    __syncthreads();
    Train tr{slots, n_slots, 0, b, A.N, p, drop_p, seed, offset0};
    f32x16 hc, hd;
    forward_jet<DM>(A, L, S, xp, kp, m, 2, hc, hd, tid, tr);
    if (valid && half == 0) {
#pragma unroll
      for (int d = 0; d < DM::DC; ++d) head_c[gp * DM::DC + d] = hc[d];
    }
    float lg[DM::V];
    gather_logits<DM>(hd, lg, lane);
    if (valid && half == 0) {
#pragma unroll
      for (int v = 0; v < DM::V; ++v) head_d[gp * DM::V + v] = lg[v];
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

template <class DM>
__global__ void __launch_bounds__(FE_THREADS, 2)
fused_generate_kernel(FusedArgs A, GenArgs Gn, float* __restrict__ x, int64_t* __restrict__ k) {
  __shared__ Lds L;
  __shared__ float4 S[DM::HFT * 4 * FE_THREADS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  constexpr int V = DM::V;
#ifdef MMB_FE_PROF
  if (tid == 0) L.tlast = clock64();
#endif
  load_constants<DM>(A, L, tid);
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    __syncthreads();
    if (tid == 0) L.cnt = (float)A.counts[b];
    const int p = 32 * wave + (lane & 31);
    const bool valid = p < A.N;
    const int64_t gp = (int64_t)b * A.N + (valid ? p : 0);
    const float m = valid ? A.maskf[gp] : 0.0f;
    float xp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DM::DC; ++d) xp[d] = valid ? x[gp * DM::DC + d] : 0.0f;
    int kp = valid ? (int)k[gp] : 0;
    for (int s = 0; s < Gn.steps; ++s) {
      __syncthreads();
      if (tid < DM::ET) L.temb[tid] = Gn.temb_steps[(int64_t)s * DM::ET + tid];
      __syncthreads();
      A.ctab = Gn.ctab_steps ? Gn.ctab_steps + (int64_t)s * Gn.n_ctab * 128 : nullptr;
      const float tt = Gn.times[s];
      f32x16 hc, hd;
      // The particle state is only needed at the two ends of a forward: it waits in LDS meanwhile
      // (each lane reads back its own slot), which keeps it out of the register allocator's way.
      auto park = [&]() {
#pragma unroll
        for (int d = 0; d < DM::DC; ++d) L.park[d][tid] = xp[d];
        L.park[3][tid] = __int_as_float(kp);
      };
      auto unpark = [&]() {
#pragma unroll
        for (int d = 0; d < DM::DC; ++d) xp[d] = L.park[d][tid];
        kp = __float_as_int(L.park[3][tid]);
      };
      // Euler step on the drift head (solvers.py:21-25); padded slots move too (K3)
      park();
      NoTrain nt;
      forward_jet<DM>(A, L, S, xp, kp, m, 0, hc, hd, tid, nt);
      unpark();
      {
        float v[3] = {hc[0], hc[1], hc[2]};
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        float sig = 0.0f;
        if (Gn.em) {                                     // both lane halves draw the same increments
          sig = Gn.em_sigma * sqrtf(tt * (1.0f - tt));
          if (Gn.delta_w != nullptr) {
            if (valid) {
              const float* dw = Gn.delta_w + (((int64_t)s * A.B + b) * A.N + p) * DM::DC;
#pragma unroll
              for (int d = 0; d < DM::DC; ++d) w[d] = dw[d];
            }
          } else {
            uint32_t q[4];
            philox4(Gn.seed, Gn.offset + (uint64_t)s, (uint64_t)gp, 0xE001u, q);
            // hardware v_sin/v_cos (argument in revolutions): distribution-level draws, and no
            // private-memory argument reduction inside the generation loop
            const float r0 = sqrtf(-2.0f * __logf(u01_open0(q[0]))), r1 = sqrtf(-2.0f * __logf(u01_open0(q[2])));
            w[0] = r0 * __builtin_amdgcn_cosf(u01(q[1]));
            w[1] = r0 * __builtin_amdgcn_sinf(u01(q[1]));
            w[2] = r1 * __builtin_amdgcn_cosf(u01(q[3]));
          }
        }
#pragma unroll
        for (int d = 0; d < DM::DC; ++d) {
          const float o = __shfl_xor(v[d], 32, 64);
          const float drift = half == 0 ? v[d] : o;
          xp[d] += Gn.em ? (Gn.dt * drift + sig * w[d]) : Gn.dt * drift;
        }
      }
      __syncthreads();
      // tau-leap on the logits head, which sees the UPDATED x at the same t (solvers.py:59-81)
      park();
      forward_jet<DM>(A, L, S, xp, kp, m, 1, hc, hd, tid, nt);
      unpark();
      float r[V];
      gather_logits<DM>(hd, r, lane);
      kp = jump_step(r, kp, tt, Gn, s == Gn.steps - 1, valid, A.B, A.N, b, p, s, gp);
    }
    if (valid && half == 0) {
#pragma unroll
      for (int d = 0; d < DM::DC; ++d) x[gp * DM::DC + d] = xp[d];
      k[gp] = kp;
    }
  }
}

// ==================================================================== UniModalEPiC ====
// Discrete-only model (multimodal_epic.py:44-64): ONE EPiC encoder with projection on
// cat[t, t, trunc(k_emb*mask)] (the time embedding enters twice: multimodal_epic.py:47-54 +
// epic.py:155 -- both column blocks are summed into one time vector on the host), whose output
// layer emits the V logits directly (no heads).

// Training (TR = Train) slots: projection 9, layers 8 each (the V logits of the output layer go to the caller).
template <class DM, class TR>
__device__ __forceinline__ void forward_uni(const FusedArgs& A, Lds& L, float4* S, const float* tab, int kp, float m,
                                            f32x16& hd, int tid, TR& nt) {
  constexpr int HT = DM::HT;
  const int lane = tid & 63, half = lane >> 5;
  int li = 0, ci = 0;
  FE_PROF(L, PF_STEP);
  f32x16 td[DM::EDT];
  ffor<DM::EDT>([&](auto tc) { constexpr int t = decltype(tc)::v;
    ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
      const float4 e = *reinterpret_cast<const float4*>(tab + kp * DM::EDP + 32 * t + 8 * q + 4 * half);
      td[t][4 * q + 0] = truncf(e.x * m); td[t][4 * q + 1] = truncf(e.y * m);      // quirk Q1
      td[t][4 * q + 2] = truncf(e.z * m); td[t][4 * q + 3] = truncf(e.w * m); }); });
  FE_PROF(L, PF_EMBED);
  f32x16 x[HT];
  Epic<HT, DM::ET, DM::G>::template project<DM::EDT>(x, [&](auto fc) -> const f32x16& { return td[decltype(fc)::v]; },
                                                    A.P, A.D, li, L, m, tid, A.ctab, ci, nt);
  if (tid < DM::G) L.gskip[tid] = L.gcur[tid];
  ffor<HT>([&](auto t) { park_tile(S, decltype(t)::v, x[decltype(t)::v], tid); });
  __syncthreads();
  for (int i = 0; i < A.nb_branch; ++i)
    Epic<HT, DM::ET, DM::G>::layer(x, [&](auto t) -> f32x16 { return fetch_tile(S, decltype(t)::v, tid); },
                                   A.use_skip != 0, A.P, A.D, li, L, m, tid, A.ctab, ci, nt);
  f32x16 o[1];
  Epic<HT, DM::ET, DM::G>::template output<1, DM::V>(o, x, A.P, A.D, li, L, m, tid, A.ctab, ci, nt);
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

template <class DM>
__global__ void __launch_bounds__(FE_THREADS, 2)
fused_uni_forward_kernel(FusedArgs A, const int64_t* __restrict__ k, const float* __restrict__ temb,
                         int64_t temb_stride, float* __restrict__ head_d) {
  static_assert(DM::V == 8, "fused path is instantiated for vocab_size 8");
  __shared__ Lds L;
  __shared__ float4 S[DM::HT * 4 * FE_THREADS];
  __shared__ __attribute__((aligned(16))) float tab[DM::V * DM::EDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  load_table_uni<DM>(A, tab, tid);
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    __syncthreads();
    if (tid < DM::ET) L.temb[tid] = temb[(int64_t)b * temb_stride + tid];
    if (tid == 0) L.cnt = (float)A.counts[b];
    const int p = 32 * wave + (lane & 31);
    const bool valid = p < A.N;
    const int64_t gp = (int64_t)b * A.N + (valid ? p : 0);
    const float m = valid ? A.maskf[gp] : 0.0f;
    const int kp = valid ? (int)k[gp] : 0;
    __syncthreads();
    f32x16 hd;
    NoTrain nt;
    forward_uni<DM>(A, L, S, tab, kp, m, hd, tid, nt);
    float lg[8];
    gather8(hd, lg, lane);
    if (valid && half == 0) {
#pragma unroll
      for (int v = 0; v < 8; ++v) head_d[gp * 8 + v] = lg[v];
    }
  }
}

// training forward of the discrete-only model: logits + every activation of the tape to HBM
template <class DM>
__global__ void __launch_bounds__(FE_THREADS, 2)
fused_uni_train_forward_kernel(FusedArgs A, const int64_t* __restrict__ k, const float* __restrict__ temb,
                               int64_t temb_stride, const SaveSlot* __restrict__ slots, int n_slots, float drop_p,
                               uint64_t seed, uint64_t offset0, float* __restrict__ head_d) {
  static_assert(DM::V == 8, "fused path is instantiated for vocab_size 8");
  __shared__ Lds L;
  __shared__ float4 S[DM::HT * 4 * FE_THREADS];
  __shared__ __attribute__((aligned(16))) float tab[DM::V * DM::EDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  load_table_uni<DM>(A, tab, tid);
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    __syncthreads();
    if (tid < DM::ET) L.temb[tid] = temb[(int64_t)b * temb_stride + tid];
    if (tid == 0) L.cnt = (float)A.counts[b];
    const int p = 32 * wave + (lane & 31);
    const bool valid = p < A.N;
    const int64_t gp = (int64_t)b * A.N + (valid ? p : 0);
    const float m = valid ? A.maskf[gp] : 0.0f;
    const int kp = valid ? (int)k[gp] : 0;
    __syncthreads();
    Train tr{slots, n_slots, 0, b, A.N, p, drop_p, seed, offset0};
    f32x16 hd;
    forward_uni<DM>(A, L, S, tab, kp, m, hd, tid, tr);
    float lg[8];
    gather8(hd, lg, lane);
    if (valid && half == 0) {
#pragma unroll
      for (int v = 0; v < 8; ++v) head_d[gp * 8 + v] = lg[v];
    }
  }
}

template <class DM>
__global__ void __launch_bounds__(FE_THREADS, 2)
fused_uni_generate_kernel(FusedArgs A, GenArgs Gn, int64_t* __restrict__ k) {
  __shared__ Lds L;
  __shared__ float4 S[DM::HT * 4 * FE_THREADS];
  __shared__ __attribute__((aligned(16))) float tab[DM::V * DM::EDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
#ifdef MMB_FE_PROF
  if (tid == 0) L.tlast = clock64();
#endif
  load_table_uni<DM>(A, tab, tid);
  for (int b = blockIdx.x; b < A.B; b += gridDim.x) {
    __syncthreads();
    if (tid == 0) L.cnt = (float)A.counts[b];
    const int p = 32 * wave + (lane & 31);
    const bool valid = p < A.N;
    const int64_t gp = (int64_t)b * A.N + (valid ? p : 0);
    const float m = valid ? A.maskf[gp] : 0.0f;
    int kp = valid ? (int)k[gp] : 0;
    for (int s = 0; s < Gn.steps; ++s) {
      __syncthreads();
      if (tid < DM::ET) L.temb[tid] = Gn.temb_steps[(int64_t)s * DM::ET + tid];
      __syncthreads();
      A.ctab = Gn.ctab_steps ? Gn.ctab_steps + (int64_t)s * Gn.n_ctab * 128 : nullptr;
      f32x16 hd;
      NoTrain nt;
      forward_uni<DM>(A, L, S, tab, kp, m, hd, tid, nt);
      float r[8];
      gather8(hd, r, lane);
      kp = jump_step(r, kp, Gn.times[s], Gn, s == Gn.steps - 1, valid, A.B, A.N, b, p, s, gp);
    }
    if (valid && half == 0) k[gp] = kp;
  }
}



static int check_fused(const char* name, int B, int N, const void* P, const void* D, const void* maskf,
                       const void* counts) {
  MMB_REQUIRE(B > 0 && N > 0 && N <= 128, "%s: needs 0 < N <= 128 (got B=%d N=%d)", name, B, N);
  MMB_REQUIRE(P && D && maskf && counts, "%s: NULL argument", name);
  return MMB_OK;
}

}  // namespace mmb

using namespace mmb;

// variant: 0 = S3 dims (Et32 Ec16 Ed16 H64/64/128 G16/16/32), 1 = "mid" dims (Et16 Ec8 Ed8 H32/32/64 G8/8/16)
extern "C" int mmb_fused_epic_forward(int variant, int B, int N, const float* packed, const int32_t* desc,
                                      int nb_branch, int nb_branch1, int nb_fused_exec, int use_skip, int off_cw, int off_cb,
                                      int off_table, const float* maskf, const int32_t* counts, const float* x,
                                      const int64_t* k, const float* temb, int64_t temb_stride, int which,
                                      float* head_c, float* head_d, void* stream) {
  int rc = check_fused("mmb_fused_epic_forward", B, N, packed, desc, maskf, counts);
  if (rc) return rc;
  MMB_REQUIRE(variant >= 0 && variant < N_MULTI_VARIANTS, "mmb_fused_epic_forward: unknown variant %d", variant);
  MMB_REQUIRE(x && k && temb, "mmb_fused_epic_forward: NULL state");
  MMB_REQUIRE(which >= 0 && which <= 2, "mmb_fused_epic_forward: which=%d", which);
  if (which != 1) MMB_REQUIRE(head_c, "mmb_fused_epic_forward: head_c is NULL");
  if (which != 0) MMB_REQUIRE(head_d, "mmb_fused_epic_forward: head_d is NULL");
  FusedArgs A{nullptr, packed, reinterpret_cast<const LDesc*>(desc), nb_branch, nb_fused_exec, use_skip, off_cw, off_cb,
              off_table, B, N, maskf, counts, nb_branch1};
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
                                            uint64_t offset0, float* head_c, float* head_d, void* stream) {
  int rc = check_fused("mmb_fused_epic_train_forward", B, N, packed, desc, maskf, counts);
  if (rc) return rc;
  MMB_REQUIRE(variant >= 0 && variant < N_MULTI_VARIANTS, "mmb_fused_epic_train_forward: unknown variant %d", variant);
  MMB_REQUIRE(x && k && temb && head_c && head_d && slots, "mmb_fused_epic_train_forward: NULL argument");
  MMB_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "mmb_fused_epic_train_forward: dropout p=%f outside [0,1)", drop_p);
  const int expect = 2 * 10 + 8 * (nb_branch + nb_branch1) + 1 + 8 * nb_fused_exec + 1 + 4;
  MMB_REQUIRE(n_slots == expect, "mmb_fused_epic_train_forward: %d save slots given, the architecture has %d", n_slots,
              expect);
  FusedArgs A{nullptr, packed, reinterpret_cast<const LDesc*>(desc), nb_branch, nb_fused_exec, use_skip, off_cw, off_cb,
              off_table, B, N, maskf, counts, nb_branch1};
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
  int rc = check_fused("mmb_fused_uni_forward", B, N, packed, desc, maskf, counts);
  if (rc) return rc;
  MMB_REQUIRE(variant >= 0 && variant < N_UNI_VARIANTS, "mmb_fused_uni_forward: unknown variant %d", variant);
  MMB_REQUIRE(k && temb && head_d, "mmb_fused_uni_forward: NULL state");
  FusedArgs A{nullptr, packed, reinterpret_cast<const LDesc*>(desc), num_blocks, 0, use_skip, 0, 0, off_table, B, N,
              maskf, counts};
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
                                           uint64_t seed, uint64_t offset0, float* head_d, void* stream) {
  int rc = check_fused("mmb_fused_uni_train_forward", B, N, packed, desc, maskf, counts);
  if (rc) return rc;
  MMB_REQUIRE(variant >= 0 && variant < N_UNI_VARIANTS, "mmb_fused_uni_train_forward: unknown variant %d", variant);
  MMB_REQUIRE(k && temb && head_d && slots, "mmb_fused_uni_train_forward: NULL argument");
  MMB_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "mmb_fused_uni_train_forward: dropout p=%f outside [0,1)", drop_p);
  const int expect = 9 + 8 * num_blocks;
  MMB_REQUIRE(n_slots == expect, "mmb_fused_uni_train_forward: %d save slots given, the architecture has %d", n_slots,
              expect);
  FusedArgs A{nullptr, packed, reinterpret_cast<const LDesc*>(desc), num_blocks, 0, use_skip, 0, 0, off_table, B, N,
              maskf, counts};
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
                                      uint64_t offset, void* stream) {
  int rc = check_fused("mmb_fused_uni_generate", B, N, packed, desc, maskf, counts);
  if (rc) return rc;
  MMB_REQUIRE(variant >= 0 && variant < N_UNI_VARIANTS, "mmb_fused_uni_generate: unknown variant %d", variant);
  MMB_REQUIRE(k && times && temb_steps && steps > 0, "mmb_fused_uni_generate: NULL state / steps");
  MMB_REQUIRE(thermo_kind >= 0 && thermo_kind <= 4, "mmb_fused_uni_generate: bad thermostat");
  FusedArgs A{nullptr, packed, reinterpret_cast<const LDesc*>(desc), num_blocks, 0, use_skip, 0, 0, off_table, B, N,
              maskf, counts};
  GenArgs Gn{steps, times, temb_steps, ctab_steps, n_ctab, dt, thermo_kind, (float)(-(double)8 * (double)gamma), delta_n,
             seed, offset, 0, 0.0f, nullptr};
  const int grid = B < 4096 ? B : 4096;
  switch (variant) {
    case 0: hipLaunchKernelGGL(fused_uni_generate_kernel<UniS2>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, k); break;
    case 1: hipLaunchKernelGGL(fused_uni_generate_kernel<UniMid>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, k); break;
    case 2: hipLaunchKernelGGL(fused_uni_generate_kernel<UniH64>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, k); break;
    default: hipLaunchKernelGGL(fused_uni_generate_kernel<UniH32>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, k); break;
  }
  return check_launch("mmb_fused_uni_generate");
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
             em, em_sigma, delta_w};
  constexpr int HH = 64;
  const size_t lds = (size_t)(HH * 33 + 3 * HH * (HH + 1) + 32 * (HH + 1) + 4 * HH + 32 + 8 * (HH / 4)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    MMB_REQUIRE(hipFuncSetAttribute((const void*)toy_generate_kernel<64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) == hipSuccess, "mmb_toy_generate: cannot raise the dynamic LDS limit");
    attr_set = true;
  }
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

extern "C" int mmb_fused_epic_generate(int variant, int B, int N, const float* packed, const int32_t* desc,
                                       int nb_branch, int nb_branch1, int nb_fused_exec, int use_skip, int off_cw, int off_cb,
                                       int off_table, const float* maskf, const int32_t* counts, float* x, int64_t* k,
                                       int steps, const float* times, const float* temb_steps,
                                       const float* ctab_steps, int n_ctab, float dt,
                                       int thermo_kind, float gamma, const float* delta_n, uint64_t seed,
                                       uint64_t offset, int em, float em_sigma, const float* delta_w, void* stream) {
  int rc = check_fused("mmb_fused_epic_generate", B, N, packed, desc, maskf, counts);
  if (rc) return rc;
  MMB_REQUIRE(variant >= 0 && variant < N_MULTI_VARIANTS, "mmb_fused_epic_generate: unknown variant %d", variant);
  MMB_REQUIRE(x && k && times && temb_steps && steps > 0, "mmb_fused_epic_generate: NULL state / steps");
  MMB_REQUIRE(thermo_kind >= 0 && thermo_kind <= 4, "mmb_fused_epic_generate: bad thermostat");
  FusedArgs A{nullptr, packed, reinterpret_cast<const LDesc*>(desc), nb_branch, nb_fused_exec, use_skip, off_cw, off_cb,
              off_table, B, N, maskf, counts, nb_branch1};
  GenArgs Gn{steps, times, temb_steps, ctab_steps, n_ctab, dt, thermo_kind, (float)(-(double)8 * (double)gamma), delta_n, seed, offset,
             em, em_sigma, delta_w};
  const int grid = B < 4096 ? B : 4096;
  switch (variant) {
    case 0: hipLaunchKernelGGL(fused_generate_kernel<DimsS3>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, x, k); break;
    case 1: hipLaunchKernelGGL(fused_generate_kernel<DimsMid>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, x, k); break;
    case 2: hipLaunchKernelGGL(fused_generate_kernel<DimsWide>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, x, k); break;
    default: hipLaunchKernelGGL(fused_generate_kernel<DimsNarrow>, dim3(grid), dim3(FE_THREADS), 0, as_stream(stream), A, Gn, x, k); break;
  }
  return check_launch("mmb_fused_epic_generate");
}
