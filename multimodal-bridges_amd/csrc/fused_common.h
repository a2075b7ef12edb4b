// Building blocks of the fused per-jet EPiC kernels (fused_epic.hip: forward / generation,
// fused_epic_bwd.hip: training backward): register-resident transposed activations in the 32x32 C/D layout of
// the MFMAs, fp32 GEMMs as six bf16-plane products (mma_x), packed A-operand weight fragments fetched with buffer loads, per-jet
// matvecs, butterfly pooling, and the row-major save/load slots of the training path.
#pragma once
#include "common.h"
#include <utility>

namespace mmb {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int I> struct IF { static constexpr int v = I; };
template <typename F, int... Is>
__device__ __forceinline__ void ffor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IF<Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void ffor(F&& f) { ffor_impl(f, std::make_integer_sequence<int, N>{}); }

// Waves per workgroup = particle tiles per jet.  The translation unit fused_epic.hip is compiled twice: with 4 waves
// (N <= 128, two workgroups per CU: every kernel, the training ones included) and, through fused_epic_w5.hip, with 5
// (N <= 160 -- AOJ's 150-particle clouds --, one workgroup per CU: forward and generation kernels only).
#ifndef MMB_FE_WAVES
#define MMB_FE_WAVES 4
#endif
constexpr int FE_WAVES = MMB_FE_WAVES;
constexpr int FE_THREADS = 64 * FE_WAVES;
static_assert(FE_WAVES == 4 || FE_WAVES == 5, "fused EPiC kernels: 4 or 5 waves per jet");

// A value the optimiser cannot see through: everything derived from it is computed where it is used instead of being
// hoisted out of the grid-point loop and kept -- i.e. spilled -- across the whole kernel (lane offsets of every matvec
// shape, LDS addresses: 40 registers of them were going to scratch and coming back behind vmcnt(0) waits).
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}


// ------------------------------------------------------------------ small helpers ------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// sum over the 32 lanes that share (lane >> 5); every lane of the half ends up with the sum
__device__ __forceinline__ float half_sum32(float v) {
  v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]  (xor 1)
  v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]  (xor 2)
  v += dpp_mov<0x141>(v);    // row_half_mirror      (other quad of the 8)
  v += dpp_mov<0x140>(v);    // row_mirror           (other 8 of the 16)
  v += __shfl_xor(v, 16, 64);
  return v;
}

// erf without divergence: the library's erff branches on |a| and reduces the range of its exp by hand (~45
// instructions with the exec-mask traffic when a wave holds both ranges, and the fused kernels were spending a third
// of their vector issue there).  Both fits are evaluated and one is selected: 1 - exp(p(|a|)) above 0.927734375,
// a + a q(a^2) below; v_exp_f32 for the exponential (its argument is <= 0).  Measured against erf in double over
// [-12, 12] in steps of 1e-5: max relative error 8.7e-8 (below one ulp), erf(+-inf) = +-1, denormal-small a -> 2a/sqrt(pi).
__device__ __forceinline__ float erf_f(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(r * 1.4426950408889634f), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  q = fmaf(q, a, a);
  return t > 0.927734375f ? big : q;
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_f(x * 0.70710678118654752440f)); }
// (as max(x, 0.01 x) it is one vector instruction less and measured SLOWER twice, rounds 1 and 3: S3 -0.8 %, S2 -3.5 %;
// fmaxf brings a canonicalisation with it and the schedule around the MFMA stream changes)
__device__ __forceinline__ float lrelu_f(float x) { return x > 0.0f ? x : 0.01f * x; }
template <int ACT>
__device__ __forceinline__ float act_f(float x) {
  if constexpr (ACT == MMB_ACT_GELU) return gelu_f(x);
  if constexpr (ACT == MMB_ACT_LRELU) return lrelu_f(x);
  return x;
}

// per-layer descriptor: offsets (in floats) into the packed parameter buffer, -1 = absent
struct LDesc {
  int wx;   // MFMA part, packed [KT][2][OT][3 planes][64][8 bf16]  (per-particle layers; see mma_x)
  int wt;   // transposed [Kt][O]: time part (per-particle layers) or the whole W^T (per-jet layers)
  int wg;   // transposed [G][O]: global-vector part
  int b;    // bias [O] (padded to a multiple of 32 for per-particle layers)
};

constexpr int MAX_DESC = 96;        // layer descriptors cached in LDS (S3: 59, S2: 45)
struct alignas(16) Lds {
  float temb[128];
  float gcur[64], gskip[64], gnew[64], gcat[64];
  float partial[FE_WAVES * 128];
  float gh[128];
  float cvec[128], dvec[128], bvec[128];   // per-layer time vector, global-vector part + bias, second bias
  float pooled[400];                // [mean | 0.01 sum | tail]: the input of fc_glob1 / mlp_global.0 (MV::vidx layout)
  float cw[32 * 4], cb[32];       // continuous embedding weight [Ec][Dc<=4], bias
  float table[8 * 32];            // discrete embedding table [V<=8][Ed<=32]
  float cnt;
  int red[8];
  LDesc desc[MAX_DESC];
#ifdef MMB_FE_PROF
  unsigned long long tlast;
#endif
};

// exchange with the lane whose index differs in bit BIT
template <int BIT>
__device__ __forceinline__ float lane_xchg(float v) {
  if constexpr (BIT == 0) return dpp_mov<0xB1>(v);            // quad_perm [1,0,3,2]
  else if constexpr (BIT == 1) return dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
  else if constexpr (BIT == 3) return dpp_mov<0x128>(v);      // row_ror:8 == xor 8 inside a row of 16
  else if constexpr (BIT == 2)
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x101F));   // xor 4
  else if constexpr (BIT == 4)
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // xor 16
  else return __shfl_xor(v, 32, 64);
}

constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x / 2); }

// ---- training forward: every tensor the reverse pass needs goes to HBM (row-major, the layout of
// the layered path's tape, encoders/backward.py) straight from the accumulator registers, and the
// inverted dropout of epic.py:196-206 is applied with the layered path's Philox indexing.
struct SaveSlot {
  float* p;        // (rows, ld) row-major destination; rows = B*N (per-particle) or B (per-jet)
  int64_t ld;      // row pitch in floats (a column slice of a wider buffer has ld > width)
  int64_t n;       // floats available from p: stores are clipped to it whatever the slot order
};
struct NoTrain {
  static constexpr bool on = false;
};

// Keep-bits of the inverted dropout for the HT*16 values a lane holds of a (rows, 32*HT) activation: bit
// 16 t + r belongs to register r of tile t.  Same Philox stream and indexing as mmb_linear_cat's epilogue
// (element (row, col) -> counter (row*H + col) >> 2, word col & 3): a lane's registers 4q..4q+3 are four
// consecutive columns, i.e. ONE counter.  Deliberately a rolled loop: unrolled, the HT*4 Philox states are
// all live at once and push the MFMA loops around it into scratch.
template <int HT>
__device__ __forceinline__ uint64_t dropout_keep_bits(float p, uint64_t seed, uint64_t off, uint64_t row, int half) {
  uint64_t bits = 0;
#pragma unroll 1
  for (int tq = 0; tq < HT * 4; ++tq) {
    const uint64_t idx = row * (32 * HT) + (uint64_t)(32 * (tq >> 2) + 8 * (tq & 3) + 4 * half);
    uint32_t rr[4];
    philox4(seed, off, idx >> 2, 0xD0D0u, rr);
    const uint32_t nib = (u01(rr[0]) < p ? 0u : 1u) | (u01(rr[1]) < p ? 0u : 2u) | (u01(rr[2]) < p ? 0u : 4u) |
                         (u01(rr[3]) < p ? 0u : 8u);
    bits |= (uint64_t)nib << (4 * tq);
  }
  return bits;
}
struct Train {
  static constexpr bool on = true;
  const SaveSlot* tab;
  int n_slots, si;
  int b, N, p;               // jet, particles per jet, this lane's particle
  float drop_p;
  uint64_t seed, off;
  int dead = 0;              // 1: a padded slot -- its rows of the tape are never read, the stores go beyond the buffer

  __device__ __forceinline__ SaveSlot next() {
    const int i = si < n_slots ? si : n_slots - 1;
    ++si;
    return tab[i];
  }
  // HT tiles in C/D layout -> rows [b*N, b*N+N) of the slot; lanes of padded particles (p >= N) fall
  // outside the buffer range and the hardware drops their stores
  template <int HT>
  __device__ __forceinline__ void tiles(const f32x16 (&x)[HT], int half) { tiles_to<HT>(next(), x, half); }
  // ... into a slot taken earlier with next(): the stores can then be issued AFTER the GEMM that consumes the tiles
  // (vmcnt counts loads and stores in one order on gfx9: a store issued just before a GEMM makes every weight
  // fragment load of that GEMM wait for the store's acknowledgement)
  template <int HT>
  __device__ __forceinline__ void tiles_to(const SaveSlot& s, const f32x16 (&x)[HT], int half) {
    const int64_t base = (int64_t)b * N * s.ld;
    int64_t room = s.n - base, want = (int64_t)N * s.ld;
    room = room < want ? room : want;
    room = room < 0 ? 0 : room;
    const __amdgpu_buffer_rsrc_t R = __builtin_amdgcn_make_buffer_rsrc(s.p + base, 0, (int)(room * 4), 0x00020000);
    const int vo = dead ? 0x7FFF0000 : (p * (int)s.ld + 4 * half) * 4;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v;
      ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
        const float4 v = make_float4(x[t][4 * q], x[t][4 * q + 1], x[t][4 * q + 2], x[t][4 * q + 3]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), R, vo, (32 * t + 8 * q) * 4, 0); }); });
  }
  // one 64-bit word per thread (sign bits of the HT*16 values a lane holds) -> word b*256 + tid of the slot;
  // only the reverse kernel reads it, with the same thread mapping
  __device__ __forceinline__ void bits(uint64_t v, int tid) {
    const SaveSlot s = next();
    const int64_t i = (int64_t)b * FE_THREADS + tid;
    if (2 * (i + 1) <= s.n) reinterpret_cast<uint64_t*>(s.p)[i] = v;
  }
  // W floats of a per-jet vector (LDS) -> row b of the slot
  __device__ __forceinline__ void vec(const float* v, int W, int tid) {
    const SaveSlot s = next();
    const int64_t i = (int64_t)b * s.ld + tid;
    if (tid < W && i < s.n) s.p[i] = v[tid];
  }
  // row b of the next slot as a pointer for a producer that writes it itself (W floats), or NULL
  __device__ __forceinline__ float* row(int W) {
    const SaveSlot s = next();
    const int64_t i = (int64_t)b * s.ld;
    return i + W <= s.n ? s.p + i : nullptr;
  }
  __device__ __forceinline__ float keep(uint32_t r) const { return u01(r) < drop_p ? 0.0f : 1.0f / (1.0f - drop_p); }
  __device__ __forceinline__ float drop(float v, uint32_t r) const { return u01(r) < drop_p ? 0.0f : v / (1.0f - drop_p); }
};

// ---- training backward: the same slot table, walked by a kernel that LOADS saved activations and
// STORES the layer gradients dZ (row-major, consumed by the weight-gradient kernel) -------------
struct Rows {                 // rows [b*N, b*N+N) of a per-particle slot seen from this lane
  __amdgpu_buffer_rsrc_t R;
  int vo;                     // byte offset of this lane's particle row + its lane-half column group
};
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 load_tile(const Rows& r, int t) {
  f32x16 v;
  ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
    const auto w = __builtin_amdgcn_raw_buffer_load_b128(r.R, r.vo, (32 * t + 8 * q) * 4, 0);
    static_assert(sizeof(w) == 16, "b128 buffer load");
    const float4 a = __builtin_bit_cast(float4, w);
    v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w; });
  return v;
}
__device__ __forceinline__ void store_tile(const Rows& r, int t, const f32x16& x) {
  ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
    const float4 v = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r.R, r.vo, (32 * t + 8 * q) * 4, 0); });
}

struct Slots {
  const SaveSlot* tab;
  int n_slots, si;
  int b, N, p, half;
  int dead = 0;               // 1: this lane's row accesses fall outside every buffer (stores dropped, loads read 0)

  __device__ __forceinline__ SaveSlot at(int i) const { return tab[i < n_slots ? i : n_slots - 1]; }
  __device__ __forceinline__ SaveSlot next() { return at(si++); }
  // lanes of padded particles (p >= N) and anything past the slot's extent fall outside the buffer
  // range: their loads return 0 and their stores are dropped by the hardware
  __device__ __forceinline__ Rows rows_of(const SaveSlot& s, int col0 = 0) const {
    const int64_t base = (int64_t)b * N * s.ld + col0;
    int64_t room = s.n - base, want = (int64_t)N * s.ld - col0;
    room = room < want ? room : want;
    room = room < 0 ? 0 : room;
    Rows r;
    r.R = __builtin_amdgcn_make_buffer_rsrc(s.p + base, 0, (int)(room * 4), 0x00020000);
    r.vo = dead ? 0x7FFF0000 : (p * (int)s.ld + 4 * half) * 4;
    return r;
  }
  __device__ __forceinline__ Rows rows() { return rows_of(next()); }
  // row b of a per-jet slot (W floats) or NULL when it does not fit
  __device__ __forceinline__ float* row_of(const SaveSlot& s, int W) const {
    const int64_t i = (int64_t)b * s.ld;
    return i + W <= s.n ? s.p + i : nullptr;
  }
  __device__ __forceinline__ float* row(int W) { return row_of(next(), W); }
  __device__ __forceinline__ uint64_t bits(int tid) {
    const SaveSlot s = next();
    const int64_t i = (int64_t)b * FE_THREADS + tid;
    return 2 * (i + 1) <= s.n ? reinterpret_cast<const uint64_t*>(s.p)[i] : 0ull;
  }
};

// bit 16 t + r = (x[t][r] > 0): what the reverse pass needs of a leaky-relu input (built from 32-bit words:
// 64-bit shifts are multi-instruction on the VALU)
template <int HT>
__device__ __forceinline__ uint64_t sign_bits(const f32x16 (&x)[HT]) {
  uint32_t w[2] = {0u, 0u};
  ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v;
    ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v;
      w[t >> 1] |= (x[t][r] > 0.0f ? 1u : 0u) << (16 * (t & 1) + r); }); });
  return (uint64_t)w[0] | ((uint64_t)w[1] << 32);
}

// Per-jet matvec  out[o] = act( sum_k WT[k*O+o] v(k) + bias[o] ) + add[o]  with ONE barrier.
// The O outputs are split over the 4 waves (OW each); inside a wave lane = part*G + og: og picks VW
// consecutive outputs (one dwordx4/x2 per k), part picks every PARTS-th k.  The PARTS partial sums
// live in lanes of the SAME wave and are folded with lane exchanges (no LDS round trip, no barrier);
// the only barrier publishes `out` to the other waves.  vfn(k) yields the input element (LDS).
template <int K, int O>
struct WaveMV {
  static_assert((O & (O - 1)) == 0 && O >= 8 && O <= 256, "WaveMV: O must be a power of two in [8,256]");
  static constexpr int OW = O / 4;                       // outputs per wave
  static constexpr int VW = OW >= 4 ? 4 : OW;            // outputs per lane
  static constexpr int G = OW / VW;                      // lane groups along the outputs
  static constexpr int PARTS = 64 / G;                   // k-partitions inside the wave
  static constexpr int ITER = (K + PARTS - 1) / PARTS;
  static constexpr bool RAGGED = (K % PARTS) != 0;
  typedef float vec __attribute__((ext_vector_type(VW)));

  template <int ACT, typename VFn>
  static __device__ __forceinline__ void run(const float* __restrict__ WT, const float* __restrict__ bias, VFn&& vfn,
                                             float* out, const float* add, int tid, float* pre = nullptr) {
    const int lane = tid & 63, wave = tid >> 6;
    if (FE_WAVES == 4 || wave < 4) {                       // (five-wave build: the matvec is laid out over four waves)
    const int og = lane & (G - 1), part = lane / G;
    const int o0 = wave * OW + og * VW;
    const vec* wp = reinterpret_cast<const vec*>(WT + o0);
    vec bv;
    ffor<VW>([&](auto ec) { bv[decltype(ec)::v] = 0.0f; });
    if (bias != nullptr && part == 0) bv = *reinterpret_cast<const vec*>(bias + o0);
    vec acc;
    ffor<VW>([&](auto ec) { acc[decltype(ec)::v] = 0.0f; });
    constexpr int UN = 12;                               // loads in flight per lane
    ffor<(ITER + UN - 1) / UN>([&](auto cc) {
      constexpr int c0 = decltype(cc)::v * UN;
      constexpr int n = (ITER - c0) < UN ? (ITER - c0) : UN;
      vec w[n];
      ffor<n>([&](auto ic) {
        constexpr int it = c0 + decltype(ic)::v;
        int kk = part + it * PARTS;
        if constexpr (RAGGED && it == ITER - 1) kk = kk < K ? kk : K - 1;
        w[decltype(ic)::v] = wp[kk * (O / VW)];
      });
      ffor<n>([&](auto ic) {
        constexpr int it = c0 + decltype(ic)::v;
        const int kk = part + it * PARTS;
        float vv;
        if constexpr (RAGGED && it == ITER - 1) vv = kk < K ? vfn(kk) : 0.0f; else vv = vfn(kk);
        acc += w[decltype(ic)::v] * vv;
      });
    });
    // fold the PARTS partitions: lane bits log2(G) .. 5
    ffor<6 - ilog2(G)>([&](auto bc) {
      constexpr int bit = ilog2(G) + decltype(bc)::v;
      ffor<VW>([&](auto ec) { constexpr int e = decltype(ec)::v; acc[e] += lane_xchg<bit>(acc[e]); });
    });
    if (part == 0) {
      ffor<VW>([&](auto ec) {
        constexpr int e = decltype(ec)::v;
        if (pre) pre[o0 + e] = acc[e] + bv[e];
        float t = act_f<ACT>(acc[e] + bv[e]);
        if (add) t += add[o0 + e];
        out[o0 + e] = t;
      });
    }
    }
    __syncthreads();
  }
};

template <int ACT, int K, int O>
__device__ __forceinline__ void matvec(const float* __restrict__ WT, const float* __restrict__ bias, const float* v,
                                       float* out, const float* add, int tid, float* pre = nullptr) {
  WaveMV<K, O>::template run<ACT>(WT, bias, [&](int k) { return v[k]; }, out, add, tid, pre);
}

// A lane's 16 values of tile t parked in / fetched from LDS: slot (t,q) holds one float4 per thread.
__device__ __forceinline__ void park_tile(float4* S, int t, const f32x16& v, int tid) {
  ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
    S[(t * 4 + q) * FE_THREADS + tid] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]); });
}
__device__ __forceinline__ f32x16 fetch_tile(const float4* S, int t, int tid) {
  f32x16 v;
  ffor<4>([&](auto qc) { constexpr int q = decltype(qc)::v;
    const float4 a = S[(t * 4 + q) * FE_THREADS + tid];
    v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w; });
  return v;
}

// Weight fragments are fetched with BUFFER loads: one descriptor for the packed parameter block
// (SGPRs), the per-lane part of the address is the constant lane*16 in ONE VGPR, the fragment's
// position is a wave-uniform SGPR offset -- no 64-bit per-load address arithmetic and no address
// registers for the allocator to keep alive across the MFMA stream.
typedef __amdgpu_buffer_rsrc_t WRsrc;
__device__ __forceinline__ WRsrc weight_rsrc(const float* P) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, 0x7FFFFFFC, 0x00020000);
}
// float4 at packed float offset `woff` (uniform) + 64*f4 (uniform, float4 units) + this lane's float4
__device__ __forceinline__ float4 wload(WRsrc R, int woff, int f4, int lane16) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(R, lane16, (woff + 4 * f4) * 4, 0);
  static_assert(sizeof(v) == 16, "b128 buffer load");
  return __builtin_bit_cast(float4, v);
}

// First A-operand group (k-steps 0..3 of every output tile) of a packed weight matrix, fetched
// ahead of the GEMM that uses it -- while the previous GEMM or the per-jet chain runs -- so that no
// GEMM starts with an exposed L2 round trip.
// ---- fp32 GEMMs on the bf16 matrix cores.  gfx950 has no reduced-precision f32 MFMA mode and its f32 MFMA runs at
// 1/16 of the bf16 rate, so every f32 operand is split into three bf16 planes (hi + mid + lo = the whole 24-bit
// significand; each residual is exact in f32) and the product is the six partial products of weight >= 2^-16:
// Wl xh + Wm xm + Wh xl + Wm xh + Wh xm + Wh xh (each bf16 x bf16 product is exact in f32, accumulation in f32, smallest
// terms first).  The three dropped products are below 2^-24 of the result: the sum is an fp32 GEMM to within the
// rounding of its accumulation order.  v_mfma_f32_32x32x16_bf16 retires 16 k-values in 32 cycles against 8 x 64
// cycles of v_mfma_f32_32x32x2_f32: the six of them cost 192 cycles instead of 512.  Registers 8jj..8jj+7 of an input tile are one B operand (k-slot (half, j) =
// feature 16 jj + 8 (j >> 2) + 4 half + (j & 3)); the weights are pre-split and packed to match:
// [KT][2][OT][3 planes][64 lanes][8 bf16].
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
struct Frag16 {
  float4 p[3];
};
template <int OT>
struct AFrag {
  static constexpr int N = 2;
  Frag16 a[N];
};
template <int OT>
__device__ __forceinline__ void mma_prefetch(AFrag<OT>& f, WRsrc R, int woff, int lane) {
  ffor<AFrag<OT>::N>([&](auto gc) { constexpr int g = decltype(gc)::v;
    ffor<3>([&](auto pc) { constexpr int pl = decltype(pc)::v; f.a[g].p[pl] = wload(R, woff, (g * 3 + pl) * 64, lane * 16); }); });
}
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
// 8 floats -> three bf16x8 planes
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
  uint32_t ph[4], pm[4], pq[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = v[2 * j], b = v[2 * j + 1];
    ph[j] = pk_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, ph[j] << 16), rb = b - __builtin_bit_cast(float, ph[j] & 0xFFFF0000u);
    pm[j] = pk_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, pm[j] << 16), sb = rb - __builtin_bit_cast(float, pm[j] & 0xFFFF0000u);
    pq[j] = pk_bf16(sa, sb);
  }
  typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
  const u32x4v H = {ph[0], ph[1], ph[2], ph[3]}, M = {pm[0], pm[1], pm[2], pm[3]}, L = {pq[0], pq[1], pq[2], pq[3]};
  h = __builtin_bit_cast(bf16x8, H);
  m = __builtin_bit_cast(bf16x8, M);
  l = __builtin_bit_cast(bf16x8, L);
}
template <int KT, int OT, typename TileFn>
__device__ __forceinline__ void mma_x(f32x16 (&acc)[OT], TileFn&& tile, WRsrc R, int woff, int lane,
                                      const AFrag<OT>& pre) {
  constexpr int NG = KT * 2 * OT;
  constexpr int D = AFrag<OT>::N;
  static_assert(NG >= D, "prefetch deeper than the GEMM");
  const int lane16 = lane * 16;
  Frag16 ring[D + 1];
  ffor<D>([&](auto gc) { constexpr int g = decltype(gc)::v; ring[g] = pre.a[g]; });
  ffor<KT>([&](auto fc) {
    constexpr int ft = decltype(fc)::v;
    decltype(auto) xt = tile(IF<ft>{});
    ffor<2>([&](auto jc) {
      constexpr int jj = decltype(jc)::v;
      const float v[8] = {xt[8 * jj + 0], xt[8 * jj + 1], xt[8 * jj + 2], xt[8 * jj + 3],
                          xt[8 * jj + 4], xt[8 * jj + 5], xt[8 * jj + 6], xt[8 * jj + 7]};
      bf16x8 bh, bm, bl;
      split8(v, bh, bm, bl);
      ffor<OT>([&](auto oc) {
        constexpr int ot = decltype(oc)::v;
        constexpr int g = (ft * 2 + jj) * OT + ot;
        constexpr int cur = g % (D + 1);
        if constexpr (g + D < NG)
          ffor<3>([&](auto pc) { constexpr int pl = decltype(pc)::v;
            ring[(g + D) % (D + 1)].p[pl] = wload(R, woff, ((g + D) * 3 + pl) * 64, lane16); });
        const bf16x8 ah = __builtin_bit_cast(bf16x8, ring[cur].p[0]), am = __builtin_bit_cast(bf16x8, ring[cur].p[1]),
                     al = __builtin_bit_cast(bf16x8, ring[cur].p[2]);
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[ot], 0, 0, 0);
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[ot], 0, 0, 0);
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[ot], 0, 0, 0);
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[ot], 0, 0, 0);
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[ot], 0, 0, 0);
        acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[ot], 0, 0, 0);
      });
    });
  });
}

// ---- forward kernels, round 3: ONE weight-fragment stream per forward ----------------------------------------
// The ring holds the next RD fragments of the stream at all times: step g of a GEMM consumes slot g % RD and refills
// it at once with fragment g + RD -- of the same matrix, or (the last RD steps) the first fragments of the NEXT GEMM
// of the forward, so that no GEMM opens with an exposed L2 round trip and no separate per-GEMM prefetch registers
// exist (round 2: 2 + 2 + 3 fragments = 84 registers; now RD = 4 fragments = 48).  Every GEMM has a multiple of RD
// steps, so the slot phase is 0 at every GEMM boundary.  PF = false: the consumer that follows has no registers for
// the ring while its per-jet chain runs and primes it itself (ring_prime).
template <int RD>
struct WRing {
  Frag16 s[RD];
};
// fragment g of the matrix at woff -> one ring slot (three planes, 3 KiB per wave)
__device__ __forceinline__ void frag_load(Frag16& f, WRsrc R, int woff, int g, int lane16) {
#ifdef MMB_WHATIF_HOTFRAG      // developer timing build (wrong results on purpose): every fragment load hits the same 3 KiB
  g = 0;
#endif
#ifdef MMB_WHATIF_ONEPLANE     // developer timing build (wrong results on purpose): one third of the weight-fragment bytes
  f.p[0] = wload(R, woff, (g * 3) * 64, lane16);
  f.p[1] = f.p[0];
  f.p[2] = f.p[0];
#else
  ffor<3>([&](auto pc) { constexpr int pl = decltype(pc)::v; f.p[pl] = wload(R, woff, (g * 3 + pl) * 64, lane16); });
#endif
}
template <int RD>
__device__ __forceinline__ void ring_prime(WRing<RD>& ring, WRsrc R, int woff, int lane16) {
  ffor<RD>([&](auto gc) { constexpr int g = decltype(gc)::v; frag_load(ring.s[g], R, woff, g, lane16); });
}
// registers 8jj..8jj+7 of a tile -> the three planes of one B operand
template <int JJ>
__device__ __forceinline__ void split_half(const f32x16& xt, bf16x8& h, bf16x8& m, bf16x8& l) {
  const float v[8] = {xt[8 * JJ + 0], xt[8 * JJ + 1], xt[8 * JJ + 2], xt[8 * JJ + 3],
                      xt[8 * JJ + 4], xt[8 * JJ + 5], xt[8 * JJ + 6], xt[8 * JJ + 7]};
  split8(v, h, m, l);
}
// The three planes of one B operand (16 k-values x this lane's particle) as dwords: pair j holds values 2j, 2j+1
struct Planes {
  uint32_t h[4], m[4], l[4];
};
// pair J of k-block JJ of tile xt -> dword J of each plane (11 vector instructions)
template <int JJ, int J>
__device__ __forceinline__ void split_pair(const f32x16& xt, Planes& p) {
  const float a = xt[8 * JJ + 2 * J], b = xt[8 * JJ + 2 * J + 1];
  p.h[J] = pk_bf16(a, b);
  const float ra = a - __builtin_bit_cast(float, p.h[J] << 16), rb = b - __builtin_bit_cast(float, p.h[J] & 0xFFFF0000u);
  p.m[J] = pk_bf16(ra, rb);
  const float sa = ra - __builtin_bit_cast(float, p.m[J] << 16), sb = rb - __builtin_bit_cast(float, p.m[J] & 0xFFFF0000u);
  p.l[J] = pk_bf16(sa, sb);
}
__device__ __forceinline__ bf16x8 plane_of(const uint32_t (&d)[4]) {
  typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
  const u32x4v v = {d[0], d[1], d[2], d[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// One GEMM of the forward stream.  Software pipeline, pinned: while the six MFMAs of step (k-block kb, output tile ot)
// run, the wave splits its share of k-block kb+1 (4/OT pairs, 11 vector instructions each) and the ring slot is
// refilled.  sched_group_barrier writes the issue order down -- ONE MFMA, then the vector instructions that fit under
// it -- because the six MFMAs of a step depend on each other through the accumulator: in program order "6 MFMAs, then
// the split" (what the scheduler emits on its own) the wave sits on the matrix pipe for 6 x 32 cycles with nothing to
// issue and does the 44 instructions of the split afterwards with the pipe idle (SQ_VALU_MFMA_COEXEC_CYCLES was 11 %
// of the MFMA-busy cycles: the two pipes took turns).
template <int KT, int OT, int RD, bool PF, typename TileFn>
__device__ __forceinline__ void mma_r(f32x16 (&acc)[OT], TileFn&& tile, WRsrc R, int woff, int next_woff, int lane16,
                                      WRing<RD>& ring) {
  constexpr int KB = KT * 2, NG = KB * OT;
  static_assert(NG % RD == 0 && NG >= RD, "every GEMM must be a whole number of ring turns");
  static_assert(OT == 1 || OT == 2 || OT == 4, "mma_r: 1, 2 or 4 output tiles");
#ifdef MMB_WHATIF_NOGEMM          // developer timing build (wrong results on purpose): what the forward costs without its GEMMs
  return;
#endif
  constexpr int PP = 4 / OT;                   // pairs of the next k-block split per step
  Planes cur, nxt;
  {
    decltype(auto) x0 = tile(IF<0>{});
    split_pair<0, 0>(x0, cur); split_pair<0, 1>(x0, cur); split_pair<0, 2>(x0, cur); split_pair<0, 3>(x0, cur);
  }
  ffor<KB>([&](auto kc) {
    constexpr int kb = decltype(kc)::v;
    constexpr bool more = kb + 1 < KB;
    const bf16x8 bh = plane_of(cur.h), bm = plane_of(cur.m), bl = plane_of(cur.l);
    ffor<OT>([&](auto oc) {
      constexpr int ot = decltype(oc)::v;
      constexpr int g = kb * OT + ot;
      constexpr int slot = g % RD;
      const bf16x8 ah = __builtin_bit_cast(bf16x8, ring.s[slot].p[0]), am = __builtin_bit_cast(bf16x8, ring.s[slot].p[1]),
                   al = __builtin_bit_cast(bf16x8, ring.s[slot].p[2]);
      acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[ot], 0, 0, 0);
      acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[ot], 0, 0, 0);
      acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[ot], 0, 0, 0);
      acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[ot], 0, 0, 0);
      acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[ot], 0, 0, 0);
      acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[ot], 0, 0, 0);
      if constexpr (more) {
        decltype(auto) xn = tile(IF<(kb + 1) / 2>{});
        ffor<PP>([&](auto pc) { split_pair<(kb + 1) % 2, ot * PP + decltype(pc)::v>(xn, nxt); });
      }
      if constexpr (g + RD < NG) {
        frag_load(ring.s[slot], R, woff, g + RD, lane16);
      } else if constexpr (PF) {               // no run-time condition here: a branch in the stream makes the
        frag_load(ring.s[slot], R, next_woff, g + RD - NG, lane16);       // wait-count pass drain the queue at the join
      }
      // issue order of the step: MFMA, vector instructions under it, ... ; then the refill
      if constexpr (more) {
        constexpr int NV = 11 * PP, per = (NV + 5) / 6;
        ffor<6>([&](auto) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, per, 0);
        });
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
      // nothing crosses: without it the scheduler sinks the refill to its first use under register pressure (load,
      // vmcnt(0), MFMA -- the ring undone)
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (more) cur = nxt;
  });
}

// ---- per-jet matvecs of the chain, round 3.  out[o] = act(sum_k W^T[k][o] v(k) + b[o]) (+ add[o]).
// Lane = og * PARTS + part.  The PARTS partial sums of an output group sit in ADJACENT lanes, so the fold is DPP only
// for PARTS <= 16 (quad_perm, quad_perm, row_half_mirror, row_mirror: no LDS-pipe exchange, no lgkmcnt wait).  A lane
// owns ITER CONSECUTIVE rows k = part*ITER + it: its slice of the input vector is ITER/4 ds_read_b128 issued together
// (round 2 and the first version of this: one ds_read_b32 + wait per row, 36 dependent LDS latencies in fc_glob1 of a
// 128-wide encoder).  The input vector is stored with one slot group of ITERP >= ITER floats per partition
// (vidx): ITERP = ITER rounded up to the read width, so that the reads stay 16-byte aligned for any K.  The weights
// come through BUFFER loads -- lane part of the address in one VGPR, the row in the scalar offset -- in batches of BS
// rows: prefetch() requests the first batch (and the bias) as early as the caller can afford the registers, i.e.
// before the input vector exists; run() requests batch b + 1 before it multiplies batch b.  No barrier inside: the
// caller publishes `out`.
template <int K, int O>
struct MV {
  static_assert((O & (O - 1)) == 0 && O >= 8 && O <= 128, "MV: O must be a power of two in [8,128]");
  static constexpr int OW = O / 4;                                   // outputs per wave
  static constexpr int VW = OW >= 16 ? 4 : (OW >= 8 ? 2 : 1);        // outputs per lane
  static constexpr int G = OW / VW;                                  // lane groups along the outputs
  static constexpr int PARTS = 64 / G;                               // k-partitions: the low lane bits
  static constexpr int ITER = (K + PARTS - 1) / PARTS;               // rows per lane
  static constexpr bool RAGGED = (K % PARTS) != 0;                   // the last partitions own rows >= K: clamped, input 0
  static constexpr int VR = ITER >= 4 ? 4 : (ITER >= 2 ? 2 : 1);     // floats per input read
  static constexpr int ITERP = (ITER + VR - 1) / VR * VR;            // slot group of a partition in the input vector
  static constexpr int KPAD = PARTS * ITERP;                         // floats of the stored input vector
  static constexpr int BS = 16;                                      // rows per batch (<= 64 registers)
  static constexpr int NB = (ITER + BS - 1) / BS;
  typedef float vec __attribute__((ext_vector_type(VW)));
  template <int B>
  static constexpr int bsize() { return (ITER - B * BS) < BS ? (ITER - B * BS) : BS; }
  struct Pre {
    vec w[bsize<0>()];
    vec bv;
  };
  // where element k of the input vector is stored
  static __device__ __forceinline__ constexpr int vidx(int k) { return (k / ITER) * ITERP + (k % ITER); }

  static __device__ __forceinline__ vec ld(WRsrc R, int voff, int soff_floats) {
    if constexpr (VW == 4) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(R, voff, soff_floats * 4, 0);
      return __builtin_bit_cast(vec, v);
    } else if constexpr (VW == 2) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(R, voff, soff_floats * 4, 0);
      return __builtin_bit_cast(vec, v);
    } else {
      const auto v = __builtin_amdgcn_raw_buffer_load_b32(R, voff, soff_floats * 4, 0);
      return __builtin_bit_cast(vec, v);
    }
  }
  // byte offset of this lane's first weight row; rows >= K (ragged K) are clamped to row K-1 -- their input is 0, but
  // the buffer holds bf16 planes after the fp32 part, whose bits may read as NaN
  template <int IT>
  static __device__ __forceinline__ int voff(int part, int o0) {
    int row = part * ITER;
    if constexpr (RAGGED) row = (row + IT < K ? row + IT : K - 1) - IT;
    return (row * O + o0) * 4;
  }
  template <int B>
  static __device__ __forceinline__ void load_batch(vec* w, WRsrc R, int wt_off, int part, int o0) {
    ffor<bsize<B>()>([&](auto ic) { constexpr int it = B * BS + decltype(ic)::v;
      w[decltype(ic)::v] = ld(R, voff<it>(part, o0), wt_off + it * O); });
  }
  template <bool BIAS = true>
  static __device__ __forceinline__ void prefetch(Pre& p, WRsrc R, int wt_off, int b_off, int tid_) {
    const int tid = opaque(tid_);
    const int lane = tid & 63, wave = tid >> 6;
    if constexpr (FE_WAVES > 4) { if (wave >= 4) return; }     // the per-jet matvecs are laid out over four waves
    const int part = lane & (PARTS - 1), o0 = wave * OW + (lane / PARTS) * VW;
    load_batch<0>(p.w, R, wt_off, part, o0);
    if constexpr (BIAS) p.bv = ld(R, o0 * 4, b_off);
    else ffor<VW>([&](auto ec) { p.bv[decltype(ec)::v] = 0.0f; });
  }
  // v: the input vector in LDS in the vidx layout (ITERP == ITER: the plain vector; rows past K are masked here)
  template <int ACT>
  static __device__ __forceinline__ void run(const Pre& p, WRsrc R, int wt_off, const float* v, float* out,
                                             const float* add, int tid_, float* pre = nullptr) {
    const int tid = opaque(tid_);
    const int lane = tid & 63, wave = tid >> 6;
    if constexpr (FE_WAVES > 4) { if (wave >= 4) return; }
    const int part = lane & (PARTS - 1), o0 = wave * OW + (lane / PARTS) * VW;
    vec acc;
    ffor<VW>([&](auto ec) { acc[decltype(ec)::v] = 0.0f; });
    vec wn[NB > 1 ? BS : 1];
    ffor<NB>([&](auto bc) {
      constexpr int b = decltype(bc)::v;
      constexpr int nrow = bsize<b>(), nslot = (nrow + VR - 1) / VR * VR;       // BS is a multiple of 4: aligned reads
      vec wc[nrow];
      ffor<nrow>([&](auto ic) { constexpr int i = decltype(ic)::v;
        if constexpr (b == 0) wc[i] = p.w[i]; else wc[i] = wn[i]; });
      if constexpr (b + 1 < NB) load_batch<b + 1>(wn, R, wt_off, part, o0);
      float vv[nslot];
      ffor<nslot / VR>([&](auto jc) { constexpr int j = decltype(jc)::v;
        const float* src = v + part * ITERP + b * BS + VR * j;
        if constexpr (VR == 4) {
          const float4 t = *reinterpret_cast<const float4*>(src);
          vv[4 * j] = t.x; vv[4 * j + 1] = t.y; vv[4 * j + 2] = t.z; vv[4 * j + 3] = t.w;
        } else if constexpr (VR == 2) {
          const float2 t = *reinterpret_cast<const float2*>(src);
          vv[2 * j] = t.x; vv[2 * j + 1] = t.y;
        } else {
          vv[j] = *src;
        } });
      // ragged K: the last partitions own rows >= K; their weights are row K-1's (clamped) and whatever sits behind the
      // input vector in LDS takes no part
      ffor<nrow>([&](auto ic) { constexpr int i = decltype(ic)::v;
        if constexpr (RAGGED) acc += wc[i] * (part * ITER + b * BS + i < K ? vv[i] : 0.0f);
        else acc += wc[i] * vv[i]; });
    });
    ffor<ilog2(PARTS)>([&](auto bc) {
      constexpr int bit = decltype(bc)::v;
      ffor<VW>([&](auto ec) { constexpr int e = decltype(ec)::v;
        if constexpr (bit == 0) acc[e] += dpp_mov<0xB1>(acc[e]);
        else if constexpr (bit == 1) acc[e] += dpp_mov<0x4E>(acc[e]);
        else if constexpr (bit == 2) acc[e] += dpp_mov<0x141>(acc[e]);      // row_half_mirror: quads are uniform by now
        else if constexpr (bit == 3) acc[e] += dpp_mov<0x140>(acc[e]);      // row_mirror: half rows are uniform by now
        else acc[e] += lane_xchg<bit>(acc[e]); });
    });
    if (part == 0) {
      ffor<VW>([&](auto ec) {
        constexpr int e = decltype(ec)::v;
        const float z = acc[e] + p.bv[e];
        if (pre) pre[o0 + e] = z;
        float t = act_f<ACT>(z);
        if (add) t += add[o0 + e];
        out[o0 + e] = t;
      });
    }
  }
};

// The input of fc_glob1 / mlp_global.0 in the layout of its consumer CMV (MV::vidx): element k = [sum/cnt, 0.01 sum]
// of the four wave partials (epic.py:100-107: ONE division per feature, by the thread that owns it -- round 2
// evaluated it per consumer lane, 4..16 IEEE divisions per lane and layer), then the `ntail` floats of `tail` (global
// vector / context), zeros in every other slot.  Caller: barrier before and after.
template <int H, class CMV>
__device__ __forceinline__ void pooled_combine(float* pooled, const float* partial, float cnt, const float* tail,
                                               int ntail, int tid_) {
  const int tid = opaque(tid_);
  ffor<(CMV::KPAD + FE_THREADS - 1) / FE_THREADS>([&](auto rc) {
    const int s = tid + decltype(rc)::v * FE_THREADS;
    const int part = s / CMV::ITERP, j = s % CMV::ITERP, k = part * CMV::ITER + j;
    float v = 0.0f;
    if (j < CMV::ITER && k < 2 * H) {
      const int f = k >= H ? k - H : k;
      float sum = (partial[f] + partial[128 + f]) + (partial[256 + f] + partial[384 + f]);
      if constexpr (FE_WAVES > 4) sum += partial[512 + f];
      v = k >= H ? sum * 0.01f : sum / cnt;
    } else if (j < CMV::ITER && k < 2 * H + ntail) {
      v = tail[k - 2 * H];
    }
    if (s < CMV::KPAD) pooled[s] = v;
  });
}

// acc[ot][r] = m * c[f] + d[f]  (c, d in LDS or global; either may be null)
template <int OT>
__device__ __forceinline__ void acc_init(f32x16 (&acc)[OT], float m, const float* c, const float* d, int half) {
  ffor<OT>([&](auto oc) {
    constexpr int ot = decltype(oc)::v;
    ffor<4>([&](auto qc) {
      constexpr int q = decltype(qc)::v;
      const int f = 32 * ot + 8 * q + 4 * half;
      float4 cv = c ? *reinterpret_cast<const float4*>(c + f) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 dv = d ? *reinterpret_cast<const float4*>(d + f) : make_float4(0.f, 0.f, 0.f, 0.f);
      acc[ot][4 * q + 0] = m * cv.x + dv.x;
      acc[ot][4 * q + 1] = m * cv.y + dv.y;
      acc[ot][4 * q + 2] = m * cv.z + dv.z;
      acc[ot][4 * q + 3] = m * cv.w + dv.w;
    });
  });
}

template <int OT, int ACT>
__device__ __forceinline__ void act_tiles(f32x16 (&a)[OT]) {
  ffor<OT>([&](auto oc) { constexpr int ot = decltype(oc)::v;
    ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; a[ot][r] = act_f<ACT>(a[ot][r]); }); });
}

// One butterfly stage of the transpose-reduce: the pair (in[2j], in[2j+1]) is split between the two
// lanes of an exchange pair -- the lane with bit STEP clear keeps the sum of in[2j], its partner
// the sum of in[2j+1] -- so the number of live values halves while the number of lanes summed doubles.
template <int N, int STEP>
__device__ __forceinline__ void bfly_stage(const float (&in)[N], float (&out)[N / 2], bool bit) {
  ffor<N / 2>([&](auto jc) {
    constexpr int j = decltype(jc)::v;
    const float keep = bit ? in[2 * j + 1] : in[2 * j];
    const float send = bit ? in[2 * j] : in[2 * j + 1];
    out[j] = keep + lane_xchg<STEP>(send);
  });
}

// Masked column sums of this wave's 32 particles for all 32*HT features -> partial[wave][feature].
// Transpose-reduce butterfly: 16*HT*(1/2+1/4+..) exchanges instead of 16*HT*5; afterwards lane l
// holds the flat register indices idx = 32 j + (l & 31)  (idx = 16 t + r of tile t, register r).
template <int HT>
__device__ __forceinline__ void pool_partials(const f32x16 (&x)[HT], float m, float* partial, int wave, int lane) {
  constexpr int N0 = 16 * HT;
  const int half = lane >> 5;
  float a0[N0];
  ffor<HT>([&](auto tc) { constexpr int t = decltype(tc)::v;
    ffor<16>([&](auto rc) { constexpr int r = decltype(rc)::v; a0[16 * t + r] = x[t][r] * m; }); });
  float a1[N0 / 2], a2[N0 / 4], a3[N0 / 8], a4[N0 / 16];
  bfly_stage<N0, 0>(a0, a1, (lane & 1) != 0);
  bfly_stage<N0 / 2, 1>(a1, a2, (lane & 2) != 0);
  bfly_stage<N0 / 4, 2>(a2, a3, (lane & 4) != 0);
  bfly_stage<N0 / 8, 3>(a3, a4, (lane & 8) != 0);
  float* dst = partial + wave * 128;
  if constexpr (HT == 1) {                         // one value left: plain sum over the last lane bit
    const float v = a4[0] + lane_xchg<4>(a4[0]);
    const int r = lane & 15;
    if ((lane & 16) == 0) dst[(r & 3) + 8 * (r >> 2) + 4 * half] = v;
  } else {
    float a5[N0 / 32];
    bfly_stage<N0 / 16, 4>(a4, a5, (lane & 16) != 0);
    ffor<N0 / 32>([&](auto jc) {
      constexpr int j = decltype(jc)::v;
      const int idx = 32 * j + (lane & 31);
      const int t = idx >> 4, r = idx & 15;
      dst[32 * t + (r & 3) + 8 * (r >> 2) + 4 * half] = a5[j];
    });
  }
}

// element k of  pooled = [sum/cnt, 0.01*sum, tail]  (epic.py:100-107) straight from the 4 wave partials
template <int H>
struct PooledIn {
  const float* partial;
  const float* tail;
  float cnt;
  __device__ __forceinline__ float operator()(int k) const {
    if (k >= 2 * H) return tail[k - 2 * H];
    const int f = k >= H ? k - H : k;
    const float s = (partial[f] + partial[128 + f]) + (partial[256 + f] + partial[384 + f]);
    return k >= H ? s * 0.01f : s / cnt;
  }
};

// time vector c = W_t temb of a layer: taken from the per-grid-point table when the caller
// precomputed it (generation: t is the same for every jet), else a matvec into L.cvec.
template <int ET, int O>
__device__ __forceinline__ const float* time_vec(const float* ctab, int& ci, const float* WtT, Lds& L, int tid) {
  if (ctab != nullptr) {
    const float* c = ctab + (int64_t)ci * 128;
    ++ci;
    return c;
  }
  matvec<MMB_ACT_NONE, ET, O>(WtT, nullptr, L.temb, L.cvec, nullptr, tid);
  return L.cvec;
}


// ------------------------------------------------------------------ model dimensions ----
template <int ET_, int EC_, int ED_, int HB_, int GB_, int DC_, int V_>
struct Dims {
  static constexpr int ET = ET_, EC = EC_, ED = ED_, HB = HB_, GB = GB_, DC = DC_, V = V_;
  static constexpr int HBT = HB / 32;          // tiles per branch
  static constexpr int HFT = 2 * HBT;          // tiles of the fused encoder
  static constexpr int GF = 2 * GB;
};

// The compiled width sets (variant id = position; mmb_hip/fused.py: VARIANTS / UNI_VARIANTS must list the same).
// Constraints: HB a multiple of 32 and <= 64 (the fused encoder's 2*HB/32 tiles x 2 live in registers), GB a power of
// two <= 32, ET a power of two <= 128, EC, ED <= 32.
typedef Dims<32, 16, 16, 64, 16, 3, 8> DimsS3;       // 0: configs/S3_multimodal.yaml
typedef Dims<16, 8, 8, 32, 8, 3, 8> DimsMid;         // 1: the mm_mid golden / small models
typedef Dims<64, 32, 32, 64, 32, 3, 8> DimsWide;     // 2: Et64 Ec32 Ed32 H64/64/128 G32/32/64
typedef Dims<32, 16, 16, 32, 16, 3, 8> DimsNarrow;   // 3: Et32 Ec16 Ed16 H32/32/64 G16/16/32
constexpr int N_MULTI_VARIANTS = 4;

// discrete-only model (UniModalEPiC): one encoder with projection
template <int ET_, int ED_, int H_, int G_, int V_>
struct UniDims {
  static constexpr int ET = ET_, ED = ED_, H = H_, G = G_, V = V_;
  static constexpr int HT = H / 32;
  static constexpr int EDT = (ED + 31) / 32;      // input tiles
  static constexpr int EDP = 32 * EDT;            // padded embedding width
};

typedef UniDims<128, 128, 128, 16, 8> UniS2;         // 0: configs/S2_discrete.yaml
typedef UniDims<16, 16, 32, 8, 8> UniMid;            // 1: the uni_mid golden
typedef UniDims<64, 64, 64, 16, 8> UniH64;           // 2: Et64 Ed64 H64 G16
typedef UniDims<32, 32, 32, 16, 8> UniH32;           // 3: Et32 Ed32 H32 G16
constexpr int N_UNI_VARIANTS = 4;

// Which of a jet's 32-particle tiles do per-particle work in the generation loops and the training kernels.  A padded slot never enters a pooled
// sum (the mask) and its own inputs are zero (embedder.py:74-80 + apply_mask): every padded slot of a jet carries the
// same row through the encoder.  The solvers move padded slots all the same (K3), so that row is still needed -- once:
// a tile with nothing but padded slots stays idle and takes drift and logits of every grid point from one
// representative padded slot of an active tile (rows of an MFMA product do not depend on their position: bit-identical).
// AOJ multiplicities (48 particles on average, 82 % of the jets below 65) leave two of four tiles idle on average.
// Training (need_rep = false): the loss is masked, so nothing of a padded slot is needed at all -- its activations
// are multiplied by gradients that are exactly zero.
struct TilePlan {
  bool active;       // wave-uniform: this wave's tile computes
  bool idle_any;     // workgroup-uniform: some tile with valid slots is idle, the representative publishes the heads
  bool rep;          // this lane holds the representative padded slot (lane half 0)
};
__device__ __forceinline__ TilePlan plan_tiles(bool on, bool need_rep, float m, bool valid, int wave, int lane, int* red) {
  const unsigned long long pad_b = __ballot(valid && m == 0.0f);
  const bool real = __ballot(m != 0.0f) != 0;
  __syncthreads();                                       // the readers of red's last use
  if (lane == 0) red[wave] = (real ? 1 : 0) | (pad_b != 0 ? 2 : 0);
  __syncthreads();
  int act = 0, pad = 0;
  for (int w = 0; w < FE_WAVES; ++w) { const int f = red[w]; act |= (f & 1) << w; pad |= ((f >> 1) & 1) << w; }
  int rep = -1;
  if (act & pad) rep = __ffs(act & pad) - 1;
  else if (pad && (need_rep || act == 0)) { rep = __ffs(pad) - 1; act |= 1 << rep; }   // no padded slot next to a real one: one more tile works
  if (!on) { act = (1 << FE_WAVES) - 1; rep = -1; }
  TilePlan t;
  t.active = __builtin_amdgcn_readfirstlane((act >> wave) & 1) != 0;
  t.idle_any = __builtin_amdgcn_readfirstlane(pad & ~act) != 0;
  t.rep = wave == rep && lane == (int)__ffsll((long long)pad_b) - 1;
  return t;
}

// Slot order of a jet inside the generation kernels: real particles first (stable), padded slots behind them, so that a
// short jet fills whole tiles and leaves whole tiles idle whatever slots its mask marks -- the reference's sources permute
// every mask (distributions/noise.py:36-37).  EPiC is permutation-equivariant: a particle's row does not depend on its
// slot, only the pooled sums move in their last bits with the order of summation (not at all for a mask that already is
// a prefix).  Injected noise and the Philox keys stay indexed by the true slot.  Returns the true slot behind `slot`.
// CH: 32-slot chunks of a jet per wave.  1 in general; 2 when the four-wave generation kernel takes a jet of up to 160
// slots whose real particles fit four tiles (chunk c of wave w = slots 32 (w + c FE_WAVES) ...): p[c] receives the true
// slot behind this lane's ordered slot of chunk c.
template <int CH, class OT>
__device__ __forceinline__ void jet_order_n(bool on, const float* maskf, int N, int wave, int lane, int* red, OT* order,
                                            int (&p)[CH]) {
  static_assert(FE_WAVES * CH <= 8, "jet_order_n: red[] holds eight chunk counts");
  int slot[CH];
  ffor<CH>([&](auto cc) { constexpr int c = decltype(cc)::v; slot[c] = 32 * (wave + c * FE_WAVES) + (lane & 31); p[c] = slot[c]; });
  if (!on) return;
  float m[CH];
  unsigned real_b[CH], pad_b[CH];
  ffor<CH>([&](auto cc) { constexpr int c = decltype(cc)::v;
    const bool valid = slot[c] < N;
    m[c] = valid ? maskf[slot[c]] : 0.0f;
    real_b[c] = (unsigned)__ballot(m[c] != 0.0f);                            // lanes 0..31 (32..63 repeat them)
    pad_b[c] = (unsigned)__ballot(valid && m[c] == 0.0f); });
  __syncthreads();                                       // the readers of red's last use
  if (lane == 0)
    ffor<CH>([&](auto cc) { constexpr int c = decltype(cc)::v;
      red[wave + c * FE_WAVES] = __popc(real_b[c]) | (__popc(pad_b[c]) << 8); });
  __syncthreads();
  int n_real = 0;
  for (int w = 0; w < FE_WAVES * CH; ++w) n_real += red[w] & 255;
  const unsigned below = (1u << (lane & 31)) - 1u;
  ffor<CH>([&](auto cc) { constexpr int c = decltype(cc)::v;
    int real_before = 0, pad_before = 0;
    for (int w = 0; w < wave + c * FE_WAVES; ++w) { const int f = red[w]; real_before += f & 255; pad_before += f >> 8; }
    const int pos = m[c] != 0.0f ? real_before + __popc(real_b[c] & below) : n_real + pad_before + __popc(pad_b[c] & below);
    if (slot[c] < N && lane < 32) order[pos] = (OT)slot[c]; });
  __syncthreads();
  ffor<CH>([&](auto cc) { constexpr int c = decltype(cc)::v; if (slot[c] < N) p[c] = order[slot[c]]; });
}
__device__ __forceinline__ int jet_order(bool on, const float* maskf, int N, int slot, int wave, int lane, int* red,
                                         short* order) {
  int p[1];
  jet_order_n<1, short>(on, maskf, N, wave, lane, red, order, p);
  return p[0];
}

// Two jets share a CU, one wave of each on every SIMD, and most jets leave tiles idle: which SIMDs the working tiles sit
// on decides whether the two jets' GEMMs meet on the same matrix pipes.  Tile t goes to SIMD t in the workgroup that
// holds the first wave slot and to SIMD 3 - t in the other one (measured with two working tiles per jet: 62 ms -> 56 ms
// for the S3 loop).  Returns the tile of this wave; the hardware wave order when the four waves do not sit on four SIMDs.
__device__ __forceinline__ int steer_wave(bool on, int hw_wave, int lane, int* red) {
  if (FE_WAVES != 4 || !on) return hw_wave;
  const unsigned id = __builtin_amdgcn_s_getreg((5 << 11) | 4);      // HW_REG_HW_ID[5:0]: wave slot [3:0], SIMD [5:4]
  if (lane == 0) red[hw_wave] = (int)id;
  __syncthreads();
  int seen = 0;
  for (int w = 0; w < 4; ++w) seen |= 1 << ((red[w] >> 4) & 3);
  const int flip = red[0] & 1, simd = (int)(id >> 4) & 3;
  const int wave = seen == 0xF ? (flip ? 3 - simd : simd) : hw_wave;
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(wave);
}

}  // namespace mmb
