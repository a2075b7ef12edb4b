// K-A: fused bridge sampling (training) -- HBM-bound, one lane per particle.
// Reference: model/multimodal_bridge_matching.py:194-225, model/bridges.py:22-37,87-108,124-213.
#include "common.h"

namespace mmb {

struct TelegraphW {
  float w_t1, w_0t, w_01, invV, ninvV;
};

__device__ inline TelegraphW telegraph_w(int kind, float c, float t, int V) {
  TelegraphW r;
  r.w_t1 = thermo_w(kind, c, t, 1.0f);
  r.w_0t = thermo_w(kind, c, 0.0f, t);
  r.w_01 = thermo_w(kind, c, 0.0f, 1.0f);
  r.invV = (float)(1.0 / (double)V);
  r.ninvV = (float)(-1.0 / (double)V);
  return r;
}

// p(v) = p(v->k1;t,1) * p(k0->v;0,t) / p(k0->k1;0,1), evaluated in the reference's order
// 1/V + w*((-1/V) + kron)  (bridges.py:191,212)
__device__ inline float telegraph_p(const TelegraphW& w, int v, int k0, int k1, float p3) {
  float p1 = w.invV + w.w_t1 * (w.ninvV + (v == k1 ? 1.0f : 0.0f));
  float p2 = w.invV + w.w_0t * (w.ninvV + (k0 == v ? 1.0f : 0.0f));
  return (p1 * p2) / p3;
}

__device__ inline int telegraph_draw(const TelegraphW& w, int V, int k0, int k1, float u) {
  float p3 = w.invV + w.w_01 * (w.ninvV + (k0 == k1 ? 1.0f : 0.0f));
  float S = 0.0f;
  for (int v = 0; v < V; ++v) S += telegraph_p(w, v, k0, k1, p3);
  float thr = u * S;
  float acc = 0.0f;
  int cnt = 0;
  for (int v = 0; v < V; ++v) {
    acc += telegraph_p(w, v, k0, k1, p3);
    cnt += (acc <= thr) ? 1 : 0;
  }
  return cnt < V ? cnt : V - 1;
}

__device__ inline float bridge_std(int bridge_kind, float sigma, float t) {
  return bridge_kind == MMB_BRIDGE_SCHRODINGER ? sigma * sqrtf(t * (1.0f - t)) : sigma;
}

__global__ void __launch_bounds__(256)
bridge_sample_kernel(int P, int N, int Dc, int V, const float* __restrict__ t_u, float eps,
                     const float* __restrict__ x0, const float* __restrict__ x1,
                     const float* __restrict__ z, const int64_t* __restrict__ k0,
                     const int64_t* __restrict__ k1, const float* __restrict__ u, int bridge_kind,
                     float sigma, int thermo_kind, float c, float* __restrict__ t_out,
                     float* __restrict__ xt, int64_t* __restrict__ kt) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    int b = p / N;
    float t = eps + (1.0f - eps) * t_u[b];
    if (p - b * N == 0) t_out[b] = t;
    if (x0 != nullptr) {
      float std = bridge_std(bridge_kind, sigma, t);
      float omt = 1.0f - t;
      const int64_t base = (int64_t)p * Dc;
      for (int d = 0; d < Dc; ++d) {
        float m = t * x1[base + d] + omt * x0[base + d];
        xt[base + d] = m + std * z[base + d];
      }
    }
    if (k0 != nullptr) {
      TelegraphW w = telegraph_w(thermo_kind, c, t, V);
      kt[p] = (int64_t)telegraph_draw(w, V, (int)k0[p], (int)k1[p], u[p]);
    }
  }
}

// Fused-RNG mode: t, source (x0,k0), z and u all come from Philox in registers.
// Draws are distribution-level (not bit-compared with anything), so the transcendental work uses
// the hardware approximations (v_exp/v_log/v_sin/v_cos): the kernel must stay HBM-bound.
__device__ inline void box_muller_fast(uint32_t r0, uint32_t r1, float& n0, float& n1) {
  const float u = u01_open0(r0);
  const float v = u01(r1);
  const float rad = sqrtf(-2.0f * __logf(u));
  n0 = rad * __builtin_amdgcn_cosf(v);            // v_cos_f32 takes revolutions: cos(2 pi v)
  n1 = rad * __builtin_amdgcn_sinf(v);
}

__device__ inline float thermo_w_fast(int kind, float c, float t0, float t1) {
  float I;
  switch (kind) {
    case MMB_THERMO_CONSTANT: I = t1 - t0; break;
    case MMB_THERMO_INVERSE: I = __logf(t1 / t0); break;
    case MMB_THERMO_LINEAR: I = (t1 * t1 - t0 * t0) * 0.5f; break;
    case MMB_THERMO_INVERSE_SQUARE: I = (t1 - t0) / (t1 * t0); break;
    default: I = tanhf(t1 * 0.5f) - tanhf(t0 * 0.5f); break;
  }
  return __expf(c * I);
}

// posterior draw without the common 1/p3 factor (the inverse CDF is scale invariant)
__device__ inline int telegraph_draw_fast(float w_t1, float w_0t, float invV, int V, int k0, int k1, float u) {
  // p(v) ~ (1/V + w_t1 (d(v,k1) - 1/V)) (1/V + w_0t (d(v,k0) - 1/V)): only 4 distinct values
  const float a0 = invV - w_t1 * invV, a1 = a0 + w_t1;      // factor 1 off/on
  const float b0 = invV - w_0t * invV, b1 = b0 + w_0t;      // factor 2 off/on
  const float base = a0 * b0;
  float S = base * (float)V;
  const float e1 = (k1 == k0) ? (a1 * b1 - base) : (a1 * b0 - base);
  const float e0 = (k1 == k0) ? 0.0f : (a0 * b1 - base);
  S += e1 + e0;
  const float thr = u * S;
  float acc = 0.0f;
  int cnt = 0;
  for (int v = 0; v < V; ++v) {
    acc += base + (v == k1 ? e1 : 0.0f) + (v == k0 ? e0 : 0.0f);
    cnt += (acc <= thr) ? 1 : 0;
  }
  return cnt < V ? cnt : V - 1;
}

// The same posterior drawn in O(1): it has only three classes -- v = k1, v = k0 (when k0 != k1) and the
// V-1 or V-2 "other" states of equal weight -- so one uniform picks the class and, inside "other",
// the state.  (The order in which the CDF visits the states is irrelevant for the law of k_t.)
struct TelegraphDrawCtx {
  float base, e_same, e_k1, e_k0, rbase;   // a0*b0 ; extra weights ; 1/base
  int V;
};
__device__ inline TelegraphDrawCtx telegraph_ctx(float w_t1, float w_0t, float invV, int V) {
  const float a0 = invV - w_t1 * invV, a1 = a0 + w_t1;
  const float b0 = invV - w_0t * invV, b1 = b0 + w_0t;
  const float base = a0 * b0;
  return TelegraphDrawCtx{base, a1 * b1 - base, a1 * b0 - base, a0 * b1 - base, __builtin_amdgcn_rcpf(base), V};
}
__device__ inline int telegraph_draw_o1(const TelegraphDrawCtx& c, int k0, int k1, float u) {
  const bool same = k0 == k1;
  const float wb = c.base + (same ? c.e_same : c.e_k1);          // weight of v = k1
  const float wa = same ? 0.0f : c.base + c.e_k0;                // weight of v = k0 (k0 != k1)
  const int n_other = c.V - (same ? 1 : 2);
  float thr = u * (wb + wa + c.base * (float)n_other);
  if (thr < wb) return k1;
  thr -= wb;
  if (thr < wa) return k0;
  thr -= wa;
  int idx = (int)(thr * c.rbase);
  idx = idx < n_other - 1 ? idx : n_other - 1;
  const int lo = k0 < k1 ? k0 : k1, hi = k0 < k1 ? k1 : k0;
  int v = idx;
  v += (v >= lo) ? 1 : 0;
  v += (!same && v >= hi) ? 1 : 0;
  return v;
}

// per-jet time draw t = eps + (1-eps) U  (multimodal_bridge_matching.py:202); its own tiny launch so
// that the per-particle kernel does not repeat a Philox call (32-bit multiplies are quarter rate)
__global__ void __launch_bounds__(256)
draw_time_kernel(int B, float eps, uint64_t seed, uint64_t offset, float* __restrict__ t_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) {
    uint32_t r[4];
    philox4(seed, offset, (uint64_t)b, 0x7001u, r);
    t_out[b] = eps + (1.0f - eps) * u01(r[0]);
  }
}

__global__ void __launch_bounds__(256)
bridge_sample_rng_kernel(int P, int N, int Dc, int V, float eps, const float* __restrict__ x1,
                         const int64_t* __restrict__ k1, int bridge_kind, float sigma,
                         int thermo_kind, float c, uint64_t seed, uint64_t offset,
                         float* __restrict__ t_out, float* __restrict__ x0_out,
                         int64_t* __restrict__ k0_out, float* __restrict__ xt,
                         int64_t* __restrict__ kt) {
  const float invV = 1.0f / (float)V;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    int b = p / N;
    uint32_t r[4];
    const float t = t_out[b];                                 // drawn by draw_time_kernel
    philox4(seed, offset, (uint64_t)p, 1u, r);                // per-particle stream, call 1
    float uk = u01(r[0]);
    int k0v = (int)(r[1] % (uint32_t)V);
    if (x1 != nullptr) {
      float std = bridge_std(bridge_kind, sigma, t);
      float omt = 1.0f - t;
      const int64_t base = (int64_t)p * Dc;
      // r[2], r[3] of call 1 feed the first Box-Muller pair; further pairs take their own call
      float a0, a1, z0, z1;
      box_muller_fast(r[2], r[3], a0, a1);                    // source x0[0], x0[1]
      uint32_t q[4];
      philox4(seed, offset, (uint64_t)p, 2u, q);
      box_muller_fast(q[0], q[1], z0, z1);                    // z[0], z[1]
      float s2 = 0.0f, z2 = 0.0f;
      if (Dc > 2) box_muller_fast(q[2], q[3], s2, z2);         // x0[2], z[2]
      x0_out[base] = a0;
      xt[base] = (t * x1[base] + omt * a0) + std * z0;
      if (Dc > 1) { x0_out[base + 1] = a1; xt[base + 1] = (t * x1[base + 1] + omt * a1) + std * z1; }
      if (Dc > 2) { x0_out[base + 2] = s2; xt[base + 2] = (t * x1[base + 2] + omt * s2) + std * z2; }
      for (int d0 = 3; d0 < Dc; d0 += 2) {
        uint32_t g[4];
        philox4(seed, offset, (uint64_t)p, 3u + (uint32_t)(d0 >> 1), g);
        float s0, s1, y0, y1;
        box_muller_fast(g[0], g[1], s0, s1);
        box_muller_fast(g[2], g[3], y0, y1);
        x0_out[base + d0] = s0;
        xt[base + d0] = (t * x1[base + d0] + omt * s0) + std * y0;
        if (d0 + 1 < Dc) { x0_out[base + d0 + 1] = s1; xt[base + d0 + 1] = (t * x1[base + d0 + 1] + omt * s1) + std * y1; }
      }
    }
    if (k1 != nullptr) {
      const float w_t1 = thermo_w_fast(thermo_kind, c, t, 1.0f);
      const float w_0t = thermo_w_fast(thermo_kind, c, 0.0f, t);
      if (k0_out != nullptr) k0_out[p] = k0v;
      kt[p] = (int64_t)telegraph_draw_fast(w_t1, w_0t, invV, V, k0v, (int)k1[p], uk);
    }
  }
}

// Dc == 3 fast form: one lane owns FOUR consecutive particles = 12 contiguous floats, so x1 is read
// and x_t / x0 are written as three dwordx4 per lane (a 12-byte particle stride with dword
// accesses leaves two thirds of every wave request unused), and k as two 16-byte accesses.
__global__ void __launch_bounds__(256)
bridge_sample_rng4_kernel(int P4, int N, int V, const float* __restrict__ x1, const int64_t* __restrict__ k1,
                          int bridge_kind, float sigma, int thermo_kind, float c, uint64_t seed, uint64_t offset,
                          const float* __restrict__ t_in, float* __restrict__ x0_out, int64_t* __restrict__ k0_out,
                          float* __restrict__ xt, int64_t* __restrict__ kt) {
  const float invV = 1.0f / (float)V;
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < P4; g += gridDim.x * blockDim.x) {
    const int p0 = g * 4;
    float xs[12], x0v[12], xtv[12];
    long long k1v[4], k0v[4], ktv[4];
    if (x1 != nullptr) {
      const float4* src = reinterpret_cast<const float4*>(x1 + (int64_t)p0 * 3);
      const float4 a = src[0], b = src[1], d = src[2];
      xs[0] = a.x; xs[1] = a.y; xs[2] = a.z; xs[3] = a.w; xs[4] = b.x; xs[5] = b.y; xs[6] = b.z; xs[7] = b.w;
      xs[8] = d.x; xs[9] = d.y; xs[10] = d.z; xs[11] = d.w;
    }
    if (k1 != nullptr) {
      const longlong2* ks = reinterpret_cast<const longlong2*>(k1 + p0);
      const longlong2 a = ks[0], b = ks[1];
      k1v[0] = a.x; k1v[1] = a.y; k1v[2] = b.x; k1v[3] = b.y;
    }
    // N % 4 == 0 (host-checked): the four particles of a lane belong to ONE jet -> per-jet terms once
    const float t = t_in[p0 / N];
    const float std = bridge_std(bridge_kind, sigma, t), omt = 1.0f - t;
    TelegraphDrawCtx tc;
    if (k1 != nullptr)
      tc = telegraph_ctx(thermo_w_fast(thermo_kind, c, t, 1.0f), thermo_w_fast(thermo_kind, c, 0.0f, t), invV, V);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = p0 + j;
      uint32_t r[4];
      philox4<7>(seed, offset, (uint64_t)p, 1u, r);
      const float uk = u01(r[0]);
      const int k0i = (int)__umulhi(r[1], (uint32_t)V);            // uniform on {0..V-1}
      if (x1 != nullptr) {
        float a0, a1, z0, z1, s2, z2;
        box_muller_fast(r[2], r[3], a0, a1);
        uint32_t q[4];
        philox4<7>(seed, offset, (uint64_t)p, 2u, q);
        box_muller_fast(q[0], q[1], z0, z1);
        box_muller_fast(q[2], q[3], s2, z2);
        x0v[3 * j] = a0; x0v[3 * j + 1] = a1; x0v[3 * j + 2] = s2;
        xtv[3 * j] = (t * xs[3 * j] + omt * a0) + std * z0;
        xtv[3 * j + 1] = (t * xs[3 * j + 1] + omt * a1) + std * z1;
        xtv[3 * j + 2] = (t * xs[3 * j + 2] + omt * s2) + std * z2;
      }
      if (k1 != nullptr) {
        k0v[j] = k0i;
        ktv[j] = telegraph_draw_o1(tc, k0i, (int)k1v[j], uk);
      }
    }
    if (x1 != nullptr) {
      float4* o0 = reinterpret_cast<float4*>(x0_out + (int64_t)p0 * 3);
      float4* o1 = reinterpret_cast<float4*>(xt + (int64_t)p0 * 3);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        o0[q] = make_float4(x0v[4 * q], x0v[4 * q + 1], x0v[4 * q + 2], x0v[4 * q + 3]);
        o1[q] = make_float4(xtv[4 * q], xtv[4 * q + 1], xtv[4 * q + 2], xtv[4 * q + 3]);
      }
    }
    if (k1 != nullptr) {
      longlong2* o = reinterpret_cast<longlong2*>(kt + p0);
      o[0] = make_longlong2(ktv[0], ktv[1]);
      o[1] = make_longlong2(ktv[2], ktv[3]);
      if (k0_out != nullptr) {
        longlong2* o2 = reinterpret_cast<longlong2*>(k0_out + p0);
        o2[0] = make_longlong2(k0v[0], k0v[1]);
        o2[1] = make_longlong2(k0v[2], k0v[3]);
      }
    }
  }
}

__global__ void __launch_bounds__(256)
telegraph_probs_kernel(int P, int N, int V, const float* __restrict__ t,
                       const int64_t* __restrict__ k0, const int64_t* __restrict__ k1,
                       int thermo_kind, float c, float* __restrict__ probs) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    TelegraphW w = telegraph_w(thermo_kind, c, t[p / N], V);
    int a = (int)k0[p], bq = (int)k1[p];
    float p3 = w.invV + w.w_01 * (w.ninvV + (a == bq ? 1.0f : 0.0f));
    for (int v = 0; v < V; ++v) probs[(int64_t)p * V + v] = telegraph_p(w, v, a, bq, p3);
  }
}

__device__ inline float drift_value(int bridge_kind, float t, float xt, float x0, float x1) {
  if (bridge_kind == MMB_BRIDGE_SCHRODINGER) {
    float den = t * (1.0f - t);
    float A = (1.0f - 2.0f * t) / den;
    float Bc = (t * t) / den;
    float C = -1.0f * ((1.0f - t) * (1.0f - t)) / den;
    return A * xt + Bc * x1 + C * x0;
  }
  return 0.0f * xt + 1.0f * x1 + (-1.0f) * x0;   // the literal 0*x_t term is kept (bridges.py:34-37)
}

__global__ void __launch_bounds__(256)
bridge_drift_kernel(int64_t n, int ND, const float* __restrict__ t, const float* __restrict__ xt,
                    const float* __restrict__ x0, const float* __restrict__ x1, int bridge_kind,
                    float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = drift_value(bridge_kind, t[i / ND], xt[i], x0[i], x1[i]);
}

}  // namespace mmb

using namespace mmb;

extern "C" int mmb_bridge_sample(int B, int N, int Dc, int V, const float* t_u, float eps,
                                 const float* x0, const float* x1, const float* z, const int64_t* k0,
                                 const int64_t* k1, const float* u, int bridge_kind, float sigma,
                                 int thermo_kind, float gamma, float* t_out, float* xt, int64_t* kt,
                                 void* stream) {
  MMB_REQUIRE(B > 0 && N > 0, "mmb_bridge_sample: empty shape B=%d N=%d", B, N);
  MMB_REQUIRE((int64_t)B * N < (1ll << 31), "mmb_bridge_sample: B*N too large");
  MMB_REQUIRE(t_u && t_out, "mmb_bridge_sample: t_u/t_out must not be NULL");
  MMB_REQUIRE(x0 || k0, "mmb_bridge_sample: no modality given");
  if (x0) MMB_REQUIRE(x1 && z && xt && Dc > 0, "mmb_bridge_sample: continuous mode needs x1,z,xt,Dc>0");
  if (k0) MMB_REQUIRE(k1 && u && kt && V > 1, "mmb_bridge_sample: discrete mode needs k1,u,kt,V>1");
  MMB_REQUIRE(thermo_kind >= 0 && thermo_kind <= 4, "mmb_bridge_sample: bad thermostat %d", thermo_kind);
  MMB_REQUIRE(bridge_kind == 0 || bridge_kind == 1, "mmb_bridge_sample: bad bridge %d", bridge_kind);
  int P = B * N;
  float c = (float)(-(double)V * (double)gamma);
  hipLaunchKernelGGL(bridge_sample_kernel, dim3(stream_grid(P, 256)), dim3(256), 0, as_stream(stream), P, N,
                     Dc, V, t_u, eps, x0, x1, z, k0, k1, u, bridge_kind, sigma, thermo_kind, c, t_out, xt, kt);
  return check_launch("mmb_bridge_sample");
}

extern "C" int mmb_bridge_sample_rng(int B, int N, int Dc, int V, float eps, const float* x1,
                                     const int64_t* k1, int bridge_kind, float sigma, int thermo_kind,
                                     float gamma, uint64_t seed, uint64_t offset, float* t_out,
                                     float* x0_out, int64_t* k0_out, float* xt, int64_t* kt,
                                     void* stream) {
  MMB_REQUIRE(B > 0 && N > 0, "mmb_bridge_sample_rng: empty shape");
  MMB_REQUIRE((int64_t)B * N < (1ll << 31), "mmb_bridge_sample_rng: B*N too large");
  MMB_REQUIRE(t_out, "mmb_bridge_sample_rng: t_out must not be NULL");
  MMB_REQUIRE(x1 || k1, "mmb_bridge_sample_rng: no modality given");
  if (x1) MMB_REQUIRE(x0_out && xt && Dc > 0, "mmb_bridge_sample_rng: continuous mode needs x0_out,xt");
  if (k1) MMB_REQUIRE(kt && V > 1, "mmb_bridge_sample_rng: discrete mode needs kt,V>1");
  MMB_REQUIRE(thermo_kind >= 0 && thermo_kind <= 4, "mmb_bridge_sample_rng: bad thermostat");
  MMB_REQUIRE(bridge_kind == 0 || bridge_kind == 1, "mmb_bridge_sample_rng: bad bridge");
  int P = B * N;
  float c = (float)(-(double)V * (double)gamma);
  hipLaunchKernelGGL(draw_time_kernel, dim3((B + 255) / 256), dim3(256), 0, as_stream(stream), B, eps, seed, offset,
                     t_out);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool fast4 = (P % 4 == 0) && (N % 4 == 0) && (!x1 || (Dc == 3 && al16(x1) && al16(x0_out) && al16(xt))) &&
                     (!k1 || (al16(k1) && al16(kt) && (!k0_out || al16(k0_out))));
  if (fast4)
    hipLaunchKernelGGL(bridge_sample_rng4_kernel, dim3(stream_grid(P / 4, 256)), dim3(256), 0, as_stream(stream),
                       P / 4, N, V, x1, k1, bridge_kind, sigma, thermo_kind, c, seed, offset, t_out, x0_out, k0_out,
                       xt, kt);
  else
    hipLaunchKernelGGL(bridge_sample_rng_kernel, dim3(stream_grid(P, 256)), dim3(256), 0, as_stream(stream), P,
                       N, Dc, V, eps, x1, k1, bridge_kind, sigma, thermo_kind, c, seed, offset, t_out, x0_out,
                       k0_out, xt, kt);
  return check_launch("mmb_bridge_sample_rng");
}

extern "C" int mmb_telegraph_probs(int B, int N, int V, const float* t, const int64_t* k0,
                                   const int64_t* k1, int thermo_kind, float gamma, float* probs,
                                   void* stream) {
  MMB_REQUIRE(B > 0 && N > 0 && V > 1, "mmb_telegraph_probs: bad shape");
  MMB_REQUIRE(t && k0 && k1 && probs, "mmb_telegraph_probs: NULL argument");
  MMB_REQUIRE(thermo_kind >= 0 && thermo_kind <= 4, "mmb_telegraph_probs: bad thermostat");
  int P = B * N;
  float c = (float)(-(double)V * (double)gamma);
  hipLaunchKernelGGL(telegraph_probs_kernel, dim3(stream_grid(P, 256)), dim3(256), 0, as_stream(stream), P, N,
                     V, t, k0, k1, thermo_kind, c, probs);
  return check_launch("mmb_telegraph_probs");
}

extern "C" int mmb_bridge_drift(int B, int N, int Dc, const float* t, const float* xt, const float* x0,
                                const float* x1, int bridge_kind, float* out, void* stream) {
  MMB_REQUIRE(B > 0 && N > 0 && Dc > 0, "mmb_bridge_drift: bad shape");
  MMB_REQUIRE(t && xt && x0 && x1 && out, "mmb_bridge_drift: NULL argument");
  int64_t n = (int64_t)B * N * Dc;
  hipLaunchKernelGGL(bridge_drift_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), n,
                     N * Dc, t, xt, x0, x1, bridge_kind, out);
  return check_launch("mmb_bridge_drift");
}
