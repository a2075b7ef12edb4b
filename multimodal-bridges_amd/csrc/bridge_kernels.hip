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
__global__ void __launch_bounds__(256)
bridge_sample_rng_kernel(int P, int N, int Dc, int V, float eps, const float* __restrict__ x1,
                         const int64_t* __restrict__ k1, int bridge_kind, float sigma,
                         int thermo_kind, float c, uint64_t seed, uint64_t offset,
                         float* __restrict__ t_out, float* __restrict__ x0_out,
                         int64_t* __restrict__ k0_out, float* __restrict__ xt,
                         int64_t* __restrict__ kt) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    int b = p / N;
    uint32_t r[4];
    philox4(seed, offset, (uint64_t)b, 0x7001u, r);           // per-jet stream
    float t = eps + (1.0f - eps) * u01(r[0]);
    if (p - b * N == 0) t_out[b] = t;
    philox4(seed, offset, (uint64_t)p, 1u, r);                // per-particle stream, call 1
    float uk = u01(r[0]);
    int k0v = (int)(r[1] % (uint32_t)V);
    if (x1 != nullptr) {
      float std = bridge_std(bridge_kind, sigma, t);
      float omt = 1.0f - t;
      const int64_t base = (int64_t)p * Dc;
      for (int d0 = 0; d0 < Dc; d0 += 2) {
        uint32_t q[4];
        philox4(seed, offset, (uint64_t)p, 2u + (uint32_t)(d0 >> 1), q);
        float a0, a1, z0, z1;
        box_muller(q[0], q[1], a0, a1);                        // source x0
        box_muller(q[2], q[3], z0, z1);                        // regulariser z
        float s0 = a0;
        x0_out[base + d0] = s0;
        xt[base + d0] = (t * x1[base + d0] + omt * s0) + std * z0;
        if (d0 + 1 < Dc) {
          float s1 = a1;
          x0_out[base + d0 + 1] = s1;
          xt[base + d0 + 1] = (t * x1[base + d0 + 1] + omt * s1) + std * z1;
        }
      }
    }
    if (k1 != nullptr) {
      TelegraphW w = telegraph_w(thermo_kind, c, t, V);
      if (k0_out != nullptr) k0_out[p] = k0v;
      kt[p] = (int64_t)telegraph_draw(w, V, k0v, (int)k1[p], uk);
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
