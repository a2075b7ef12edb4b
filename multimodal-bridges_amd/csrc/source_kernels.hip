// SourceUniform on device (a7) + library-level plumbing (error string, version).
// Reference: distributions/noise.py:212-243.  One 64-lane wave per jet; the multiplicity is an
// inverse-CDF draw from the pmf and the "n random slots" of noise.py:221-224 (randperm) become
// "the n slots with the smallest Philox keys" -- the same uniform distribution over n-subsets.
#include "common.h"
#include <string.h>

namespace mmb {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

constexpr int SRC_MAXN = 1024;

__global__ void __launch_bounds__(64)
source_uniform_kernel(int N, int Dc, int V, int sample_masks, const float* __restrict__ pmf, int n_pmf,
                      uint64_t seed, uint64_t offset, float* __restrict__ x0, int64_t* __restrict__ k0,
                      int64_t* __restrict__ mask_i64, float* __restrict__ maskf, int32_t* __restrict__ counts) {
  __shared__ uint32_t keys[SRC_MAXN];
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  int n_on = N;
  if (sample_masks) {
    // multiplicity ~ Categorical(pmf): every lane walks the same cdf (n_pmf ~ 151, trivial)
    uint32_t r[4];
    philox4(seed, offset, (uint64_t)b, 0x5001u, r);
    float S = 0.0f;
    for (int i = 0; i < n_pmf; ++i) S += pmf[i];
    float thr = u01(r[0]) * S, acc = 0.0f;
    int mult = 0;
    for (int i = 0; i < n_pmf; ++i) { acc += pmf[i]; mult += (acc <= thr) ? 1 : 0; }
    if (mult > n_pmf - 1) mult = n_pmf - 1;
    n_on = mult < N ? mult : N;                     // mask[i,:n]=1 slicing clips at N (noise.py:222)
    for (int n = lane; n < N; n += 64) {
      uint32_t q[4];
      philox4(seed, offset, (uint64_t)b * N + n, 0x5002u, q);
      keys[n] = q[0];
    }
    __syncthreads();
  }
  for (int n = lane; n < N; n += 64) {
    const int64_t p = (int64_t)b * N + n;
    int on = 1;
    if (sample_masks) {
      // rank of this slot's key (ties broken by index) < n_on
      const uint32_t mine = keys[n];
      int rank = 0;
      for (int j = 0; j < N; ++j) {
        uint32_t o = keys[j];
        rank += (o < mine || (o == mine && j < n)) ? 1 : 0;
      }
      on = rank < n_on ? 1 : 0;
    }
    if (mask_i64) mask_i64[p] = on;
    if (maskf) maskf[p] = (float)on;
    uint32_t q[4];
    philox4(seed, offset, (uint64_t)p, 0x5003u, q);
    if (k0) k0[p] = (int64_t)(q[0] % (uint32_t)V);
    if (x0) {
      for (int d0 = 0; d0 < Dc; d0 += 4) {
        uint32_t g[4];
        philox4(seed, offset, (uint64_t)p, 0x5100u + (uint32_t)(d0 >> 2), g);
        float a, bq, cq, dq;
        box_muller(g[0], g[1], a, bq);
        box_muller(g[2], g[3], cq, dq);
        float vals[4] = {a, bq, cq, dq};
        for (int j = 0; j < 4 && d0 + j < Dc; ++j) x0[p * Dc + d0 + j] = vals[j];
      }
    }
  }
  if (counts && lane == 0) counts[b] = n_on;
}

// ---- jet-level observables of a (post-processed) sample: datamodules/utils.py:437-475 (JetFeatures),
// :21-42 (ParticleClouds kinematics), :558-574 (flavour counts), :548-556 (jet charge).  One wave per jet:
// lanes stride over the N slots, wavefront reduction, lane 0 finishes the jet four-vector.
// out (B, 19): numParticles, px, py, pz, e, pt, m, eta, phi, counts[0..7] (tokens of REAL particles),
// charge (kappa=0), jet charge (kappa=1).  As upstream, the kinematic sums run over ALL slots (a
// post-processed sample has zeros in the padded ones) and only the flavour counts look at the mask.
__global__ void __launch_bounds__(256)
jet_observables_kernel(int B, int N, int Dc, const float* __restrict__ x, const int64_t* __restrict__ k,
                       const float* __restrict__ maskf, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;         // n, px, py, pz, e, cnt[8], q, q*pt
  for (int n = lane; n < N; n += 64) {
    const int64_t p = (int64_t)b * N + n;
    const float m = maskf[p];
    acc[0] += m;
    float pt = 0.0f;
    if (x != nullptr) {
      pt = x[p * Dc];
      const float eta = x[p * Dc + 1], phi = x[p * Dc + 2];
      acc[1] += pt * cosf(phi);
      acc[2] += pt * sinf(phi);
      acc[3] += pt * sinhf(eta);
      acc[4] += pt * coshf(eta);
    }
    if (k != nullptr) {
      const int tok = (int)k[p];
      if (m > 0.0f && tok >= 0 && tok < 8) acc[5 + tok] += 1.0f;
      const float q = (tok >= 2 && tok < 8) ? ((tok & 1) ? 1.0f : -1.0f) : 0.0f;    // map_tokens_to_basis(...)[..., -1]
      acc[13] += q;
      acc[14] += q * pt;
    }
  }
#pragma unroll
  for (int i = 0; i < 15; ++i) acc[i] = wave_sum(acc[i]);
  if (lane == 0) {
    float* o = out + (int64_t)b * 19;
    const float px = acc[1], py = acc[2], pz = acc[3], e = acc[4];
    const float pt = sqrtf(px * px + py * py);
    o[0] = acc[0]; o[1] = px; o[2] = py; o[3] = pz; o[4] = e;
    o[5] = pt;
    o[6] = sqrtf(e * e - pt * pt - pz * pz);
    o[7] = 0.5f * logf((pt + pz) / (pt - pz));
    o[8] = atan2f(py, px);
#pragma unroll
    for (int v = 0; v < 8; ++v) o[9 + v] = acc[5 + v];
    o[17] = acc[13];
    o[18] = acc[14] / pt;
  }
}

}  // namespace mmb

using namespace mmb;

extern "C" const char* mmb_last_error(void) { return g_err; }
extern "C" int mmb_version(void) { return 100; }

extern "C" int mmb_source_uniform(int B, int N, int Dc, int V, int sample_masks, const float* pmf, int n_pmf,
                                  uint64_t seed, uint64_t offset, float* x0, int64_t* k0, int64_t* mask_i64,
                                  float* maskf, int32_t* counts, void* stream) {
  MMB_REQUIRE(B > 0 && N > 0, "mmb_source_uniform: empty shape B=%d N=%d", B, N);
  MMB_REQUIRE(N <= SRC_MAXN, "mmb_source_uniform: N=%d exceeds %d", N, SRC_MAXN);
  if (x0) MMB_REQUIRE(Dc > 0, "mmb_source_uniform: Dc must be > 0 with x0");
  if (k0) MMB_REQUIRE(V > 0, "mmb_source_uniform: V must be > 0 with k0");
  if (sample_masks) MMB_REQUIRE(pmf && n_pmf > 0, "mmb_source_uniform: sample_masks needs a pmf");
  hipLaunchKernelGGL(source_uniform_kernel, dim3(B), dim3(64), 0, as_stream(stream), N, Dc, V, sample_masks, pmf,
                     n_pmf, seed, offset, x0, k0, mask_i64, maskf, counts);
  return check_launch("mmb_source_uniform");
}

extern "C" int mmb_jet_observables(int B, int N, int Dc, const float* x, const int64_t* k, const float* maskf,
                                   float* out, void* stream) {
  MMB_REQUIRE(B > 0 && N > 0, "mmb_jet_observables: empty batch");
  MMB_REQUIRE(maskf && out && (x || k), "mmb_jet_observables: NULL argument");
  if (x) MMB_REQUIRE(Dc >= 3, "mmb_jet_observables: needs (pt, eta_rel, phi_rel), got Dc=%d", Dc);
  hipLaunchKernelGGL(jet_observables_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), B, N, Dc, x, k, maskf,
                     out);
  return check_launch("mmb_jet_observables");
}
