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
