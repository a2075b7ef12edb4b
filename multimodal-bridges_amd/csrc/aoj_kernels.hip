// AspenOpenJets pre-processing on device.  Reference: datamodules/aoj.py:147-245 --
// _filter_particles (:179-184), _compute_continuous_coordinates (:202-245), _map_pid_to_tokens (:186-200).
// Input: the raw PFCands block of an AOJ file, (J, M, F) f32 with columns px, py, pz, E, d0, d0Err, dz, dzErr, ...,
// pid at column F-2.  One 64-lane wave per jet: lanes stride over the M candidates, a wavefront reduction gives the
// jet four-vector (the axis the relative coordinates refer to), then every lane writes its candidates' features.
#include "common.h"

namespace mmb {

__device__ __forceinline__ int pid_token(int pid) {
  switch (pid) {                      // aoj.py:189-198; anything else -> 0 (pid_map.get(x, 0))
    case 22: return 0;
    case 130: return 1;
    case -211: return 2;
    case 211: return 3;
    case -11: return 4;
    case 11: return 5;
    case -13: return 6;
    case 13: return 7;
    default: return 0;
  }
}

__global__ void __launch_bounds__(256)
aoj_features_kernel(int J, int M, int F, const float* __restrict__ pf, float* __restrict__ feats,
                    int64_t* __restrict__ tokens, int64_t* __restrict__ mask) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= J) return;
  const float* base = pf + (int64_t)j * M * F;
  // jet axis: sum of the four-vectors of the candidates that survive the pid filter (|pid| < 11 rows are zeroed)
  float sx = 0.0f, sy = 0.0f, sz = 0.0f;
  for (int n = lane; n < M; n += 64) {
    const float* c = base + (int64_t)n * F;
    const bool bad = fabsf(c[F - 2]) < 11.0f;
    sx += bad ? 0.0f : c[0];
    sy += bad ? 0.0f : c[1];
    sz += bad ? 0.0f : c[2];
  }
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
  const float jet_eta = asinhf(sz / sqrtf(sx * sx + sy * sy));
  const float jet_phi = atan2f(sy, sx);
  const float PI = 3.14159265358979323846f, TWO_PI = 6.28318530717958647692f;
  for (int n = lane; n < M; n += 64) {
    const float* c = base + (int64_t)n * F;
    const bool bad = fabsf(c[F - 2]) < 11.0f;
    const float px = bad ? 0.0f : c[0], py = bad ? 0.0f : c[1], pz = bad ? 0.0f : c[2], e = bad ? 0.0f : c[3];
    const int pid = bad ? 0 : (int)c[F - 2];
    const float pt = sqrtf(px * px + py * py);
    const float eta = asinhf(pt != 0.0f ? pz / pt : 0.0f);
    const float phi = atan2f(py, px);
    float dphi = fmodf(phi - jet_phi + PI, TWO_PI);           // numpy's % : result carries the sign of the divisor
    if (dphi < 0.0f) dphi += TWO_PI;
    const float m = e > 0.0f ? 1.0f : 0.0f;
    const int64_t p = (int64_t)j * M + n;
    feats[p * 3 + 0] = pt * m;
    feats[p * 3 + 1] = (eta - jet_eta) * m;
    feats[p * 3 + 2] = (dphi - PI) * m;
    tokens[p] = pid_token(pid);
    mask[p] = e > 0.0f ? 1 : 0;
  }
}

}  // namespace mmb

using namespace mmb;

extern "C" int mmb_aoj_features(int J, int M, int F, const float* pfcands, float* feats, int64_t* tokens, int64_t* mask,
                                void* stream) {
  MMB_REQUIRE(J > 0 && M > 0, "mmb_aoj_features: empty block J=%d M=%d", J, M);
  MMB_REQUIRE(F >= 6, "mmb_aoj_features: PFCands rows have %d columns, need px,py,pz,E,...,pid,x", F);
  MMB_REQUIRE(pfcands && feats && tokens && mask, "mmb_aoj_features: NULL argument");
  hipLaunchKernelGGL(aoj_features_kernel, dim3((J + 3) / 4), dim3(256), 0, as_stream(stream), J, M, F, pfcands, feats,
                     tokens, mask);
  return check_launch("mmb_aoj_features");
}
