// Shared helpers for the gfx950 kernels of libmmb_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "../../include/mmb_hip.h"

namespace mmb {

// ---- host-side error plumbing ---------------------------------------------------------
void set_error(const char* fmt, ...);

#define MMB_REQUIRE(cond, ...)                       \
  do {                                               \
    if (!(cond)) {                                   \
      mmb::set_error(__VA_ARGS__);                   \
      return MMB_ERR_INVALID_ARG;                    \
    }                                                \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MMB_ERR_LAUNCH;
  }
  return MMB_OK;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is kept PER DEVICE by the runtime: a process that drives several GPUs
// must raise the limit on each of them.  One flag per (call site, device), set after the call succeeded; two threads
// racing on the first call both make the (idempotent) call.  Returns false when the runtime refuses.
constexpr int kMaxDevices = 64;
struct PerDeviceOnce {
  std::atomic<bool> done[kMaxDevices];
  template <typename F>
  bool run(F&& f) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return false;
    if (dev >= kMaxDevices) return f();                       // beyond the table: every call pays the (cheap) set
    if (done[dev].load(std::memory_order_acquire)) return true;
    if (!f()) return false;
    done[dev].store(true, std::memory_order_release);
    return true;
  }
};

// grid for a memory-bound grid-stride kernel: enough blocks to fill 256 CUs x 8, capped
inline int stream_grid(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (int)g;
}

// ---- Philox4x32-10 ---------------------------------------------------------------------
struct Philox {
  uint32_t c[4];
  uint32_t k[2];
};

__host__ __device__ inline void philox_round(uint32_t* c, const uint32_t* k) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint64_t p0 = (uint64_t)M0 * c[0];
  uint64_t p1 = (uint64_t)M1 * c[2];
  uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
  uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
  uint32_t n0 = hi1 ^ c[1] ^ k[0];
  uint32_t n1 = lo1;
  uint32_t n2 = hi0 ^ c[3] ^ k[1];
  uint32_t n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// 4 x 32 random bits for (seed, offset, index, lane-of-call).  ROUNDS = 10 is the standard
// Philox4x32-10; 7 is the smallest Crush-resistant variant (Salmon et al., SC'11) and is used
// where the 32-bit multiplies (quarter rate on CDNA) would make an HBM-bound kernel ALU-bound.
template <int ROUNDS = 10>
__host__ __device__ inline void philox4(uint64_t seed, uint64_t offset, uint64_t index, uint32_t sub,
                                        uint32_t out[4]) {
  uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), (uint32_t)offset,
                   (uint32_t)(offset >> 32) ^ (sub * 0x9E3779B9u)};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    philox_round(c, k);
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

// uniform in [0,1) with 24 bits, like torch.rand for fp32
__host__ __device__ inline float u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }
// uniform in (0,1]
__host__ __device__ inline float u01_open0(uint32_t r) { return ((float)(r >> 8) + 1.0f) * (1.0f / 16777216.0f); }

__device__ inline void box_muller(uint32_t r0, uint32_t r1, float& n0, float& n1) {
  float u = u01_open0(r0);
  float v = u01(r1);
  float rad = sqrtf(-2.0f * logf(u));
  float s, c;
  sincosf(6.28318530717958647692f * v, &s, &c);
  n0 = rad * c;
  n1 = rad * s;
}

// ---- thermostat (model/thermostats.py:21-52) ---------------------------------------------
// c = fp32(-V*gamma); w = exp(c * integral(t0,t1))
__device__ inline float thermo_w(int kind, float c, float t0, float t1) {
  float I;
  switch (kind) {
    case MMB_THERMO_CONSTANT: I = t1 - t0; break;
    case MMB_THERMO_INVERSE: I = logf(t1 / t0); break;
    case MMB_THERMO_LINEAR: I = (t1 * t1 - t0 * t0) / 2.0f; break;
    case MMB_THERMO_INVERSE_SQUARE: I = (t1 - t0) / (t1 * t0); break;
    default: I = tanhf(t1 / 2.0f) - tanhf(t0 / 2.0f); break;
  }
  return expf(c * I);
}

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace mmb
