// Does the ORDER of the MFMAs of a wave tile change the power-capped rate?  The matrix loops are energy-bound (DESIGN.md §3.0:
// 1332 W of the 1400 W package limit on random operands, 937 W on zeros), and an operand that stays on the pipe's input between
// two consecutive MFMAs does not toggle it.  This tool runs a 2 x 5 fragment wave tile (tile 80's: 2 A fragments, 5 B
// fragments, 10 accumulators) with nothing but v_mfma_f32_32x32x16_f16 in several issue orders, random normal f16 operands,
// all 256 CUs, for about half a second each, and reports the sustained TF/s and the clock:
//   0  m-outer          (a0: b0..b4) (a1: b0..b4)                 A changes 2x, B 10x per round
//   1  n-outer          (b0: a0 a1) (b1: a0 a1) ...               A changes 10x, B 5x
//   2  m-outer snake    (a0: b0..b4) (a1: b4..b0)                 A 2x, B 9x
//   3  n-outer snake    a0b0 a1b0 a1b1 a0b1 a0b2 a1b2 ...         exactly ONE operand changes per MFMA (10 changes)
//   4  all different    10 A and 10 B registers                   20 changes
//   5  all the same     a0 b0 ten times                           0 changes (random data, nothing toggles on the inputs)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_order.hip -o gpurun_out/mfma_order && gpurun_out/mfma_order
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MM(AI, BI, CI)                                                                               \
  acc[CI] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[AI], b[BI], acc[CI], 0, 0, 0);                  \
  __builtin_amdgcn_sched_barrier(0);

template <int PAT>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ data, int iters, unsigned long long* out, float* sink) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[10];
  for (int j = 0; j < 10; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f16x8 a[10], b[10];
  for (int e = 0; e < 10; ++e) {
    a[e] = __builtin_bit_cast(f16x8, data[(threadIdx.x * 20 + e) & 4095]);
    b[e] = __builtin_bit_cast(f16x8, data[(threadIdx.x * 20 + 10 + e) & 4095]);
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (PAT == 0) {
      MM(0, 0, 0) MM(0, 1, 1) MM(0, 2, 2) MM(0, 3, 3) MM(0, 4, 4) MM(1, 0, 5) MM(1, 1, 6) MM(1, 2, 7) MM(1, 3, 8) MM(1, 4, 9)
    } else if (PAT == 1) {
      MM(0, 0, 0) MM(1, 0, 5) MM(0, 1, 1) MM(1, 1, 6) MM(0, 2, 2) MM(1, 2, 7) MM(0, 3, 3) MM(1, 3, 8) MM(0, 4, 4) MM(1, 4, 9)
    } else if (PAT == 2) {
      MM(0, 0, 0) MM(0, 1, 1) MM(0, 2, 2) MM(0, 3, 3) MM(0, 4, 4) MM(1, 4, 9) MM(1, 3, 8) MM(1, 2, 7) MM(1, 1, 6) MM(1, 0, 5)
    } else if (PAT == 3) {
      MM(0, 0, 0) MM(1, 0, 5) MM(1, 1, 6) MM(0, 1, 1) MM(0, 2, 2) MM(1, 2, 7) MM(1, 3, 8) MM(0, 3, 3) MM(0, 4, 4) MM(1, 4, 9)
    } else if (PAT == 4) {
      MM(0, 0, 0) MM(1, 1, 1) MM(2, 2, 2) MM(3, 3, 3) MM(4, 4, 4) MM(5, 5, 5) MM(6, 6, 6) MM(7, 7, 7) MM(8, 8, 8) MM(9, 9, 9)
    } else {
      MM(0, 0, 0) MM(0, 0, 1) MM(0, 0, 2) MM(0, 0, 3) MM(0, 0, 4) MM(0, 0, 5) MM(0, 0, 6) MM(0, 0, 7) MM(0, 0, 8) MM(0, 0, 9)
    }
  }
  float s = 0.f;
  for (int j = 0; j < 10; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 123.456f) sink[0] = s;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && (threadIdx.x >> 6) == 0) {
    out[blockIdx.x * 2] = t1 - t0;
    out[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

static unsigned short f2h(float f) {
  _Float16 h = (_Float16)f;
  unsigned short u;
  memcpy(&u, &h, 2);
  return u;
}
static float nrand() {
  const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
  return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

template <int PAT>
static void run(const char* what, int waves_per_simd, const uint4* d, unsigned long long* out, float* sink, double seconds) {
  const int iters = 6000, per = 50;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double total = 0, flops = 0;
  unsigned long long ho[512];
  float last_ms = 0;
  while (total < seconds) {
    hipEventRecord(e0, 0);
    for (int i = 0; i < per; ++i) hipLaunchKernelGGL((k<PAT>), dim3(256), dim3(256 * waves_per_simd), 0, 0, d, iters, out, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&last_ms, e0, e1);
    total += last_ms * 1e-3;
    flops += (double)per * iters * 10 * 32768.0 * 4 * waves_per_simd * 256;
  }
  hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
  double st = 0, rt = 0;
  for (int i = 0; i < 256; ++i) {
    st += ho[2 * i];
    rt += ho[2 * i + 1];
  }
  printf("pattern %d %-16s %d wave/SIMD: %6.0f TF/s over %.2f s (last batch %6.0f TF/s), clock %5.0f MHz, %5.1f cycles per MFMA-slot\n",
         PAT, what, waves_per_simd, flops / total * 1e-12, total,
         (double)per * iters * 10 * 32768.0 * 4 * waves_per_simd * 256 / (last_ms * 1e-3) * 1e-12, st / rt * 100.0,
         st / 256 / ((double)iters * 10 * waves_per_simd));
  fflush(stdout);
}

int main() {
  unsigned short* h = (unsigned short*)malloc(4096 * 16);
  for (int i = 0; i < 4096 * 8; ++i) h[i] = f2h(nrand());
  uint4* d;
  unsigned long long* out;
  float* sink;
  hipMalloc(&d, 4096 * 16);
  hipMalloc(&out, 256 * 16);
  hipMalloc(&sink, 64);
  hipMemcpy(d, h, 4096 * 16, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep)
    for (int w = 1; w <= 2; ++w) {
      run<0>("m-outer", w, d, out, sink, 0.4);
      run<1>("n-outer", w, d, out, sink, 0.4);
      run<2>("m-outer snake", w, d, out, sink, 0.4);
      run<3>("n-outer snake", w, d, out, sink, 0.4);
      run<4>("all different", w, d, out, sink, 0.4);
      run<5>("all the same", w, d, out, sink, 0.4);
    }
  return 0;
}
