// MFMA-only ceiling of this chip WITH a clock counter: how fast does a pure v_mfma_f32_32x32x16
// loop run, at which shader clock, and how much of that is operand-data dependent (DVFS)?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_clock.hip -o gpurun_out/mfma_clock && gpurun_out/mfma_clock
// Per variant: wall time (HIP events, all 256 CUs busy), s_memtime ticks (shader clock) and s_memrealtime ticks (100 MHz
// reference) over the timed loop of wave 0 of every block -> effective clock = s_memtime / s_memrealtime x 100 MHz, shader
// cycles per MFMA, TF/s.  Operand fills: zeros, one small constant, uniform [-1, 1), normal(0, 1) — f16 and bf16.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int BF, int CH>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ data, int iters, unsigned long long* out, float* sink) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[CH];
  for (int j = 0; j < CH; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // 4 A and 4 B operand registers per lane (different per lane and per wave), rotated through the chains
  uint4 a[4], b[4];
  for (int e = 0; e < 4; ++e) {
    a[e] = data[(threadIdx.x * 8 + e) & 4095];
    b[e] = data[(threadIdx.x * 8 + 4 + e) & 4095];
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (BF)
        acc[i % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i & 3]),
                                                              __builtin_bit_cast(bf16x8, b[(i >> 2) & 3]), acc[i % CH], 0, 0, 0);
      else
        acc[i % CH] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i & 3]),
                                                             __builtin_bit_cast(f16x8, b[(i >> 2) & 3]), acc[i % CH], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < CH; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];   // forces the last MFMAs to retire before the end stamp is meaningful
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
static unsigned short f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}
static float nrand() {
  const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
  return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

template <int BF, int CH>
static void run(const char* what, int fill, int waves_per_simd, int iters) {
  unsigned short* h = (unsigned short*)malloc(4096 * 16);
  for (int i = 0; i < 4096 * 8; ++i) {
    float v = 0.f;
    if (fill == 1) v = 0.5f;
    if (fill == 2) v = 2.f * rand() / (float)RAND_MAX - 1.f;
    if (fill == 3) v = nrand();
    h[i] = BF ? f2bf(v) : f2h(v);
  }
  uint4* d;
  unsigned long long* out;
  float* sink;
  hipMalloc(&d, 4096 * 16);
  hipMalloc(&out, 256 * 16);
  hipMalloc(&sink, 64);
  hipMemcpy(d, h, 4096 * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  unsigned long long ho[512];
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<BF, CH>), dim3(256), dim3(256 * waves_per_simd), 0, 0, d, iters, out, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) {
      best = ms;
      hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
    }
  }
  double st = 0, rt = 0;
  for (int i = 0; i < 256; ++i) {
    st += ho[2 * i];
    rt += ho[2 * i + 1];
  }
  st /= 256;
  rt /= 256;
  const double nm = (double)iters * 16;                       // MFMAs per wave
  const double flops = nm * 32768.0 * 4 * waves_per_simd * 256;
  printf("%-5s %-14s %d wave/SIMD %d chains: wall %8.1f us  %6.0f TF/s | s_memtime %9.0f ticks = %5.1f per MFMA-slot, "
         "s_memrealtime %7.0f x10ns -> clock %5.0f MHz\n",
         BF ? "bf16" : "f16", what, waves_per_simd, CH, best * 1e3, flops / (best * 1e-3) * 1e-12, st,
         st / (nm * waves_per_simd), rt, st / rt * 100.0);
  hipFree(d);
  hipFree(out);
  hipFree(sink);
  free(h);
}

// `mfma_clock sustain <fill 0..3> <seconds>`: keep the f16 2-wave / SIMD loop running for <seconds> and report TF/s and the
// effective clock of every half second (tools/power_probe.sh samples rocm-smi beside it)
static void sustain(int fill, double seconds) {
  unsigned short* h = (unsigned short*)malloc(4096 * 16);
  for (int i = 0; i < 4096 * 8; ++i) {
    float v = 0.f;
    if (fill == 1) v = 0.5f;
    if (fill == 2) v = 2.f * rand() / (float)RAND_MAX - 1.f;
    if (fill == 3) v = nrand();
    h[i] = f2h(v);
  }
  uint4* d;
  unsigned long long* out;
  float* sink;
  hipMalloc(&d, 4096 * 16);
  hipMalloc(&out, 256 * 16);
  hipMalloc(&sink, 64);
  hipMemcpy(d, h, 4096 * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000, per = 200;                          // 200 launches of ~2.5 ms per report
  double total = 0;
  unsigned long long ho[512];
  while (total < seconds) {
    hipEventRecord(e0, 0);
    for (int i = 0; i < per; ++i) hipLaunchKernelGGL((k<0, 8>), dim3(256), dim3(512), 0, 0, d, iters, out, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
    double st = 0, rt = 0;
    for (int i = 0; i < 256; ++i) {
      st += ho[2 * i];
      rt += ho[2 * i + 1];
    }
    total += ms * 1e-3;
    printf("sustain fill %d  t=%5.2f s  %6.0f TF/s  clock of the last launch %5.0f MHz\n", fill, total,
           (double)per * iters * 16 * 32768.0 * 4 * 2 * 256 / (ms * 1e-3) * 1e-12, st / rt * 100.0);
    fflush(stdout);
  }
  free(h);
}

int main(int argc, char** argv) {
  if (argc >= 4 && !strcmp(argv[1], "sustain")) {
    sustain(atoi(argv[2]), atof(argv[3]));
    return 0;
  }
  const int iters = 4000;
  const char* names[4] = {"zeros", "const 0.5", "uniform[-1,1)", "normal(0,1)"};
  for (int fill = 0; fill < 4; ++fill) {
    run<0, 8>(names[fill], fill, 1, iters);
    run<0, 8>(names[fill], fill, 2, iters);
    run<1, 8>(names[fill], fill, 1, iters);
    run<1, 8>(names[fill], fill, 2, iters);
  }
  run<0, 4>(names[3], 3, 1, iters);   // the 4-chain loop of tools/probes/mfma_valu_overlap.hip for comparison
  run<0, 4>(names[3], 3, 2, iters);
  return 0;
}
