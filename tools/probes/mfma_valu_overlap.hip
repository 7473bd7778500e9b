// Microbenchmark: do MFMA work and VALU / transcendental work overlap on one gfx950 SIMD — within one wave (interleaved
// instruction streams) and across the waves resident on it?  (The flash-attention tile loop, csrc/attention.hip, spends
// per 64-key tile and wave 16 MFMAs = 512 matrix cycles, 33 v_exp_f32 and ~115 other VALU instructions, and runs at
// ~1650 cycles per wave-tile of SIMD time = the SUM of the three, profiles/r2_attention_anatomy.txt.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o gpurun_out/mvo && gpurun_out/mvo
// Every variant runs `iters` rounds of a "tile": NM MFMAs (32x32x16 f16, 4 independent accumulator chains), NE v_exp_f32
// and NF v_fma_f32 (8 independent chains each).  Reported: SIMD cycles per tile-round (wall time x 2.4 GHz / rounds,
// divided by nothing: waves per SIMD are stated per line).
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { MODE_MFMA = 1, MODE_VALU = 2, MODE_BOTH_SEQ = 3, MODE_BOTH_INTERLEAVED = 4, MODE_SPLIT_WAVES = 5, MODE_EXP_ONLY = 6,
       MODE_FMA_ONLY = 7 };

template <int NM>
__device__ __forceinline__ void mfma_block(f32x16 (&acc)[4], f16x8 a, f16x8 b) {
#pragma unroll
  for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
}

template <int NE, int NF>
__device__ __forceinline__ void valu_block(float (&v)[8], float c) {
#pragma unroll
  for (int i = 0; i < NE; ++i) v[i & 7] = __builtin_amdgcn_exp2f(v[i & 7]);
#pragma unroll
  for (int i = 0; i < NF; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], c, 0.25f);
}

template <int MODE, int NM, int NE, int NF>
__global__ __launch_bounds__(1024) void k(int iters, float c, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(0.001f * (lane + e));
    b[e] = (_Float16)(0.002f * (lane - e));
  }
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = -0.01f * (lane + e);
  const bool do_m = MODE == MODE_MFMA || MODE == MODE_BOTH_SEQ || MODE == MODE_BOTH_INTERLEAVED ||
                    (MODE == MODE_SPLIT_WAVES && wave < nw / 2);
  const bool do_v = MODE == MODE_VALU || MODE == MODE_BOTH_SEQ || MODE == MODE_BOTH_INTERLEAVED ||
                    (MODE == MODE_SPLIT_WAVES && wave >= nw / 2);
  for (int it = 0; it < iters; ++it) {
    if (MODE == MODE_BOTH_INTERLEAVED) {
      // one MFMA, then its share of the VALU work, explicitly interleaved in program order
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < NE / NM; ++e) v[(i * (NE / NM) + e) & 7] = __builtin_amdgcn_exp2f(v[(i * (NE / NM) + e) & 7]);
#pragma unroll
        for (int e = 0; e < NF / NM; ++e) v[(i * (NF / NM) + e) & 7] = __builtin_fmaf(v[(i * (NF / NM) + e) & 7], c, 0.25f);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == MODE_EXP_ONLY) {
      valu_block<NE, 0>(v, c);
    } else if (MODE == MODE_FMA_ONLY) {
      valu_block<0, NF>(v, c);
    } else {
      if (do_m) mfma_block<NM>(acc, a, b);
      __builtin_amdgcn_sched_barrier(0);
      if (do_v) valu_block<NE, NF>(v, c);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int e = 0; e < 8; ++e) s += v[e];
  if (s == 123.456f) sink[0] = s;
}

// sweep: NM MFMAs per round, each followed by FE v_exp_f32 and FF v_fma_f32 (program order, sched_barrier between gaps)
template <int FE, int FF>
__global__ __launch_bounds__(1024) void ksweep(int iters, float c, float* sink) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(0.001f * (lane + e));
    b[e] = (_Float16)(0.002f * (lane - e));
  }
  float v[16];
  for (int e = 0; e < 16; ++e) v[e] = -0.01f * (lane + e);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < FE; ++e) v[(i * FE + e) & 15] = __builtin_amdgcn_exp2f(v[(i * FE + e) & 15]);
#pragma unroll
      for (int e = 0; e < FF; ++e) v[(i * FF + e + 5) & 15] = __builtin_fmaf(v[(i * FF + e + 5) & 15], c, 0.25f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int e = 0; e < 16; ++e) s += v[e];
  if (s == 123.456f) sink[0] = s;
}

template <int FE, int FF>
static void sweep(int iters, float* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double cyc[2];
  for (int w = 1; w <= 2; ++w) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL((ksweep<FE, FF>), dim3(256), dim3(256 * w), 0, 0, iters, 0.999f, sink);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    cyc[w - 1] = best * 1e-3 * 2.4e9 / iters / 16 / w;
  }
  printf("per MFMA gap: %d v_exp_f32 + %2d v_fma_f32   nominal cycles per MFMA and wave: %6.1f (1 wave/SIMD)  %6.1f (2 waves/SIMD)\n", FE, FF,
         cyc[0], cyc[1]);
}

template <int MODE, int NM, int NE, int NF>
static double run(int threads, int blocks_per_cu, int iters, float* sink, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, NM, NE, NF>), dim3(256 * blocks_per_cu), dim3(threads), 0, 0, iters, 0.999f, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double cyc = best * 1e-3 * 2.4e9 / iters;
  printf("%-58s waves/SIMD %d  %8.1f us  %7.0f cycles per round (at 2.4 GHz)\n", what, threads / 256 * blocks_per_cu, best * 1e3,
         cyc);
  return cyc;
}

int main() {
  float* sink;
  hipMalloc(&sink, 64);
  const int iters = 2000;
  constexpr int NM = 16, NE = 32, NF = 112;
  printf("tile-round = %d MFMA 32x32x16 + %d v_exp_f32 + %d v_fma_f32 per wave (the attention tile's mix)\n", NM, NE, NF);
  for (int wps = 1; wps <= 3; ++wps) {
    const int threads = 256 * wps;  // wps waves on each of the 4 SIMDs of a CU (one workgroup per CU)
    if (threads > 1024) break;
    run<MODE_MFMA, NM, NE, NF>(threads, 1, iters, sink, "MFMA only");
    run<MODE_EXP_ONLY, NM, NE, NF>(threads, 1, iters, sink, "v_exp only");
    run<MODE_FMA_ONLY, NM, NE, NF>(threads, 1, iters, sink, "v_fma only");
    run<MODE_VALU, NM, NE, NF>(threads, 1, iters, sink, "v_exp + v_fma");
    run<MODE_BOTH_SEQ, NM, NE, NF>(threads, 1, iters, sink, "MFMA block then VALU block, every wave");
    run<MODE_BOTH_INTERLEAVED, NM, NE, NF>(threads, 1, iters, sink, "MFMA / VALU interleaved per instruction, every wave");
    if (wps >= 2 && wps % 2 == 0)
      run<MODE_SPLIT_WAVES, NM, NE, NF>(threads, 1, iters, sink, "half the waves MFMA only, half VALU only (same SIMDs)");
  }
  // 4 waves per SIMD via 2 workgroups of 512
  run<MODE_BOTH_SEQ, NM, NE, NF>(512, 2, iters, sink, "MFMA block then VALU block, 2 workgroups x 512");
  run<MODE_SPLIT_WAVES, NM, NE, NF>(512, 2, iters, sink, "half MFMA / half VALU waves, 2 workgroups x 512");
  printf("\nfillers per MFMA gap (16 MFMAs per round, 4 accumulator chains)\n");
  sweep<0, 0>(iters, sink);
  sweep<0, 1>(iters, sink);
  sweep<0, 2>(iters, sink);
  sweep<0, 3>(iters, sink);
  sweep<0, 4>(iters, sink);
  sweep<0, 5>(iters, sink);
  sweep<0, 6>(iters, sink);
  sweep<0, 8>(iters, sink);
  sweep<0, 10>(iters, sink);
  sweep<0, 12>(iters, sink);
  sweep<0, 16>(iters, sink);
  sweep<1, 0>(iters, sink);
  sweep<2, 0>(iters, sink);
  sweep<3, 0>(iters, sink);
  sweep<4, 0>(iters, sink);
  sweep<2, 4>(iters, sink);
  sweep<2, 7>(iters, sink);
  return 0;
}
