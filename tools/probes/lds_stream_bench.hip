// Microbenchmark: how fast can every CU stream a SHARED weight sequence L2 -> LDS with `buffer_load ... lds`?
// (the fetch-path ceiling of the fused transformer kernels, csrc/xformer.hip: one 512-thread workgroup per CU copies
// 20.5 KB tiles into a 3-slot LDS ring with counted vmcnt + one barrier per tile, optionally issuing the MFMAs of a
// real tile (10 per wave) between barriers.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_stream_bench.hip -o gpurun_out/lsb && gpurun_out/lsb
// Prints GB/s per CU and in aggregate for stream sizes that sit in L2 (3.4 MB), in the Infinity Cache (13 / 54 MB) and
// in HBM (860 MB), with all workgroups reading the same tile at (nearly) the same time or each starting at its own offset.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int TILE = 20992, NSLOT = 3;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int MFMAS>
__global__ __launch_bounds__(512) void stream_kernel(const char* w, int ntiles, int iters, int skew, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, ntiles * TILE, 0x00020000);
  int s_t = skew ? (int)(((long long)blockIdx.x * 977) % ntiles) : 0, s_slot = 0, c_slot = 0, issued = 0;
  const int voff = tid * 16, lane16 = lane * 16;
#define STAGE()                                                                                                    \
  do {                                                                                                             \
    char* dst = smem + s_slot * TILE;                                                                              \
    const int so = s_t * TILE;                                                                                     \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lptr_t)(dst + wave * 1024), 16, voff, so, 0, 0);                \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lptr_t)(dst + 8192 + wave * 1024), 16, voff, so + 8192, 0, 0);  \
    if (wave < 4) {                                                                                                \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lptr_t)(dst + 16384 + wave * 1024), 16, voff, so + 16384, 0, 0); \
    } else if (wave == 4) {                                                                                        \
      if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lptr_t)(dst + 20480), 16, lane16, so + 20480, 0, 0); \
    }                                                                                                              \
    ++issued;                                                                                                      \
    s_slot = s_slot + 1 == NSLOT ? 0 : s_slot + 1;                                                                 \
    s_t = s_t + 1 == ntiles ? 0 : s_t + 1;                                                                         \
  } while (0)
  STAGE();
  STAGE();
  f32x16 acc[5];
  for (int j = 0; j < 5; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const int after = issued - it - 1;
    if (after >= 2) {
      if (wave <= 4) wait_vm<6>(); else wait_vm<4>();
    } else if (after == 1) {
      if (wave <= 4) wait_vm<3>(); else wait_vm<2>();
    } else {
      wait_vm<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const char* base = smem + c_slot * TILE;
    c_slot = c_slot + 1 == NSLOT ? 0 : c_slot + 1;
    if (MFMAS) {
      f16x8 xf = *reinterpret_cast<const f16x8*>(base + lane16);
      f16x8 wf[5];
      for (int j = 0; j < 5; ++j) wf[j] = *reinterpret_cast<const f16x8*>(base + (1 + j) * 1024 + lane16);
      __builtin_amdgcn_sched_barrier(0);
      if (issued < iters) STAGE();
#pragma unroll
      for (int ks = 0; ks < MFMAS / 5; ++ks)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf, acc[j], 0, 0, 0);
    } else {
      if (issued < iters) STAGE();
    }
  }
  float s = 0.f;
  for (int j = 0; j < 5; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 123.456f) sink[0] = s;
}

// Producer / consumer split: waves 0-7 only read fragments and issue MFMAs, NLOAD extra waves only issue the direct-to-LDS
// copies of the weight tiles (and wait for them): does the copy then run UNDER the MFMAs instead of in front of them?
// (one workgroup barrier per tile as above; loader l copies pieces l, l + NLOAD, ... of a tile, loader 0 also the side data)
template <int MFMAS, int NLOAD>
__global__ __launch_bounds__(64 * (8 + NLOAD)) void stream_spec_kernel(const char* w, int ntiles, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, ntiles * TILE, 0x00020000);
  const int lane16 = lane * 16;
  constexpr int PPL = 20 / NLOAD;  // full pieces per loader and tile
  f32x16 acc[5];
  for (int j = 0; j < 5; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  if (wave >= 8) {
    const int l = wave - 8;
    int s_t = 0, s_slot = 0;
#define LSTAGE()                                                                                                     \
  do {                                                                                                               \
    char* dst = smem + s_slot * TILE;                                                                                \
    const int so = s_t * TILE;                                                                                       \
    _Pragma("unroll") for (int q = 0; q < PPL; ++q)                                                                  \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lptr_t)(dst + (l + q * NLOAD) * 1024), 16, lane16,            \
                                                 so + (l + q * NLOAD) * 1024, 0, 0);                                 \
    if (l == 0) {                                                                                                    \
      if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (lptr_t)(dst + 20480), 16, lane16, so + 20480, 0, 0); \
    }                                                                                                                \
    s_slot = s_slot + 1 == NSLOT ? 0 : s_slot + 1;                                                                   \
    s_t = s_t + 1 == ntiles ? 0 : s_t + 1;                                                                           \
  } while (0)
    LSTAGE();
    LSTAGE();
    for (int it = 0; it < iters; ++it) {
      // tile `it` landed: at most the loads of tile it + 1 may stay in flight
      if (it + 1 < iters + 1) {
        if (l == 0) wait_vm<PPL + 1>(); else wait_vm<PPL>();
      }
      asm volatile("s_barrier" ::: "memory");
      if (it + 2 < iters + 2) LSTAGE();  // into the slot the consumers finished before this barrier
    }
    wait_vm<0>();
#undef LSTAGE
    return;
  }
  int c_slot = 0;
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const char* base = smem + c_slot * TILE;
    c_slot = c_slot + 1 == NSLOT ? 0 : c_slot + 1;
    if (MFMAS) {
      f16x8 xf = *reinterpret_cast<const f16x8*>(base + lane16);
      f16x8 wf[5];
      for (int j = 0; j < 5; ++j) wf[j] = *reinterpret_cast<const f16x8*>(base + (1 + j + (wave & 3)) * 1024 + lane16);
#pragma unroll
      for (int ks = 0; ks < MFMAS / 5; ++ks)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf, acc[j], 0, 0, 0);
    }
  }
  float sres = 0.f;
  for (int j = 0; j < 5; ++j)
    for (int r = 0; r < 16; ++r) sres += acc[j][r];
  if (sres == 123.456f) sink[0] = sres;
}

template <int MFMAS, int NLOAD>
static void run_spec(const char* w, int ntiles, int iters, float* sink, const char* what) {
  auto kern = &stream_spec_kernel<MFMAS, NLOAD>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * TILE);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * (8 + NLOAD)), NSLOT * TILE, 0, w, ntiles, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = 256.0 * iters * TILE;
  printf("%-34s stream %7.1f MB  mfma/tile/wave %2d  %8.1f us  %7.1f GB/s per CU  %6.2f TB/s aggregate  (%.2f us per tile)\n", what,
         ntiles * (double)TILE / 1e6, MFMAS, best * 1e3, bytes / 256 / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12,
         best * 1e3 / iters);
}

template <int MFMAS>
static void run(const char* w, int ntiles, int iters, int skew, float* sink, const char* what) {
  auto kern = &stream_kernel<MFMAS>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * TILE);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), NSLOT * TILE, 0, w, ntiles, iters, skew, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = 256.0 * iters * TILE;
  printf("%-34s stream %7.1f MB  mfma/tile/wave %2d  %8.1f us  %7.1f GB/s per CU  %6.2f TB/s aggregate  (%.2f us per tile)\n", what,
         ntiles * (double)TILE / 1e6, MFMAS, best * 1e3, bytes / 256 / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12,
         best * 1e3 / iters);
}

int main() {
  const int maxtiles = 41000;
  char* w;
  float* sink;
  hipMalloc(&w, (size_t)maxtiles * TILE);
  hipMalloc(&sink, 64);
  hipMemset(w, 0x11, (size_t)maxtiles * TILE);
  const int iters = 1600;
  for (int ntiles : {160, 640, 2560, 41000}) {
    run<0>(w, ntiles, iters, 0, sink, "copy only, same tile everywhere");
    run<0>(w, ntiles, iters, 1, sink, "copy only, skewed start");
    run<10>(w, ntiles, iters, 0, sink, "copy + MFMAs, same tile");
    run<10>(w, ntiles, iters, 1, sink, "copy + MFMAs, skewed start");
    if (ntiles <= 640) {
      run<5>(w, ntiles, iters, 0, sink, "copy + 5 MFMAs, same tile");
      run_spec<10, 2>(w, ntiles, iters, sink, "8 MFMA waves + 2 loader waves");
      run_spec<10, 4>(w, ntiles, iters, sink, "8 MFMA waves + 4 loader waves");
      run_spec<5, 2>(w, ntiles, iters, sink, "8 MFMA waves (5) + 2 loaders");
      run_spec<5, 4>(w, ntiles, iters, sink, "8 MFMA waves (5) + 4 loaders");
      run_spec<0, 2>(w, ntiles, iters, sink, "no MFMAs, 2 loader waves");
      run_spec<0, 4>(w, ntiles, iters, sink, "no MFMAs, 4 loader waves");
    }
  }
  return 0;
}
