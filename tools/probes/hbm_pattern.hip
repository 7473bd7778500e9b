// Diagnostic (not part of the product): how fast can 256 CUs move a [65536 x 320] f16 activation matrix (42 MB in, 42 MB
// out) with (1) linear streaming, (2) the GEMM K-tile pattern — 128-row panels fetched as ten 64-byte column slices, three
// slices in flight (what gemm_pers.hip / gemm_glds.hip do), (3) whole 80 KB row panels fetched contiguously up front.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_pattern.hip -o gpurun_out/hbm_pattern   (tools/r2_call11.sh)
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int M = 65536, K = 320, ROWB = K * 2;

__global__ __launch_bounds__(256) void linear_kernel(const uint4* __restrict__ a, uint4* __restrict__ o, long n16) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
    uint4 v = a[i];
    v.x ^= 1;
    o[i] = v;
  }
}

// panel of 128 rows per block; slice kt = bytes [64 kt, 64 kt + 64) of every row; DEPTH slices in flight
template <int DEPTH>
__global__ __launch_bounds__(256) void ktile_kernel(const char* __restrict__ a, char* __restrict__ o, int delay) {
  const int tid = threadIdx.x, row = tid >> 2, ch = tid & 3;
  for (int panel = blockIdx.x; panel < M / 128; panel += gridDim.x) {
    const char* base = a + (long)panel * 128 * ROWB;
    char* ob = o + (long)panel * 128 * ROWB;
    uint4 v[10][2];
#pragma unroll
    for (int kt = 0; kt < 10; ++kt) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        v[kt][i] = *reinterpret_cast<const uint4*>(base + (long)(row + 64 * i) * ROWB + kt * 64 + ch * 16);
      if (kt >= DEPTH - 1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (DEPTH - 1)) : "memory");
        for (int t = 0; t < delay; ++t) __builtin_amdgcn_s_sleep(8);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int kt = 0; kt < 10; ++kt)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        uint4 w = v[kt][i];
        w.x ^= 1;
        *reinterpret_cast<uint4*>(ob + (long)(row + 64 * i) * ROWB + kt * 64 + ch * 16) = w;
      }
  }
}

__global__ __launch_bounds__(256) void panel_kernel(const char* __restrict__ a, char* __restrict__ o) {
  const int tid = threadIdx.x;
  for (int panel = blockIdx.x; panel < M / 128; panel += gridDim.x) {
    const char* base = a + (long)panel * 128 * ROWB;
    char* ob = o + (long)panel * 128 * ROWB;
    uint4 v[20];
#pragma unroll
    for (int j = 0; j < 20; ++j) v[j] = *reinterpret_cast<const uint4*>(base + j * 4096 + tid * 16);
#pragma unroll
    for (int j = 0; j < 20; ++j) {
      uint4 w = v[j];
      w.x ^= 1;
      *reinterpret_cast<uint4*>(ob + j * 4096 + tid * 16) = w;
    }
  }
}

template <typename F>
float time_it(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / 20;
}

int main() {
  const size_t bytes = (size_t)M * ROWB;
  char *a, *o;
  hipMalloc(&a, bytes);
  hipMalloc(&o, bytes);
  hipMemset(a, 1, bytes);
  const double mb = 2.0 * bytes / 1e6;
  float t = time_it([&] { hipLaunchKernelGGL(linear_kernel, dim3(2048), dim3(256), 0, 0, (const uint4*)a, (uint4*)o, (long)(bytes / 16)); });
  printf("linear streaming            %7.1f us  %6.2f TB/s\n", t, mb / t);
  for (int grid : {512, 1024}) {
    t = time_it([&] { hipLaunchKernelGGL(panel_kernel, dim3(grid), dim3(256), 0, 0, a, o); });
    printf("row panels (80 KB) grid %4d  %7.1f us  %6.2f TB/s\n", grid, t, mb / t);
    for (int delay : {0, 4, 16}) {
      t = time_it([&] { hipLaunchKernelGGL(ktile_kernel<3>, dim3(grid), dim3(256), 0, 0, a, o, delay); });
      printf("K slices depth 3 grid %4d delay %2d  %7.1f us  %6.2f TB/s\n", grid, delay, t, mb / t);
    }
    t = time_it([&] { hipLaunchKernelGGL(ktile_kernel<10>, dim3(grid), dim3(256), 0, 0, a, o, 0); });
    printf("K slices depth 10 grid %4d  %7.1f us  %6.2f TB/s\n", grid, t, mb / t);
  }
  return 0;
}
