// Swin (shifted-)window attention core for SwinIR (reference swinir.py:120-151, 245-285).
//
// SwinIR is 0.16 % of the FLOPs of an image: this kernel is deliberately a simple f32 VALU
// kernel — one block per (window, head), one thread per query token; K/V of the window live in LDS — that fuses
// everything around the two small matmuls: cyclic roll, window partition / reverse, q scaling, relative
// position bias gather, the 0/-100 shift mask (computed from coordinates, no mask tensor), softmax and PV.
// The qkv / proj GEMMs around it run on the MFMA GEMM kernel.
#include "common.h"

namespace {

constexpr int MAX_HD = 32;
constexpr int MAX_N = 64;

template <typename T>
__global__ void window_attn_kernel(const u16* __restrict__ qkv, long long ld, u16* __restrict__ out, long long ldo,
                                   const float* __restrict__ bias_table, int H, int W, int C, int heads, int ws,
                                   int shift, float scale) {
  __shared__ float ks[MAX_N][MAX_HD + 1];
  __shared__ float vs[MAX_N][MAX_HD + 1];
  __shared__ int rid[MAX_N];
  const int N = ws * ws;
  const int hd = C / heads;
  const int i = threadIdx.x;  // query token within the window
  const int head = blockIdx.y;
  const int nwx = W / ws, nwy = H / ws;
  const int win = blockIdx.x % (nwx * nwy);
  const int b = blockIdx.x / (nwx * nwy);
  const int wy = i / ws, wx = i % ws;
  const int sy = (win / nwx) * ws + wy, sx = (win % nwx) * ws + wx;  // coords in the rolled image
  const int oy = (sy + shift) % H, ox = (sx + shift) % W;            // coords in the un-rolled image
  const long long row = ((long long)b * H + oy) * W + ox;
  const u16* base = qkv + row * ld + head * hd;
  float q[MAX_HD];
#pragma unroll
  for (int d = 0; d < MAX_HD; ++d) {
    if (d < hd) {
      q[d] = T::to_f32(base[d]) * scale;
      ks[i][d] = T::to_f32(base[C + d]);
      vs[i][d] = T::to_f32(base[2 * C + d]);
    } else {
      q[d] = 0.f;
      ks[i][d] = 0.f;
      vs[i][d] = 0.f;
    }
  }
  int my_rid = 0;
  if (shift > 0) {
    const int ry = sy < H - ws ? 0 : (sy < H - shift ? 1 : 2);
    const int rx = sx < W - ws ? 0 : (sx < W - shift ? 1 : 2);
    my_rid = ry * 3 + rx;
  }
  rid[i] = my_rid;
  __syncthreads();
  float s[MAX_N];
  float mx = -1e30f;
#pragma unroll
  for (int j = 0; j < MAX_N; ++j) {
    if (j < N) {
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < MAX_HD; ++d) a += q[d] * ks[j][d];
      const int jy = j / ws, jx = j % ws;
      const int ridx = (wy - jy + ws - 1) * (2 * ws - 1) + (wx - jx + ws - 1);
      a += bias_table[ridx * heads + head];
      if (shift > 0 && rid[j] != my_rid) a += -100.0f;
      s[j] = a;
      mx = fmaxf(mx, a);
    } else {
      s[j] = -1e30f;
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < MAX_N; ++j) {
    const float e = j < N ? __expf(s[j] - mx) : 0.f;
    s[j] = e;
    sum += e;
  }
  const float inv = 1.f / sum;
  float o[MAX_HD];
#pragma unroll
  for (int d = 0; d < MAX_HD; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < MAX_N; ++j) {
    if (j < N) {
      const float pj = s[j] * inv;
#pragma unroll
      for (int d = 0; d < MAX_HD; ++d) o[d] += pj * vs[j][d];
    }
  }
  u16* op = out + row * ldo + head * hd;
#pragma unroll
  for (int d = 0; d < MAX_HD; ++d)
    if (d < hd) op[d] = T::from_f32(o[d]);
}

}  // namespace

extern "C" int dbir_window_attention(int dtype, const void* qkv, long long ld, void* out, long long ldo,
                                     const float* bias_table, int B, int H, int W, int C, int heads, int ws,
                                     int shift, float scale, void* stream) {
  DBIR_CHECK_ARG(qkv && out && bias_table, "dbir_window_attention: null pointer");
  DBIR_CHECK_ARG(ws * ws <= MAX_N && C % heads == 0 && C / heads <= MAX_HD, "dbir_window_attention: ws^2<=64, hd<=32");
  DBIR_CHECK_ARG(H % ws == 0 && W % ws == 0 && shift >= 0 && shift < ws, "dbir_window_attention: bad H/W/shift");
  DBIR_CHECK_ARG(ld >= 3 * C && ldo >= C, "dbir_window_attention: bad ld");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((long long)B * (H / ws) * (W / ws)), heads);
  if (dtype == DBIR_F16)
    hipLaunchKernelGGL((window_attn_kernel<F16>), grid, dim3(ws * ws), 0, s, (const u16*)qkv, ld, (u16*)out, ldo,
                       bias_table, H, W, C, heads, ws, shift, scale);
  else if (dtype == DBIR_BF16)
    hipLaunchKernelGGL((window_attn_kernel<BF16>), grid, dim3(ws * ws), 0, s, (const u16*)qkv, ld, (u16*)out, ldo,
                       bias_table, H, W, C, heads, ws, shift, scale);
  else {
    dbir_set_error("dbir_window_attention: bad dtype");
    return DBIR_ERR_ARG;
  }
  DBIR_CHECK_LAUNCH("dbir_window_attention");
  return DBIR_OK;
}
