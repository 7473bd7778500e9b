// Swin (shifted-)window attention core for SwinIR (reference swinir.py:120-151, 245-285).
//
// Two kernels, both fusing everything around the two small matmuls — cyclic roll, window partition / reverse, q scaling,
// relative position bias gather, the 0/-100 shift mask (computed from coordinates, no mask tensor), softmax and PV; the
// qkv / proj GEMMs around them run on the MFMA GEMM kernel:
//   * window_attn_mfma_kernel (8x8 windows, head_dim <= 32: SwinIR's configuration): one wave64 per (window, head);
//     S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_32x32x16 with the head dimension zero-padded 30 -> 32, the same
//     "lane owns a query row" register dataflow as attention.hip (softmax in registers, P feeds the second MFMA from
//     the accumulators); K / V^T of the window are staged once per wave in LDS.
//   * window_attn_kernel: the general fallback (any window <= 64 tokens), f32 VALU, one thread per query token.
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

// ---------------------------------------------------------------------------------------------------------------
constexpr int WK_LD = 40;   // K row (halfs): 80 B -> conflict-free ds_read_b128 for 16 consecutive rows
constexpr int WV_LD = 72;   // V^T row (halfs): 144 B

template <typename T>
__global__ __launch_bounds__(256) void window_attn_mfma_kernel(const u16* __restrict__ qkv, long long ld,
                                                               u16* __restrict__ out, long long ldo,
                                                               const float* __restrict__ bias_table, int B, int H, int W,
                                                               int C, int heads, int shift, float scale) {
  constexpr int ws = 8, N = 64;
  __shared__ __attribute__((aligned(16))) u16 Ksh[4][N * WK_LD];
  __shared__ __attribute__((aligned(16))) u16 Vsh[4][32 * WV_LD];
  __shared__ int ridsh[4][N];
  __shared__ float bsh[15 * 15 * 8];   // relative position bias [225][heads <= 8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, lq = lane & 31;
  const int hd = C / heads;
  const int nwx = W / ws, nwy = H / ws;
  for (int i = tid; i < 225 * heads; i += 256) bsh[i] = bias_table[i];
  const long long items = (long long)B * nwx * nwy * heads;
  const long long item = (long long)blockIdx.x * 4 + wave;
  const bool live = item < items;
  const int head = live ? (int)(item % heads) : 0;
  const long long wi = live ? item / heads : 0;
  const int win = (int)(wi % (nwx * nwy)), b = (int)(wi / (nwx * nwy));
  u16* Ks = Ksh[wave];
  u16* Vs = Vsh[wave];
  // token -> pixel row of the (un-rolled) feature map; lane t stages token t of the window
  auto token_row = [&](int t, int& rid) -> long long {
    const int wy = t >> 3, wx = t & 7;
    const int sy = (win / nwx) * ws + wy, sx = (win % nwx) * ws + wx;   // coords in the rolled image
    rid = 0;
    if (shift > 0) {
      const int ry = sy < H - ws ? 0 : (sy < H - shift ? 1 : 2);
      const int rx = sx < W - ws ? 0 : (sx < W - shift ? 1 : 2);
      rid = ry * 3 + rx;
    }
    const int oy = (sy + shift) % H, ox = (sx + shift) % W;
    return ((long long)b * H + oy) * W + ox;
  };
  {
    int rid;
    const long long row = token_row(lane, rid);
    ridsh[wave][lane] = rid;
    const u16* kp = qkv + row * ld + C + head * hd;
    const u16* vp = kp + C;
#pragma unroll
    for (int d2 = 0; d2 < 16; ++d2) {   // 2 halfs at a time (rows are 4-byte aligned: head * hd * 2 B with hd even)
      uint32_t kk = 0, vv = 0;
      if (live && 2 * d2 + 1 < hd) {
        kk = *reinterpret_cast<const uint32_t*>(kp + 2 * d2);
        vv = *reinterpret_cast<const uint32_t*>(vp + 2 * d2);
      } else if (live && 2 * d2 < hd) {
        kk = kp[2 * d2];
        vv = vp[2 * d2];
      }
      *reinterpret_cast<uint32_t*>(&Ks[lane * WK_LD + 2 * d2]) = kk;
      Vs[(2 * d2) * WV_LD + lane] = (u16)(vv & 0xffff);
      Vs[(2 * d2 + 1) * WV_LD + lane] = (u16)(vv >> 16);
    }
  }
  __syncthreads();   // bias table + this wave's K / V^T / region ids (uniform: every wave reaches it)
  if (!live) return;
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int tq = qb * 32 + lq;
    int my_rid;
    const long long qrow = token_row(tq, my_rid);
    // Q fragments (B operand): Q[tq][16*ks + 8*hi .. +7], zero beyond hd
    typename T::vec8 qf[2];
    {
      const u16* qp = qkv + qrow * ld + head * hd;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint32_t w4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int d0 = 16 * ks + 8 * hi + 2 * j;
          w4[j] = d0 + 1 < hd ? *reinterpret_cast<const uint32_t*>(qp + d0) : (d0 < hd ? (uint32_t)qp[d0] : 0u);
        }
        qf[ks] = __builtin_bit_cast(typename T::vec8, make_uint4(w4[0], w4[1], w4[2], w4[3]));
      }
    }
    f32x16 s_acc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s_acc[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const typename T::vec8 kf =
            *reinterpret_cast<const typename T::vec8*>(&Ks[(kb * 32 + lq) * WK_LD + ks * 16 + hi * 8]);
        s_acc[kb] = T::mfma32(kf, qf[ks], s_acc[kb]);
      }
    }
    const int qy = tq >> 3, qx = tq & 7;
    float mx = -1e30f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int ridx = (qy - (key >> 3) + ws - 1) * (2 * ws - 1) + (qx - (key & 7) + ws - 1);
        float a = s_acc[kb][r] * scale + bsh[ridx * heads + head];
        if (shift > 0 && ridsh[wave][key] != my_rid) a += -100.0f;
        s_acc[kb][r] = a;
        mx = fmaxf(mx, a);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __expf(s_acc[kb][r] - mx);
        s_acc[kb][r] = pv;
        psum += pv;
      }
    psum += __shfl_xor(psum, 32, 64);
    f32x16 o_acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      float pf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = s_acc[st >> 1][8 * (st & 1) + j];
      const uint4 pp = pack8<T>(pf);
      const u16* vrow = &Vs[lq * WV_LD + 16 * st + 4 * hi];
      const uint2 v0 = *reinterpret_cast<const uint2*>(vrow);
      const uint2 v1 = *reinterpret_cast<const uint2*>(vrow + 8);
      const uint4 vv = make_uint4(v0.x, v0.y, v1.x, v1.y);
      o_acc = T::mfma32(__builtin_bit_cast(typename T::vec8, vv), __builtin_bit_cast(typename T::vec8, pp), o_acc);
    }
    const float inv = 1.f / psum;
    u16* op = out + qrow * ldo + head * hd;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = 8 * g + 4 * hi;
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        if (d0 + e + 1 < hd) {
          *reinterpret_cast<uint32_t*>(op + d0 + e) =
              (uint32_t)T::from_f32(o_acc[4 * g + e] * inv) | ((uint32_t)T::from_f32(o_acc[4 * g + e + 1] * inv) << 16);
        } else if (d0 + e < hd) {
          op[d0 + e] = T::from_f32(o_acc[4 * g + e] * inv);
        }
      }
    }
  }
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
  if (dtype != DBIR_F16 && dtype != DBIR_BF16) {
    dbir_set_error("dbir_window_attention: bad dtype");
    return DBIR_ERR_ARG;
  }
  const int hd = C / heads;
  if (ws == 8 && heads <= 8 && hd % 2 == 0 && ld % 2 == 0 && ldo % 2 == 0 && C % 2 == 0 &&
      (reinterpret_cast<uintptr_t>(qkv) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
    const long long items = (long long)B * (H / ws) * (W / ws) * heads;
    const dim3 gridm((unsigned)((items + 3) / 4));
    if (dtype == DBIR_F16)
      hipLaunchKernelGGL((window_attn_mfma_kernel<F16>), gridm, dim3(256), 0, s, (const u16*)qkv, ld, (u16*)out, ldo,
                         bias_table, B, H, W, C, heads, shift, scale);
    else
      hipLaunchKernelGGL((window_attn_mfma_kernel<BF16>), gridm, dim3(256), 0, s, (const u16*)qkv, ld, (u16*)out, ldo,
                         bias_table, B, H, W, C, heads, shift, scale);
    DBIR_CHECK_LAUNCH("dbir_window_attention(mfma)");
    return DBIR_OK;
  }
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
