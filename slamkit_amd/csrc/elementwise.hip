// HBM-bound kernels of the Slam forward/backward/optimizer step for gfx950.
// One wave64 per token row for the row-wise ops, 16-byte (bf16x8) accesses everywhere.
// Reference semantics: site-packages transformers/models/qwen2/modeling_qwen2.py
//   RMSNorm :247-252, RoPE :91-135, SwiGLU :41-48, embedding :356;
// loss: /root/reference slamkit/model/unit_lm.py:13-29; optimizer: torch AdamW + HF
// clip_grad_norm_.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int MAXC_LIMIT = 4;  // chunks of 8 per lane -> hidden <= 2048 (kernels are instantiated for 1..4)

// ------------------------------------------------------------------------------------------
// RMSNorm forward: y = bf16( x * rsqrt(mean(x^2)+eps) * w ), fp32 math, rstd saved.
// Algorithmic traffic: 4 B/element (read bf16 + write bf16).
template <int MAXC>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, float* __restrict__ rstd,
                                                          int M, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = H >> 3;
  const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * H);
  uint4 v[MAXC];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    int c = lane + 64 * i;
    if (c < nch) {
      v[i] = xr[c];
      float f[8];
      unpack_bf16x8(v[i], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
    }
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(ss / (float)H + eps);
  if (lane == 0 && rstd) rstd[row] = r;
  uint4* yr = reinterpret_cast<uint4*>(y + (size_t)row * H);
  const uint4* wr = reinterpret_cast<const uint4*>(w);
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    int c = lane + 64 * i;
    if (c < nch) {
      float f[8], g[8];
      unpack_bf16x8(v[i], f);
      unpack_bf16x8(wr[c], g);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = f[j] * r * g[j];
      yr[c] = pack_bf16x8(f);
    }
  }
}

// RMSNorm backward. dx = rstd*(dy*w - xhat*mean(dy*w*xhat)) (+ dres); dw partial per block.
// Each wave walks rows wave, wave+4*gridDim.. accumulating its dw slice in registers.
// Algorithmic traffic: 6 B/element (+2 with the fused residual-gradient add).
template <int MAXC>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy,
                                                          const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w,
                                                          const float* __restrict__ rstd,
                                                          const bf16_t* __restrict__ dres,
                                                          bf16_t* __restrict__ dx, float* __restrict__ dw_part,
                                                          int M, int H) {
  __shared__ float red[4][MAXC * 64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = H >> 3;
  float dwa[MAXC][8];
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) dwa[i][j] = 0.f;
  const uint4* wr = reinterpret_cast<const uint4*>(w);
  float wv[MAXC][8];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    int c = lane + 64 * i;
    if (c < nch) unpack_bf16x8(wr[c], wv[i]);
  }
  // software-pipelined over rows: the next row's loads are in flight while this row is reduced
  auto load_row = [&](int row, uint4* rx, uint4* rdy, uint4* rdr) {
    const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * H);
    const uint4* dyr = reinterpret_cast<const uint4*>(dy + (size_t)row * H);
    const uint4* drr = dres ? reinterpret_cast<const uint4*>(dres + (size_t)row * H) : nullptr;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      int c = lane + 64 * i;
      if (c < nch) {
        rx[i] = xr[c];
        rdy[i] = dyr[c];
        if (drr) rdr[i] = drr[c];
      }
    }
  };
  const int rstep = gridDim.x * 4;
  int row = blockIdx.x * 4 + wave;
  uint4 cx[MAXC], cdy[MAXC], cdr[MAXC];
  float cr = 0.f;
  if (row < M) { load_row(row, cx, cdy, cdr); cr = rstd[row]; }
  for (; row < M; row += rstep) {
    uint4 nx[MAXC], ndy[MAXC], ndr[MAXC];
    float nr = 0.f;
    const int nrow = row + rstep;
    if (nrow < M) { load_row(nrow, nx, ndy, ndr); nr = rstd[nrow]; }
    const float r = cr;
    float xh[MAXC][8], gy[MAXC][8];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      int c = lane + 64 * i;
      if (c < nch) {
        float fx[8], fd[8];
        unpack_bf16x8(cx[i], fx);
        unpack_bf16x8(cdy[i], fd);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xh[i][j] = fx[j] * r;
          gy[i][j] = fd[j] * wv[i][j];
          dot += gy[i][j] * xh[i][j];
          dwa[i][j] += fd[j] * xh[i][j];
        }
      }
    }
    dot = wave_sum(dot) / (float)H;
    uint4* dxr = reinterpret_cast<uint4*>(dx + (size_t)row * H);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      int c = lane + 64 * i;
      if (c < nch) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = r * (gy[i][j] - xh[i][j] * dot);
        if (dres) {
          float a[8];
          unpack_bf16x8(cdr[i], a);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += a[j];
        }
        dxr[c] = pack_bf16x8(o);
      }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) { cx[i] = nx[i]; cdy[i] = ndy[i]; cdr[i] = ndr[i]; }
    cr = nr;
  }
  // cross-wave reduction of the dw partials, then one row of dw_part per block
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[wave][(lane + 64 * i) * 8 + j] = dwa[i][j];
  __syncthreads();
  for (int e = threadIdx.x; e < H; e += 256)
    dw_part[(size_t)blockIdx.x * H + e] = red[0][e] + red[1][e] + red[2][e] + red[3][e];
}

// column sums of a bf16 matrix (bias gradient): part[blockIdx.y][N] fp32. Block = 16 column chunks (8 columns
// each) x 16 row lanes; a row lane walks rows lane, lane + 16 gridDim.y, ...; the 16 row lanes are combined in
// LDS in a fixed order. Algorithmic traffic: 2 B/element read.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ X, int ld, int M, int N,
                                                          float* __restrict__ part) {
  __shared__ float red[16][16 * 8 + 1];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;  // chunk index
  const bool ok = c * 8 < N;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (ok) {
    for (int m = blockIdx.y * 16 + rl; m < M; m += gridDim.y * 16) {
      float f[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(X + (size_t)m * ld + c * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][cl * 8 + j] = s[j];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int col = blockIdx.x * 128 + threadIdx.x;
    if (col < N) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
      part[(size_t)blockIdx.y * N + col] = t;
    }
  }
}

// out[c] = (acc? out[c]:0) + sum_b part[b][c]; block = 16 columns x 16 row-slices, fixed order.
// blockIdx.y selects one of several equally shaped instances (per-layer partial slabs finished together).
// img_only / sumsq: GradSink semantics (kernels.h) - slot = blockIdx.y * gridDim.x + blockIdx.x
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int nb, int N,
                                                            float* __restrict__ out, int accumulate,
                                                            size_t part_stride, size_t out_stride, bf16_t* __restrict__ img,
                                                            int img_only, float* __restrict__ sumsq) {
  __shared__ float red[16][17];
  part += (size_t)blockIdx.y * part_stride;
  out += (size_t)blockIdx.y * out_stride;
  if (img) img += (size_t)blockIdx.y * out_stride;
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < N)
    for (int b = sl; b < nb; b += 16) s += part[(size_t)b * N + c];
  red[sl][cl] = s;
  __syncthreads();
  float sq = 0.f;
  if (sl == 0 && c < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cl];
    const float o = (accumulate ? out[c] : 0.f) + t;
    if (!img_only) { out[c] = o; sq = o * o; }
    if (img) {
      const bf16_t b = f32_to_bf16(o);
      img[c] = b;
      if (img_only) { const float r = bf16_to_f32(b); sq = r * r; }
    }
  }
  if (sumsq) {  // the 16 column results of the block, added in column order
    __syncthreads();
    if (sl == 0) red[0][cl] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[0][k];
      sumsq[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------
// RoPE tables (fp32 cos/sin [M][hd/2]) from positions; position_ids == nullptr -> m % T.
__global__ void rope_table_kernel(const int64_t* __restrict__ pos, int M, int T, int half, float theta,
                                  float* __restrict__ cs, float* __restrict__ sn, float* __restrict__ csq,
                                  float* __restrict__ snq, float qscale) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * half) return;
  int m = i / half, d = i % half;
  float p = pos ? (float)pos[m] : (float)(m % T);
  float inv = 1.0f / powf(theta, (float)(2 * d) / (float)(2 * half));
  float a = p * inv;
  float s, c;
  sincosf(a, &s, &c);
  cs[i] = c;
  sn[i] = s;
  if (csq) {  // the query heads' table: the rotation and the softmax scale * log2(e) in one multiply (attention.hip)
    csq[i] = c * qscale;
    snq[i] = s * qscale;
  }
}

// In-place rotate-half RoPE on the first `nrot` heads of each row of qkv [M][ld] (head_dim hd, a
// multiple of 16). dir = +1 forward, -1 backward (transpose rotation).
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ qkv, int ld, int M, int nrot, int hd,
                                                   const float* __restrict__ cs, const float* __restrict__ sn,
                                                   float dir, int q_heads, float q_scale) {
  // one thread: 8 low-half elems + their 8 high-half partners of one head; hd/16 threads per head
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int tph = hd >> 4, half = hd >> 1;
  int per_row = nrot * tph;
  if (idx >= (size_t)M * per_row) return;
  int m = (int)(idx / per_row), r = (int)(idx % per_row);
  int head = r / tph, part = r % tph;
  bf16_t* base = qkv + (size_t)m * ld + head * hd + part * 8;
  uint4 lo = *reinterpret_cast<uint4*>(base), hi = *reinterpret_cast<uint4*>(base + half);
  float a[8], b[8], c[8], s[8];
  unpack_bf16x8(lo, a);
  unpack_bf16x8(hi, b);
  const float4* cp = reinterpret_cast<const float4*>(cs + (size_t)m * half + part * 8);
  const float4* sp = reinterpret_cast<const float4*>(sn + (size_t)m * half + part * 8);
  *reinterpret_cast<float4*>(c) = cp[0]; *reinterpret_cast<float4*>(c + 4) = cp[1];
  *reinterpret_cast<float4*>(s) = sp[0]; *reinterpret_cast<float4*>(s + 4) = sp[1];
  float o1[8], o2[8];
  const float hs = head < q_heads ? q_scale : 1.f;  // query heads leave pre-scaled by scale * log2(e)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float sj = s[j] * dir;
    o1[j] = (a[j] * c[j] - b[j] * sj) * hs;
    o2[j] = (b[j] * c[j] + a[j] * sj) * hs;
  }
  *reinterpret_cast<uint4*>(base) = pack_bf16x8(o1);
  *reinterpret_cast<uint4*>(base + half) = pack_bf16x8(o2);
}

// ------------------------------------------------------------------------------------------
// SwiGLU: gu [M][2I] -> act [M][I]. Column layout of gu: blocks of `blk` gate columns followed by the
// matching `blk` up columns (blk = I: the plain gate|up halves; blk = 32: the engine's interleaved
// layout that puts a gate/up pair into the same MFMA lane of the gate|up GEMM epilogue).
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ act,
                                                         size_t M, int I, int blk) {
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  int nch = I >> 3;
  if (idx >= M * nch) return;
  size_t m = idx / nch;
  int c = idx % nch;
  const int a0 = c * 8;
  const bf16_t* row = gu + m * 2 * I + (a0 / blk) * 2 * blk + (a0 % blk);
  float g[8], u[8], o[8];
  unpack_bf16x8(*reinterpret_cast<const uint4*>(row), g);
  unpack_bf16x8(*reinterpret_cast<const uint4*>(row + blk), u);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = g[j] * fast_sigmoid(g[j]) * u[j];
  *reinterpret_cast<uint4*>(act + m * I + c * 8) = pack_bf16x8(o);
}

// dgu (written in place over gu) from dact
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(bf16_t* __restrict__ gu, const bf16_t* __restrict__ dact,
                                                         size_t M, int I, int blk) {
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  int nch = I >> 3;
  if (idx >= M * nch) return;
  size_t m = idx / nch;
  int c = idx % nch;
  const int a0 = c * 8;
  bf16_t* row = gu + m * 2 * I + (a0 / blk) * 2 * blk + (a0 % blk);
  float g[8], u[8], d[8], dg[8], du[8];
  unpack_bf16x8(*reinterpret_cast<const uint4*>(row), g);
  unpack_bf16x8(*reinterpret_cast<const uint4*>(row + blk), u);
  unpack_bf16x8(*reinterpret_cast<const uint4*>(dact + m * I + c * 8), d);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float sg = fast_sigmoid(g[j]);
    float silu = g[j] * sg;
    du[j] = d[j] * silu;
    dg[j] = d[j] * u[j] * sg * (1.f + g[j] * (1.f - sg));
  }
  *reinterpret_cast<uint4*>(row) = pack_bf16x8(dg);
  *reinterpret_cast<uint4*>(row + blk) = pack_bf16x8(du);
}

// ------------------------------------------------------------------------------------------
// Embedding gather: out[m,:] = E[ids[m],:]
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids,
                                                        const bf16_t* __restrict__ E, bf16_t* __restrict__ out,
                                                        size_t M, int H, int V) {
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  int nch = H >> 3;
  if (idx >= M * nch) return;
  size_t m = idx / nch;
  int c = idx % nch;
  int64_t id = ids[m];
  if (id < 0 || id >= V) id = 0;
  *reinterpret_cast<uint4*>(out + m * H + c * 8) = *reinterpret_cast<const uint4*>(E + (size_t)id * H + c * 8);
}

// One-hot rows for the gather-side embedding gradient (dE += onehot^T dh0 runs on the wgrad GEMM,
// deterministic); the padding_idx column is suppressed like nn.Embedding(padding_idx).
__global__ __launch_bounds__(256) void onehot_kernel(const int64_t* __restrict__ ids, bf16_t* __restrict__ oh,
                                                     size_t M, int Vp, int V, int pad_id) {
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  int nch = Vp >> 3;
  if (idx >= M * nch) return;
  size_t m = idx / nch;
  int c = idx % nch;
  int64_t id = ids[m];
  uint32_t w[4] = {0, 0, 0, 0};
  if (id >= 0 && id < V && id != pad_id && (id >> 3) == c) {
    int j = (int)(id & 7);
    w[j >> 1] = (j & 1) ? 0x3f800000u : 0x00003f80u;
  }
  *reinterpret_cast<uint4*>(oh + m * Vp + c * 8) = make_uint4(w[0], w[1], w[2], w[3]);
}

// ------------------------------------------------------------------------------------------
// Shifted cross-entropy over bf16 logits [M][Vp] (Vp = 512 padded, V valid columns).
// target(m) = labels[m+1] when m is not the last position of its batch row, else ignore.
__global__ void count_valid_kernel(const int64_t* __restrict__ labels, int B, int T, double num_items,
                                   float* __restrict__ denom) {
  __shared__ int red[256];
  int cnt = 0;
  if (num_items <= 0.0) {
    for (int i = threadIdx.x; i < B * T; i += 256) {
      int t = i % T;
      if (t < T - 1 && labels[i + 1] != -100) ++cnt;
    }
  }
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) denom[0] = num_items > 0.0 ? (float)num_items : (float)red[0];
}

// one wave per row, Vp == 512 (64 lanes x 8)
// (logits and dlogits may be the SAME buffer: a lane reads its chunk into registers before it writes it)
__global__ __launch_bounds__(256) void ce_kernel(const bf16_t* logits,
                                                 const int64_t* __restrict__ labels,
                                                 const float* __restrict__ denom, bf16_t* dlogits,
                                                 float* __restrict__ row_loss, int B, int T, int Vp, int V,
                                                 const uint8_t* __restrict__ colmask) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int M = B * T;
  if (m >= M) return;
  const int t = m % T;
  int64_t tgt = (t < T - 1) ? labels[m + 1] : -100;
  const bool valid = (tgt >= 0 && tgt < V);
  uint4* dl = dlogits ? reinterpret_cast<uint4*>(dlogits + (size_t)m * Vp) + lane : nullptr;
  if (!valid) {  // wave-uniform
    if (dl) *dl = make_uint4(0, 0, 0, 0);
    if (lane == 0) row_loss[m] = 0.f;
    return;
  }
  float f[8];
  unpack_bf16x8(reinterpret_cast<const uint4*>(logits + (size_t)m * Vp)[lane], f);
  // columns >= V (padding) and columns flagged in colmask (modality-restricted scoring: the reference sets
  // those logits to -inf, unit_lm.py:187-188) are outside the softmax
  bool on[8];
  {
    uint2 mk = colmask ? *reinterpret_cast<const uint2*>(colmask + lane * 8) : make_uint2(0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) on[j] = (lane * 8 + j < V) && !(((j < 4 ? mk.x : mk.y) >> (8 * (j & 3))) & 0xff);
  }
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (!on[j]) f[j] = -3.0e38f;
    mx = fmaxf(mx, f[j]);
  }
  mx = wave_max(mx);
  float e[8], s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    e[j] = on[j] ? __expf(f[j] - mx) : 0.f;
    s += e[j];
  }
  s = wave_sum(s);
  const float lse = mx + logf(s);
  // logit of the target
  float tl = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (lane * 8 + j == (int)tgt) tl = f[j];
  tl = wave_sum(tl);
  if (lane == 0) row_loss[m] = (colmask && colmask[tgt]) ? INFINITY : lse - tl;
  if (dl) {
    const float sc = 1.f / denom[0];
    const float inv = 1.f / s;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float pj = e[j] * inv;
      if (lane * 8 + j == (int)tgt) pj -= 1.f;
      o[j] = pj * sc;
    }
    *dl = pack_bf16x8(o);
  }
}

// Large vocabulary (Vp > 512, a multiple of 8): one 256-thread block per row, two passes over the
// row (online max / sum, then the gradient); the second read hits L2 (a 152k-column row is 300 KB).
// Algorithmic traffic: 2 B/logit read + 2 B/logit written.
// (logits and dlogits may be the SAME buffer - the engine's training path: the gradient replaces the logits chunk by
//  chunk in the second pass, after the barrier behind the first pass and after thread 0 has read the target logit)
__global__ __launch_bounds__(256) void ce_big_kernel(const bf16_t* logits,
                                                     const int64_t* __restrict__ labels,
                                                     const float* __restrict__ denom, bf16_t* dlogits,
                                                     float* __restrict__ row_loss, int B, int T, int Vp, int V,
                                                     const uint8_t* __restrict__ colmask) {
  __shared__ float red_m[4], red_s[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = blockIdx.x;
  const int t = m % T;
  const int64_t tgt = (t < T - 1) ? labels[m + 1] : -100;
  const bool valid = (tgt >= 0 && tgt < V);
  const int nch = Vp >> 3;
  const uint4* lr = reinterpret_cast<const uint4*>(logits + (size_t)m * Vp);
  uint4* dl = dlogits ? reinterpret_cast<uint4*>(dlogits + (size_t)m * Vp) : nullptr;
  if (!valid) {  // block-uniform
    if (dl)
      for (int c = tid; c < nch; c += 256) dl[c] = make_uint4(0, 0, 0, 0);
    if (tid == 0) row_loss[m] = 0.f;
    return;
  }
  const float tgt_logit = tid == 0 ? bf16_to_f32(logits[(size_t)m * Vp + tgt]) : 0.f;  // read before any in-place write
  float mx = -3.0e38f, sm = 0.f;
  for (int c = tid; c < nch; c += 256) {
    float f[8];
    unpack_bf16x8(lr[c], f);
    const uint2 mk = colmask ? *reinterpret_cast<const uint2*>(colmask + c * 8) : make_uint2(0, 0);
    bool on[8];
    float cm = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      on[j] = (c * 8 + j < V) && !(((j < 4 ? mk.x : mk.y) >> (8 * (j & 3))) & 0xff);
      if (!on[j]) f[j] = -3.0e38f;
      cm = fmaxf(cm, f[j]);
    }
    const float nm = fmaxf(mx, cm);
    float cs = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) cs += on[j] ? __expf(f[j] - nm) : 0.f;
    sm = sm * __expf(mx - nm) + cs;
    mx = nm;
  }
  const float wm = wave_max(mx);
  sm = wave_sum(sm * __expf(mx - wm));
  if (lane == 0) { red_m[wave] = wm; red_s[wave] = sm; }
  __syncthreads();
  const float bm = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
  float bs = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) bs += red_s[w] * __expf(red_m[w] - bm);
  const float lse = bm + logf(bs);
  if (tid == 0) row_loss[m] = (colmask && colmask[tgt]) ? INFINITY : lse - tgt_logit;
  if (dl) {
    const float sc = 1.f / denom[0];
    for (int c = tid; c < nch; c += 256) {
      float f[8], o[8];
      unpack_bf16x8(lr[c], f);
      const uint2 mk = colmask ? *reinterpret_cast<const uint2*>(colmask + c * 8) : make_uint2(0, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = c * 8 + j;
        const bool onj = (col < V) && !(((j < 4 ? mk.x : mk.y) >> (8 * (j & 3))) & 0xff);
        float pj = onj ? __expf(f[j] - lse) : 0.f;
        if (col == (int)tgt) pj -= 1.f;
        o[j] = pj * sc;
      }
      dl[c] = pack_bf16x8(o);
    }
  }
}

// ---- gather-side embedding gradient for large vocabularies -------------------------------------
// dE[v] += sum over tokens m with ids[m] == v of dh[m], summed in token order (deterministic, no float
// atomics): rank[m] = number of earlier tokens with the same id (brute force through LDS, M^2/2
// integer compares), count[v] by integer atomics, exclusive scan -> list of token indices per id.
__global__ __launch_bounds__(256) void embed_rank_kernel(const int64_t* __restrict__ ids, int M, int V, int pad_id,
                                                         int* __restrict__ rank, int* __restrict__ count) {
  __shared__ int sid[256];
  const int m = blockIdx.x * 256 + threadIdx.x;
  int64_t id64 = m < M ? ids[m] : -1;
  const int id = (id64 >= 0 && id64 < V && id64 != pad_id) ? (int)id64 : -1;
  int r = 0;
  for (int base = 0; base <= blockIdx.x * 256; base += 256) {
    const int j = base + threadIdx.x;
    int64_t v = j < M ? ids[j] : -1;
    __syncthreads();
    sid[threadIdx.x] = (v >= 0 && v < V) ? (int)v : -2;
    __syncthreads();
    const int lim = min(256, m - base);  // only tokens before m
    for (int k = 0; k < lim; ++k) r += (sid[k] == id);
  }
  if (m < M) {
    rank[m] = r;
    if (id >= 0) atomicAdd(count + id, 1);
  }
}
// offset = exclusive scan of count (single block of 1024 threads, each a contiguous slice)
__global__ __launch_bounds__(1024) void embed_scan_kernel(const int* __restrict__ count, int* __restrict__ offset, int V) {
  __shared__ int part[1024];
  const int per = (V + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = min(V, lo + per);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += count[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - s;
  for (int i = lo; i < hi; ++i) { offset[i] = run; run += count[i]; }
}
__global__ __launch_bounds__(256) void embed_fill_kernel(const int64_t* __restrict__ ids, int M, int V, int pad_id,
                                                         const int* __restrict__ rank, const int* __restrict__ offset,
                                                         int* __restrict__ list) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int64_t id = ids[m];
  if (id >= 0 && id < V && id != pad_id) list[offset[id] + rank[m]] = m;
}
// one block per vocabulary row with at least one token; thread = 8 columns
__global__ __launch_bounds__(256) void embed_scatter_kernel(const bf16_t* __restrict__ dh, float* __restrict__ dE, int H,
                                                            const int* __restrict__ count, const int* __restrict__ offset,
                                                            const int* __restrict__ list) {
  const int v = blockIdx.x;
  const int n = count[v];
  if (n == 0) return;
  const int* l = list + offset[v];
  for (int c = threadIdx.x; c < (H >> 3); c += 256) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < n; ++j) {
      float f[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(dh + (size_t)l[j] * H + c * 8), f);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += f[k];
    }
    float4* o = reinterpret_cast<float4*>(dE + (size_t)v * H + c * 8);
    float4 a = o[0], b = o[1];
    o[0] = make_float4(a.x + acc[0], a.y + acc[1], a.z + acc[2], a.w + acc[3]);
    o[1] = make_float4(b.x + acc[4], b.y + acc[5], b.z + acc[6], b.w + acc[7]);
  }
}

// loss = sum(row_loss) / denom  (single block, fixed order -> deterministic)
__global__ void loss_finish_kernel(const float* __restrict__ row_loss, int M, const float* __restrict__ denom,
                                   float* __restrict__ loss) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < M; i += 256) s += (double)row_loss[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = denom[0] > 0.f ? (float)(red[0] / (double)denom[0]) : 0.f;
}

// Per-sequence sum of -row_loss over valid targets (log-likelihood, unit_lm.py:184-194 shape)
__global__ void seq_loglik_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ labels, int B,
                                  int T, float* __restrict__ ll, float* __restrict__ cnt) {
  int b = blockIdx.x;
  __shared__ float rs[256], rc[256];
  float s = 0.f, c = 0.f;
  for (int t = threadIdx.x; t < T - 1; t += 256) {
    if (labels[(size_t)b * T + t + 1] != -100) { s -= row_loss[(size_t)b * T + t]; c += 1.f; }
  }
  rs[threadIdx.x] = s; rc[threadIdx.x] = c;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) { rs[threadIdx.x] += rs[threadIdx.x + k]; rc[threadIdx.x] += rc[threadIdx.x + k]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { ll[b] = rs[0]; cnt[b] = rc[0]; }
}

// strided 2-D bf16 copy in 16-byte chunks (padded logits -> user logits needs element copy; see below)
__global__ __launch_bounds__(256) void copy_cols_kernel(const bf16_t* __restrict__ src, int lds_, bf16_t* __restrict__ dst,
                                                        int ldd, size_t M, int ncols) {
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * (size_t)ncols) return;
  size_t m = idx / ncols;
  int c = idx % ncols;
  dst[m * ldd + c] = src[m * lds_ + c];
}

// dlogits[m][:] *= coef[m / T]  (per-sequence loss weights: DPO's +-beta*sigmoid(-x)/n)
__global__ __launch_bounds__(256) void scale_rows_bf16_kernel(bf16_t* __restrict__ x, const float* __restrict__ coef,
                                                              size_t M, int T, int chunks_per_row) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * chunks_per_row) return;
  const float s = coef[(i / chunks_per_row) / T];
  float f[8];
  uint4* p = reinterpret_cast<uint4*>(x) + i;
  unpack_bf16x8(*p, f);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] *= s;
  *p = pack_bf16x8(f);
}

__global__ __launch_bounds__(256) void scale_bf16_kernel(bf16_t* __restrict__ x, size_t nchunks, float s) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nchunks) return;
  float f[8];
  uint4* p = reinterpret_cast<uint4*>(x) + i;
  unpack_bf16x8(*p, f);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] *= s;
  *p = pack_bf16x8(f);
}

// ------------------------------------------------------------------------------------------
// Gradient norm (fp32 flat buffer) -> clip coefficient, and fused AdamW.
// Canonical chunked sum of squares: chunk k = elements [k C, (k+1) C) of the flat gradient buffer (absolute positions,
// C = GRAD_CHUNK), one block per chunk, fixed summation order inside it. Any partition of the buffer into chunk-aligned
// ranges - the whole buffer on one GPU, one 1/N shard per bucket per rank under the sharded optimizer - produces the
// same chunk sums bit for bit; norm_finish_kernel adds them in fp64 in chunk order.
constexpr int GRAD_CHUNK = 8192;
constexpr int CHUNKS_PER_BLOCK = 16;  // a block walks 16 consecutive chunks (512 KB): fewer, longer blocks stream better
// GT = float: the fp32 gradient buffer; GT = bf16_t: gradients kept in bf16 (same element order, so a value that is exactly
// representable in bf16 gives the same chunk sum through either instantiation)
template <typename GT>
__global__ __launch_bounds__(256) void sumsq_chunks_kernel(const GT* __restrict__ g, size_t n, size_t first_chunk, size_t n_chunks,
                                                           float* __restrict__ chunk_sums) {
  __shared__ float red[CHUNKS_PER_BLOCK][4];
  const size_t kb = (size_t)blockIdx.x * CHUNKS_PER_BLOCK;
  for (int c = 0; c < CHUNKS_PER_BLOCK && kb + c < n_chunks; ++c) {
    const size_t k = first_chunk + kb + c;
    const size_t lo = k * GRAD_CHUNK, hi = lo + GRAD_CHUNK < n ? lo + GRAD_CHUNK : n;
    float4 v[GRAD_CHUNK / 1024];
#pragma unroll
    for (int it = 0; it < GRAD_CHUNK / 1024; ++it) {  // all eight loads in flight before the first add
      const size_t i = lo + (size_t)(it * 256 + threadIdx.x) * 4;
      if (i < hi) {
        if constexpr (sizeof(GT) == 4) {
          v[it] = *reinterpret_cast<const float4*>(g + i);
        } else {
          const uint2 w = *reinterpret_cast<const uint2*>(g + i);
          v[it] = make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                              __uint_as_float(w.y & 0xffff0000u));
        }
      } else {
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < GRAD_CHUNK / 1024; ++it) s += v[it].x * v[it].x + v[it].y * v[it].y + v[it].z * v[it].z + v[it].w * v[it].w;
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x < CHUNKS_PER_BLOCK && kb + threadIdx.x < n_chunks)
    chunk_sums[first_chunk + kb + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}
// out[0] = ||g||, out[1] = clip coefficient min(1, max_norm/(norm+1e-6)) (1 when max_norm<=0)
// (one block of 1024: a thread adds its strided elements in four independent fp64 chains - the 43,760 chunk sums of the
// Slam-358M buffer took 60 us on 256 threads with one dependent chain each; the order is fixed, so every caller - the
// replicated and the sharded clip - gets the same bits)
__global__ __launch_bounds__(1024) void norm_finish_kernel(const float* __restrict__ part, int nb, float max_norm, float* __restrict__ out) {
  __shared__ double red[1024];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * 1024 < nb; i += 4 * 1024) {
    s0 += (double)part[i]; s1 += (double)part[i + 1024]; s2 += (double)part[i + 2048]; s3 += (double)part[i + 3072];
  }
  for (; i < nb; i += 1024) s0 += (double)part[i];
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int k = 512; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float nrm = (float)sqrt(red[0]);
    out[0] = nrm;
    float c = 1.f;
    if (max_norm > 0.f) { c = max_norm / (nrm + 1e-6f); if (c > 1.f) c = 1.f; }
    out[1] = c;
  }
}

struct AdamHyper { float lr, b1, b2, eps, wd, bc1, bc2_sqrt; int zero_grad; };
template <typename MT> SLAM_DEVICE void load4(const MT* q, float* f);
template <> SLAM_DEVICE void load4<float>(const float* q, float* f) {
  const float4 t = *reinterpret_cast<const float4*>(q);
  f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
}
template <> SLAM_DEVICE void load4<bf16_t>(const bf16_t* q, float* f) {
  const uint2 t = *reinterpret_cast<const uint2*>(q);
  f[0] = __uint_as_float(t.x << 16); f[1] = __uint_as_float(t.x & 0xffff0000u);
  f[2] = __uint_as_float(t.y << 16); f[3] = __uint_as_float(t.y & 0xffff0000u);
}
SLAM_DEVICE void store4(float* q, const float* f) { *reinterpret_cast<float4*>(q) = make_float4(f[0], f[1], f[2], f[3]); }
SLAM_DEVICE void store4(bf16_t* q, const float* f) {
  uint2 o;
  o.x = pack_bf16x2(f[0], f[1]); o.y = pack_bf16x2(f[2], f[3]);
  *reinterpret_cast<uint2*>(q) = o;
}
// one element of torch.optim.AdamW (fp32 master: adamw_kernel's expression; bf16 state: the fused-kernel form with lerp)
template <bool MASTER>
SLAM_DEVICE void adam_elem(float& p, float& m, float& v, float g, const AdamHyper& h) {
  p *= (1.f - h.lr * h.wd);
  if (MASTER) m = h.b1 * m + (1.f - h.b1) * g;
  else m = m + (1.f - h.b1) * (g - m);
  v = h.b2 * v + (1.f - h.b2) * g * g;
  const float den = sqrtf(v) / h.bc2_sqrt + h.eps;
  p -= (h.lr / h.bc1) * (m / den);
}
// torch.optim.AdamW semantics on fp32 master weights; writes the bf16 working copy; optional
// grad zeroing. Traffic: 30 B/param (fp32 g,p,m,v read; p,m,v + bf16 written), 34 with zeroing.
// GT: storage type of the gradients (float, or bf16_t when the last backward kept its final values in bf16 only)
template <typename GT>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, bf16_t* __restrict__ pb,
                                                    GT* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n,
                                                    const float* __restrict__ clip, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2_sqrt,
                                                    int zero_grad) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const float cs = clip ? clip[1] : 1.f;
  float gl[4];
  load4<GT>(g + i, gl);
  const float4 gv = make_float4(gl[0], gl[1], gl[2], gl[3]);
  float4 pv = *reinterpret_cast<float4*>(p + i);
  float4 mv = *reinterpret_cast<float4*>(m + i);
  float4 vv = *reinterpret_cast<float4*>(v + i);
  float ga[4] = {gv.x * cs, gv.y * cs, gv.z * cs, gv.w * cs};
  float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
  const AdamHyper h = {lr, b1, b2, eps, wd, bc1, bc2_sqrt, zero_grad};
#pragma unroll
  for (int j = 0; j < 4; ++j) adam_elem<true>(pa[j], ma[j], va[j], ga[j], h);  // the tile kernel's expression: bit-identical updates
  *reinterpret_cast<float4*>(p + i) = make_float4(pa[0], pa[1], pa[2], pa[3]);
  *reinterpret_cast<float4*>(m + i) = make_float4(ma[0], ma[1], ma[2], ma[3]);
  *reinterpret_cast<float4*>(v + i) = make_float4(va[0], va[1], va[2], va[3]);
  uint2 o;
  o.x = pack_bf16x2(pa[0], pa[1]);
  o.y = pack_bf16x2(pa[2], pa[3]);
  *reinterpret_cast<uint2*>(pb + i) = o;
  if constexpr (sizeof(GT) == 4) {
    if (zero_grad) *reinterpret_cast<float4*>(g + i) = make_float4(0, 0, 0, 0);
  }
}

// The Slam recipe's optimizer precision (/root/reference config/model/slam.yaml:9 torch_dtype bfloat16 -> bf16 parameters
// and bf16 Adam moments under torch.optim.AdamW(fused=True)): state is STORED in bf16, every update is computed in fp32
// from the stored values and rounded once on the way back (torch's fused kernel: opmath fp32, exp_avg by lerp).
// No fp32 master copy. Traffic: fp32 g read (4) + bf16 p, m, v read and written (12) = 16 B/param.
template <typename GT>
__global__ __launch_bounds__(256) void adamw_bf16_kernel(bf16_t* __restrict__ p, GT* __restrict__ g,
                                                         bf16_t* __restrict__ m, bf16_t* __restrict__ v, size_t n,
                                                         const float* __restrict__ clip, float lr, float b1, float b2,
                                                         float eps, float wd, float bc1, float bc2_sqrt, int zero_grad) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= n) return;
  const float cs = clip ? clip[1] : 1.f;
  float ga[8];
  load4<GT>(g + i, ga);
  load4<GT>(g + i + 4, ga + 4);
#pragma unroll
  for (int j = 0; j < 8; ++j) ga[j] *= cs;
  float pa[8], ma[8], va[8];
  unpack_bf16x8(*reinterpret_cast<const uint4*>(p + i), pa);
  unpack_bf16x8(*reinterpret_cast<const uint4*>(m + i), ma);
  unpack_bf16x8(*reinterpret_cast<const uint4*>(v + i), va);
  const AdamHyper h = {lr, b1, b2, eps, wd, bc1, bc2_sqrt, zero_grad};
#pragma unroll
  for (int j = 0; j < 8; ++j) adam_elem<false>(pa[j], ma[j], va[j], ga[j], h);
  *reinterpret_cast<uint4*>(p + i) = pack_bf16x8(pa);
  *reinterpret_cast<uint4*>(m + i) = pack_bf16x8(ma);
  *reinterpret_cast<uint4*>(v + i) = pack_bf16x8(va);
  if constexpr (sizeof(GT) == 4) {
    if (zero_grad) {
      *reinterpret_cast<float4*>(g + i) = make_float4(0, 0, 0, 0);
      *reinterpret_cast<float4*>(g + i + 4) = make_float4(0, 0, 0, 0);
    }
  }
}

// ---- AdamW that also writes the TRANSPOSED bf16 weight image (round 3: replaces the separate transpose_bf16 pass after
// the optimizer: 4 B per matrix element of extra traffic and a kernel that ran at 2.4 TB/s). One block = one 64 x 64 tile
// of a [R][C] weight matrix (grid.z = same-shaped matrices at a constant stride: one per layer): every row segment of the
// tile is one contiguous 256 B (fp32) / 128 B (bf16) piece of each state array; the updated bf16 tile goes out row-major
// (pb) and, through a padded LDS tile, column-major (pt[C][R]). Per-element arithmetic is the flat kernels' own.
// MT = float / bf16_t: storage type of the Adam moments; MASTER: fp32 master weights (else the bf16 parameters ARE the state).
template <typename MT, bool MASTER, int TC, typename GT>  // tile = 64 rows x TC columns (TC = 64 or 128: 256 B or 512 B fp32 row segments)
__global__ __launch_bounds__(256) void adamw_tile_kernel(float* __restrict__ p, bf16_t* __restrict__ pb, bf16_t* __restrict__ pt,
                                                         GT* __restrict__ g, MT* __restrict__ m, MT* __restrict__ v, int R, int C,
                                                         size_t batch_stride, const float* __restrict__ clip, AdamHyper h) {
  constexpr int TPR = TC / 4, RPP = 256 / TPR, NP = 64 / RPP;  // threads per row, rows per pass, passes
  __shared__ uint16_t T[TC][66];  // transposed bf16 tile: T[col][row], rows padded to 132 B
  const size_t boff = (size_t)blockIdx.z * batch_stride;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * TC;
  const int tid = threadIdx.x, rr = tid / TPR, cc = (tid % TPR) * 4;
  const float cs = clip ? clip[1] : 1.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = rr + RPP * i;
    const size_t idx = boff + (size_t)(r0 + row) * C + c0 + cc;
    float ga[4], pa[4], ma[4], va[4];
    load4<GT>(g + idx, ga);
    if (MASTER) load4<float>(p + idx, pa);
    else load4<bf16_t>(pb + idx, pa);
    load4<MT>(m + idx, ma);
    load4<MT>(v + idx, va);
#pragma unroll
    for (int j = 0; j < 4; ++j) adam_elem<MASTER>(pa[j], ma[j], va[j], ga[j] * cs, h);
    if (MASTER) store4(p + idx, pa);
    store4(m + idx, ma);
    store4(v + idx, va);
    store4(pb + idx, pa);
    if constexpr (sizeof(GT) == 4) {
      if (h.zero_grad) *reinterpret_cast<float4*>(g + idx) = make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) T[cc + j][row] = (uint16_t)(pack_bf16x2(pa[j], 0.f) & 0xffffu);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < TC / 64; ++q) {
    const int orow = q * 64 + (tid >> 2), seg = (tid & 3) * 16;  // transposed row c0 + orow, its 16 elements r0 + seg ..
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&T[orow][seg]);
    uint4 a = make_uint4(src[0], src[1], src[2], src[3]), b = make_uint4(src[4], src[5], src[6], src[7]);
    bf16_t* dst = pt + boff + (size_t)(c0 + orow) * R + r0 + seg;
    *reinterpret_cast<uint4*>(dst) = a;
    *reinterpret_cast<uint4*>(dst + 8) = b;
  }
}
// The recipe's precision end to end (bf16 parameters, moments AND gradients: round 6) with 16-byte accesses: a thread owns 8
// consecutive columns (one dwordx4 per array and row instead of two dwordx2), 16 threads per 128-column row segment, 16 rows
// per pass. Per-element arithmetic = adam_elem<false>: the same bits as the kernel above.
__global__ __launch_bounds__(256) void adamw_tile_bf16x8_kernel(bf16_t* __restrict__ pb, bf16_t* __restrict__ pt, const bf16_t* __restrict__ g,
                                                                bf16_t* __restrict__ m, bf16_t* __restrict__ v, int R, int C,
                                                                size_t batch_stride, const float* __restrict__ clip, AdamHyper h) {
  constexpr int TC = 128, TPR = TC / 8, RPP = 256 / TPR, NP = 64 / RPP;
  __shared__ uint16_t T[TC][66];  // transposed bf16 tile: T[col][row], rows padded to 132 B
  const size_t boff = (size_t)blockIdx.z * batch_stride;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * TC;
  const int tid = threadIdx.x, rr = tid / TPR, cc = (tid % TPR) * 8;
  const float cs = clip ? clip[1] : 1.f;
  uint4 gq[NP], pq[NP], mq[NP], vq[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {  // every load of the tile in flight before the first use
    const size_t idx = boff + (size_t)(r0 + rr + RPP * i) * C + c0 + cc;
    gq[i] = *reinterpret_cast<const uint4*>(g + idx);
    pq[i] = *reinterpret_cast<const uint4*>(pb + idx);
    mq[i] = *reinterpret_cast<const uint4*>(m + idx);
    vq[i] = *reinterpret_cast<const uint4*>(v + idx);
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = rr + RPP * i;
    const size_t idx = boff + (size_t)(r0 + row) * C + c0 + cc;
    float ga[8], pa[8], ma[8], va[8];
    unpack_bf16x8(gq[i], ga);
    unpack_bf16x8(pq[i], pa);
    unpack_bf16x8(mq[i], ma);
    unpack_bf16x8(vq[i], va);
#pragma unroll
    for (int j = 0; j < 8; ++j) adam_elem<false>(pa[j], ma[j], va[j], ga[j] * cs, h);
    const uint4 po = pack_bf16x8(pa);
    *reinterpret_cast<uint4*>(pb + idx) = po;
    *reinterpret_cast<uint4*>(m + idx) = pack_bf16x8(ma);
    *reinterpret_cast<uint4*>(v + idx) = pack_bf16x8(va);
    const uint32_t w[4] = {po.x, po.y, po.z, po.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) T[cc + j][row] = (uint16_t)((j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xffffu));
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < TC / 64; ++q) {
    const int orow = q * 64 + (tid >> 2), seg = (tid & 3) * 16;  // transposed row c0 + orow, its 16 elements r0 + seg ..
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&T[orow][seg]);
    uint4 a = make_uint4(src[0], src[1], src[2], src[3]), b = make_uint4(src[4], src[5], src[6], src[7]);
    bf16_t* dst = pt + boff + (size_t)(c0 + orow) * R + r0 + seg;
    *reinterpret_cast<uint4*>(dst) = a;
    *reinterpret_cast<uint4*>(dst + 8) = b;
  }
}
// the vectors between the matrices (norm weights, biases): count elements at a constant stride, grid.y = instances
template <typename MT, bool MASTER, typename GT>
__global__ __launch_bounds__(256) void adamw_strided_kernel(float* __restrict__ p, bf16_t* __restrict__ pb, GT* __restrict__ g,
                                                            MT* __restrict__ m, MT* __restrict__ v, size_t n, size_t stride,
                                                            const float* __restrict__ clip, AdamHyper h) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const size_t idx = (size_t)blockIdx.y * stride + i;
  const float cs = clip ? clip[1] : 1.f;
  float ga[4], pa[4], ma[4], va[4];
  load4<GT>(g + idx, ga);
  if (MASTER) load4<float>(p + idx, pa);
  else load4<bf16_t>(pb + idx, pa);
  load4<MT>(m + idx, ma);
  load4<MT>(v + idx, va);
#pragma unroll
  for (int j = 0; j < 4; ++j) adam_elem<MASTER>(pa[j], ma[j], va[j], ga[j] * cs, h);
  if (MASTER) store4(p + idx, pa);
  store4(m + idx, ma);
  store4(v + idx, va);
  store4(pb + idx, pa);
  if constexpr (sizeof(GT) == 4) {
    if (h.zero_grad) *reinterpret_cast<float4*>(g + idx) = make_float4(0, 0, 0, 0);
  }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, size_t n) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 v = *reinterpret_cast<const float4*>(s + i);
  uint2 o;
  o.x = pack_bf16x2(v.x, v.y);
  o.y = pack_bf16x2(v.z, v.w);
  *reinterpret_cast<uint2*>(d + i) = o;
}

// the same conversion over blocks of GRAD_CHUNK elements, each emitting the sum of squares of the ROUNDED values it stored
// (GradSink slot = block index): the image of a gradient tensor that had to be built in fp32 (embedding scatter)
__global__ __launch_bounds__(256) void f32_to_bf16_sumsq_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, size_t n,
                                                                float* __restrict__ sumsq) {
  __shared__ float red[4];
  const size_t lo = (size_t)blockIdx.x * GRAD_CHUNK;
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < GRAD_CHUNK / 1024; ++it) {
    const size_t i = lo + (size_t)(it * 256 + threadIdx.x) * 4;
    if (i < n) {
      const float4 v = *reinterpret_cast<const float4*>(s + i);
      const uint2 o = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
      *reinterpret_cast<uint2*>(d + i) = o;
      ss += sq_bf16x2(o.x) + sq_bf16x2(o.y);
    }
  }
  block_sum_store<4>(ss, red, sumsq + blockIdx.x);
}

__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ s, float* __restrict__ d, size_t n) {
  size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const uint2 v = *reinterpret_cast<const uint2*>(s + i);
  *reinterpret_cast<float4*>(d + i) = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                                                  __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}

// dst[C][R] = src[R][C]^T, bf16, 64x64 tiles through LDS (R, C multiples of 64)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                             int R, int C, size_t batch_stride) {
  __shared__ uint32_t t[64][33];  // 64 rows x 64 bf16 (+1 dword pad)
  src += (size_t)blockIdx.z * batch_stride;  // same-shaped matrices at a constant stride (one per layer)
  dst += (size_t)blockIdx.z * batch_stride;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row = (tid >> 3) + 32 * i, ch = tid & 7;
    uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)(r0 + row) * C + c0 + ch * 8);
    t[row][ch * 4 + 0] = v.x; t[row][ch * 4 + 1] = v.y; t[row][ch * 4 + 2] = v.z; t[row][ch * 4 + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int col = (tid >> 3) + 32 * i, rb = tid & 7;  // output row = col, 8 source rows rb*8..
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t a = t[rb * 8 + 2 * k][col >> 1], b = t[rb * 8 + 2 * k + 1][col >> 1];
      uint32_t lo = (col & 1) ? (a >> 16) : (a & 0xffffu);
      uint32_t hi = (col & 1) ? (b >> 16) : (b & 0xffffu);
      w[k] = lo | (hi << 16);
    }
    *reinterpret_cast<uint4*>(dst + (size_t)(c0 + col) * R + r0 + rb * 8) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

inline unsigned nblocks(size_t n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

namespace slam {

#define LAUNCH_RET() return (int)hipGetLastError()

int rmsnorm_fwd(const bf16_t* x, const bf16_t* w, bf16_t* y, float* rstd, int M, int H, float eps, hipStream_t st) {
  if ((H & 7) || H > MAXC_LIMIT * 512) return -1;
  switch ((H / 8 + 63) / 64) {
    case 1: rmsnorm_fwd_kernel<1><<<(M + 3) / 4, 256, 0, st>>>(x, w, y, rstd, M, H, eps); break;
    case 2: rmsnorm_fwd_kernel<2><<<(M + 3) / 4, 256, 0, st>>>(x, w, y, rstd, M, H, eps); break;
    case 3: rmsnorm_fwd_kernel<3><<<(M + 3) / 4, 256, 0, st>>>(x, w, y, rstd, M, H, eps); break;
    default: rmsnorm_fwd_kernel<4><<<(M + 3) / 4, 256, 0, st>>>(x, w, y, rstd, M, H, eps); break;
  }
  LAUNCH_RET();
}

int rmsnorm_bwd_blocks(int M) { int b = (M + 15) / 16; return b > 512 ? 512 : (b < 1 ? 1 : b); }  // 2 blocks/CU

// sink (nullable, with dw): how the final values of dw are kept (GradSink, kernels.h)
int rmsnorm_bwd(const bf16_t* dy, const bf16_t* x, const bf16_t* w, const float* rstd, const bf16_t* dres,
                bf16_t* dx, float* dw, int accumulate, float* part, int M, int H, hipStream_t st, bf16_t* dw_img, GradSink* sink) {
  if ((H & 7) || H > MAXC_LIMIT * 512) return -1;
  int nb = rmsnorm_bwd_blocks(M);
  switch ((H / 8 + 63) / 64) {
    case 1: rmsnorm_bwd_kernel<1><<<nb, 256, 0, st>>>(dy, x, w, rstd, dres, dx, part, M, H); break;
    case 2: rmsnorm_bwd_kernel<2><<<nb, 256, 0, st>>>(dy, x, w, rstd, dres, dx, part, M, H); break;
    case 3: rmsnorm_bwd_kernel<3><<<nb, 256, 0, st>>>(dy, x, w, rstd, dres, dx, part, M, H); break;
    default: rmsnorm_bwd_kernel<4><<<nb, 256, 0, st>>>(dy, x, w, rstd, dres, dx, part, M, H); break;
  }
  if (dw) return colsum_finish_many(part, 0, nb, H, dw, 0, 1, accumulate, st, dw_img, sink);  // dw == null: caller finishes later
  LAUNCH_RET();
}
// finish `count` equally shaped partial slabs in one launch: out[i] (+)= column sums of part[i]
int colsum_finish_many(const float* part, size_t part_stride, int nb, int N, float* out, size_t out_stride, int count,
                       int accumulate, hipStream_t st, bf16_t* img, GradSink* sink) {
  if (count <= 0) return 0;
  const dim3 grid((N + 15) / 16, count);
  if (sink) {
    sink->used = (int)(grid.x * grid.y);
    if (sink->img_only && !img) return -1;
    if (sink->sumsq && sink->used > sink->cap) return -3;
  }
  colsum_finish_kernel<<<grid, 256, 0, st>>>(part, nb, N, out, accumulate, part_stride, out_stride, img, sink ? sink->img_only : 0,
                                             sink ? sink->sumsq : nullptr);
  LAUNCH_RET();
}

int colsum_blocks(int M) { int b = (M + 63) / 64; return b > 128 ? 128 : (b < 1 ? 1 : b); }

int colsum_bf16(const bf16_t* X, int ld, int M, int N, float* out, int accumulate, float* part, hipStream_t st) {
  if (N & 7) return -1;
  int nb = colsum_blocks(M);
  dim3 grid((N / 8 + 15) / 16, nb);
  colsum_bf16_kernel<<<grid, 256, 0, st>>>(X, ld, M, N, part);
  if (out) colsum_finish_kernel<<<(N + 15) / 16, 256, 0, st>>>(part, nb, N, out, accumulate, 0, 0, nullptr, 0, nullptr);
  LAUNCH_RET();
}

int rope_table(const int64_t* pos, int M, int T, int head_dim, float theta, float* cs, float* sn, float* csq, float* snq,
               float qscale, hipStream_t st) {
  int half = head_dim / 2;
  rope_table_kernel<<<nblocks((size_t)M * half, 256), 256, 0, st>>>(pos, M, T, half, theta, cs, sn, csq, snq, qscale);
  LAUNCH_RET();
}

int rope_apply(bf16_t* qkv, int ld, int M, int nrot_heads, int head_dim, const float* cs, const float* sn, int backward,
               hipStream_t st, int q_heads, float q_scale) {
  if (head_dim & 15) return -1;
  rope_kernel<<<nblocks((size_t)M * nrot_heads * (head_dim / 16), 256), 256, 0, st>>>(qkv, ld, M, nrot_heads, head_dim, cs, sn,
                                                                                     backward ? -1.f : 1.f, q_heads, q_scale);
  LAUNCH_RET();
}

int swiglu_fwd(const bf16_t* gu, bf16_t* act, int M, int I, int blk, hipStream_t st) {
  if ((I & 7) || (blk & 7) || I % blk) return -1;
  swiglu_fwd_kernel<<<nblocks((size_t)M * (I / 8), 256), 256, 0, st>>>(gu, act, (size_t)M, I, blk);
  LAUNCH_RET();
}
int swiglu_bwd(bf16_t* gu, const bf16_t* dact, int M, int I, int blk, hipStream_t st) {
  if ((I & 7) || (blk & 7) || I % blk) return -1;
  swiglu_bwd_kernel<<<nblocks((size_t)M * (I / 8), 256), 256, 0, st>>>(gu, dact, (size_t)M, I, blk);
  LAUNCH_RET();
}

int embed_fwd(const int64_t* ids, const bf16_t* E, bf16_t* out, int M, int H, int V, hipStream_t st) {
  embed_fwd_kernel<<<nblocks((size_t)M * (H / 8), 256), 256, 0, st>>>(ids, E, out, (size_t)M, H, V);
  LAUNCH_RET();
}
int onehot(const int64_t* ids, bf16_t* oh, int M, int Vp, int V, int pad_id, hipStream_t st) {
  onehot_kernel<<<nblocks((size_t)M * (Vp / 8), 256), 256, 0, st>>>(ids, oh, (size_t)M, Vp, V, pad_id);
  LAUNCH_RET();
}

size_t embed_bwd_workspace_ints(int M, int Vp) { return (size_t)2 * M + 2 * (size_t)Vp + 64; }
int embed_bwd(const int64_t* ids, const bf16_t* dh, float* dE, int M, int H, int Vp, int V, int pad_id, int* ws,
              hipStream_t st) {
  if (H & 7) return -1;
  int* rank = ws;
  int* list = ws + M;
  int* count = ws + 2 * (size_t)M;
  int* offset = count + Vp;
  hipError_t e = hipMemsetAsync(count, 0, (size_t)Vp * sizeof(int), st);
  if (e != hipSuccess) return (int)e;
  embed_rank_kernel<<<(M + 255) / 256, 256, 0, st>>>(ids, M, V, pad_id, rank, count);
  embed_scan_kernel<<<1, 1024, 0, st>>>(count, offset, V);
  embed_fill_kernel<<<(M + 255) / 256, 256, 0, st>>>(ids, M, V, pad_id, rank, offset, list);
  embed_scatter_kernel<<<V, 256, 0, st>>>(dh, dE, H, count, offset, list);
  LAUNCH_RET();
}

int cross_entropy(const bf16_t* logits, const int64_t* labels, double num_items, bf16_t* dlogits, float* row_loss,
                  float* denom, float* loss, int B, int T, int Vp, int V, const uint8_t* colmask, hipStream_t st) {
  if (Vp < 512 || (Vp & 7) || V > Vp) return -1;
  int M = B * T;
  count_valid_kernel<<<1, 256, 0, st>>>(labels, B, T, num_items, denom);
  if (Vp == 512) ce_kernel<<<(M + 3) / 4, 256, 0, st>>>(logits, labels, denom, dlogits, row_loss, B, T, Vp, V, colmask);
  else ce_big_kernel<<<M, 256, 0, st>>>(logits, labels, denom, dlogits, row_loss, B, T, Vp, V, colmask);
  loss_finish_kernel<<<1, 256, 0, st>>>(row_loss, M, denom, loss);
  LAUNCH_RET();
}
int seq_loglik(const float* row_loss, const int64_t* labels, int B, int T, float* ll, float* cnt, hipStream_t st) {
  seq_loglik_kernel<<<B, 256, 0, st>>>(row_loss, labels, B, T, ll, cnt);
  LAUNCH_RET();
}
int copy_cols(const bf16_t* src, int lds_, bf16_t* dst, int ldd, int M, int ncols, hipStream_t st) {
  copy_cols_kernel<<<nblocks((size_t)M * ncols, 256), 256, 0, st>>>(src, lds_, dst, ldd, (size_t)M, ncols);
  LAUNCH_RET();
}
int scale_rows_bf16(bf16_t* x, const float* coef, int M, int T, int ncols, hipStream_t st) {
  if (ncols & 7) return -1;
  scale_rows_bf16_kernel<<<nblocks((size_t)M * (ncols / 8), 256), 256, 0, st>>>(x, coef, (size_t)M, T, ncols / 8);
  LAUNCH_RET();
}
int scale_bf16(bf16_t* x, size_t n, float s, hipStream_t st) {
  if (n & 7) return -1;
  scale_bf16_kernel<<<nblocks(n / 8, 256), 256, 0, st>>>(x, n / 8, s);
  LAUNCH_RET();
}

int grad_chunk_elems() { return GRAD_CHUNK; }
// chunk sums of the chunk-aligned range [off, off + cnt) (cnt may end at n instead of a chunk boundary); g_bf16: the
// gradients are bf16_t at g, not float
int grad_sumsq_chunks(const void* g, int g_bf16, size_t n, size_t off, size_t cnt, float* chunk_sums, hipStream_t st) {
  if ((n & 3) || (off % GRAD_CHUNK) || off + cnt > n || (((off + cnt) % GRAD_CHUNK) && off + cnt != n)) return -1;
  if (cnt == 0) return 0;
  const size_t nc = (cnt + GRAD_CHUNK - 1) / GRAD_CHUNK;
  const unsigned nb = (unsigned)((nc + CHUNKS_PER_BLOCK - 1) / CHUNKS_PER_BLOCK);
  if (g_bf16) sumsq_chunks_kernel<bf16_t><<<nb, 256, 0, st>>>((const bf16_t*)g, n, off / GRAD_CHUNK, nc, chunk_sums);
  else sumsq_chunks_kernel<float><<<nb, 256, 0, st>>>((const float*)g, n, off / GRAD_CHUNK, nc, chunk_sums);
  LAUNCH_RET();
}
int grad_norm_from_chunks(const float* chunk_sums, size_t n_chunks, float max_norm, float* out, hipStream_t st) {
  norm_finish_kernel<<<1, 1024, 0, st>>>(chunk_sums, (int)n_chunks, max_norm, out);
  LAUNCH_RET();
}
int grad_norm(const float* g, size_t n, float max_norm, float* part, float* out, hipStream_t st) {
  if (int r = grad_sumsq_chunks(g, 0, n, 0, n, part, st)) return r;
  return grad_norm_from_chunks(part, (n + GRAD_CHUNK - 1) / GRAD_CHUNK, max_norm, out, st);
}
int adamw(float* p, bf16_t* pb, void* g, int g_bf16, float* m, float* v, size_t n, const float* clip, double lr, double b1,
          double b2, double eps, double wd, int step, int zero_grad, hipStream_t st) {
  if (n & 3) return -1;
  // bias corrections in double like torch.optim.AdamW (python floats), then fp32 in the kernel
  float bc1 = (float)(1.0 - pow(b1, (double)step));
  float bc2s = (float)sqrt(1.0 - pow(b2, (double)step));
  if (g_bf16)
    adamw_kernel<bf16_t><<<nblocks(n / 4, 256), 256, 0, st>>>(p, pb, (bf16_t*)g, m, v, n, clip, (float)lr, (float)b1, (float)b2,
                                                              (float)eps, (float)wd, bc1, bc2s, zero_grad);
  else
    adamw_kernel<float><<<nblocks(n / 4, 256), 256, 0, st>>>(p, pb, (float*)g, m, v, n, clip, (float)lr, (float)b1, (float)b2,
                                                             (float)eps, (float)wd, bc1, bc2s, zero_grad);
  LAUNCH_RET();
}
int adamw_bf16(bf16_t* p, void* g, int g_bf16, bf16_t* m, bf16_t* v, size_t n, const float* clip, double lr, double b1, double b2,
               double eps, double wd, int step, int zero_grad, hipStream_t st) {
  if (n & 7) return -1;
  float bc1 = (float)(1.0 - pow(b1, (double)step));
  float bc2s = (float)sqrt(1.0 - pow(b2, (double)step));
  if (g_bf16)
    adamw_bf16_kernel<bf16_t><<<nblocks(n / 8, 256), 256, 0, st>>>(p, (bf16_t*)g, m, v, n, clip, (float)lr, (float)b1, (float)b2,
                                                                   (float)eps, (float)wd, bc1, bc2s, zero_grad);
  else
    adamw_bf16_kernel<float><<<nblocks(n / 8, 256), 256, 0, st>>>(p, (float*)g, m, v, n, clip, (float)lr, (float)b1, (float)b2,
                                                                  (float)eps, (float)wd, bc1, bc2s, zero_grad);
  LAUNCH_RET();
}
static AdamHyper adam_hyper(double lr, double b1, double b2, double eps, double wd, int step, int zero_grad) {
  // bias corrections in double like torch.optim.AdamW (python floats), then fp32 in the kernel
  AdamHyper h;
  h.lr = (float)lr; h.b1 = (float)b1; h.b2 = (float)b2; h.eps = (float)eps; h.wd = (float)wd;
  h.bc1 = (float)(1.0 - pow(b1, (double)step));
  h.bc2_sqrt = (float)sqrt(1.0 - pow(b2, (double)step));
  h.zero_grad = zero_grad;
  return h;
}
template <int TC, typename GT>
static void adamw_tiles_launch(int mode, dim3 grid, float* p, bf16_t* pb, bf16_t* pt, GT* g, void* m, void* v, int R, int C,
                               size_t batch_stride, const float* clip, const AdamHyper& h, hipStream_t st) {
  if (mode == 0) adamw_tile_kernel<float, true, TC, GT><<<grid, 256, 0, st>>>(p, pb, pt, g, (float*)m, (float*)v, R, C, batch_stride, clip, h);
  else if (mode == 1) adamw_tile_kernel<bf16_t, true, TC, GT><<<grid, 256, 0, st>>>(p, pb, pt, g, (bf16_t*)m, (bf16_t*)v, R, C, batch_stride, clip, h);
  else adamw_tile_kernel<bf16_t, false, TC, GT><<<grid, 256, 0, st>>>(p, pb, pt, g, (bf16_t*)m, (bf16_t*)v, R, C, batch_stride, clip, h);
}
// mode 0: fp32 master + fp32 moments; 1: fp32 master + bf16 moments; 2: bf16 parameters + bf16 moments (no master)
int adamw_tiles(int mode, float* p, bf16_t* pb, bf16_t* pt, void* g, int g_bf16, void* m, void* v, int R, int C, int batch,
                size_t batch_stride, const float* clip, double lr, double b1, double b2, double eps, double wd, int step, int zero_grad,
                hipStream_t st) {
  if ((R & 63) || (C & 63) || batch < 1 || mode < 0 || mode > 2) return -1;
  const AdamHyper h = adam_hyper(lr, b1, b2, eps, wd, step, zero_grad);
  static int tile_cols = 128;  // SLAM_ADAMW_TILE_COLS=64: 256-byte row segments (A/B knob)
  static bool read_env = false;
  if (!read_env) { const char* e = getenv("SLAM_ADAMW_TILE_COLS"); if (e && atoi(e) == 64) tile_cols = 64; read_env = true; }
  static int x8 = -1;  // SLAM_ADAMW_X8=0: the 8-byte-access kernel for the all-bf16 case as well (A/B knob)
  if (x8 < 0) { const char* e = getenv("SLAM_ADAMW_X8"); x8 = !(e && e[0] == '0'); }
  if (tile_cols == 128 && (C % 128 == 0) && mode == 2 && g_bf16 && x8) {
    adamw_tile_bf16x8_kernel<<<dim3(C / 128, R / 64, batch), 256, 0, st>>>(pb, pt, (const bf16_t*)g, (bf16_t*)m, (bf16_t*)v, R, C, batch_stride, clip, h);
  } else if (tile_cols == 128 && (C % 128 == 0)) {
    const dim3 grid(C / 128, R / 64, batch);
    if (g_bf16) adamw_tiles_launch<128, bf16_t>(mode, grid, p, pb, pt, (bf16_t*)g, m, v, R, C, batch_stride, clip, h, st);
    else adamw_tiles_launch<128, float>(mode, grid, p, pb, pt, (float*)g, m, v, R, C, batch_stride, clip, h, st);
  } else {
    const dim3 grid(C / 64, R / 64, batch);
    if (g_bf16) adamw_tiles_launch<64, bf16_t>(mode, grid, p, pb, pt, (bf16_t*)g, m, v, R, C, batch_stride, clip, h, st);
    else adamw_tiles_launch<64, float>(mode, grid, p, pb, pt, (float*)g, m, v, R, C, batch_stride, clip, h, st);
  }
  LAUNCH_RET();
}
template <typename GT>
static void adamw_strided_launch(int mode, dim3 grid, float* p, bf16_t* pb, GT* g, void* m, void* v, size_t n, size_t stride,
                                 const float* clip, const AdamHyper& h, hipStream_t st) {
  if (mode == 0) adamw_strided_kernel<float, true, GT><<<grid, 256, 0, st>>>(p, pb, g, (float*)m, (float*)v, n, stride, clip, h);
  else if (mode == 1) adamw_strided_kernel<bf16_t, true, GT><<<grid, 256, 0, st>>>(p, pb, g, (bf16_t*)m, (bf16_t*)v, n, stride, clip, h);
  else adamw_strided_kernel<bf16_t, false, GT><<<grid, 256, 0, st>>>(p, pb, g, (bf16_t*)m, (bf16_t*)v, n, stride, clip, h);
}
int adamw_strided(int mode, float* p, bf16_t* pb, void* g, int g_bf16, void* m, void* v, size_t n, int batch, size_t stride,
                  const float* clip, double lr, double b1, double b2, double eps, double wd, int step, int zero_grad, hipStream_t st) {
  if ((n & 3) || batch < 1 || mode < 0 || mode > 2) return -1;
  if (n == 0) return 0;
  const AdamHyper h = adam_hyper(lr, b1, b2, eps, wd, step, zero_grad);
  const dim3 grid(nblocks(n / 4, 256), batch);
  if (g_bf16) adamw_strided_launch<bf16_t>(mode, grid, p, pb, (bf16_t*)g, m, v, n, stride, clip, h, st);
  else adamw_strided_launch<float>(mode, grid, p, pb, (float*)g, m, v, n, stride, clip, h, st);
  LAUNCH_RET();
}
int transpose_bf16(const bf16_t* src, bf16_t* dst, int R, int C, int batch, size_t batch_stride, hipStream_t st) {
  if ((R & 63) || (C & 63) || batch < 1) return -1;
  transpose_bf16_kernel<<<dim3(C / 64, R / 64, batch), 256, 0, st>>>(src, dst, R, C, batch_stride);
  LAUNCH_RET();
}
int bf16_to_f32(const bf16_t* s, float* d, size_t n, hipStream_t st) {
  if (n & 3) return -1;
  bf16_to_f32_kernel<<<nblocks(n / 4, 256), 256, 0, st>>>(s, d, n);
  LAUNCH_RET();
}
int f32_to_bf16_sumsq_slots(size_t n) { return (int)((n + GRAD_CHUNK - 1) / GRAD_CHUNK); }
int f32_to_bf16_sumsq(const float* s, bf16_t* d, size_t n, float* sumsq, hipStream_t st) {
  if (n & 3) return -1;
  f32_to_bf16_sumsq_kernel<<<(unsigned)f32_to_bf16_sumsq_slots(n), 256, 0, st>>>(s, d, n, sumsq);
  LAUNCH_RET();
}
int f32_to_bf16(const float* s, bf16_t* d, size_t n, hipStream_t st) {
  if (n & 3) return -1;
  f32_to_bf16_kernel<<<nblocks(n / 4, 256), 256, 0, st>>>(s, d, n);
  LAUNCH_RET();
}

}  // namespace slam
