// HBM-bound pointwise / reduction kernels of the vocoder path (gfx950).
// Reference call sites are listed next to each entry point in include/hsp.h.
#include "hsp_device.h"

namespace {

inline unsigned grid_for(int64_t n, int threads, int64_t cap = 1048576) {
  int64_t b = (n + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

// ---------------------------------------------------------------- Activation1d
// Stand-alone anti-aliased SnakeBeta (used where no conv follows, and -- measured faster on
// MI355X for wide layers -- in front of a plain conv instead of the fused prologue, because
// VALU work inside the conv kernel is paid in fp32-MFMA time on the same SIMD).
// One workgroup = ACT_TILE outputs of one (b, c) row.  The raw window (aligned float4 loads),
// the 2x-rate snake signal and the result go through LDS, so HBM sees one read and one write
// of the tensor (the eager reference makes ~38 passes).
constexpr int ACT_TILE = 1024;
constexpr int ACT_THREADS = 256;
constexpr int ACT_HALO = 8;  // raw window starts at p0 - 8 (16-B aligned); 5 are needed

typedef float act_f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(ACT_THREADS) void act1d_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                            int L, const float* __restrict__ alpha_exp,
                                                            const float* __restrict__ beta_inv,
                                                            const float* __restrict__ filt, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float raw[ACT_TILE + 2 * ACT_HALO];
  __shared__ __attribute__((aligned(16))) float a2[2 * ACT_TILE + 16];
  const int tile = blockIdx.x % n_tiles;
  const int row = blockIdx.x / n_tiles;  // b * C + c
  const int c = row % C;
  const int p0 = tile * ACT_TILE;
  const float* xrow = x + (int64_t)row * L;
  float* yrow = y + (int64_t)row * L;
  const float kf = alpha_exp[c] * 0.318309886183790672f, kb = 0.5f * beta_inv[c];
  const int n_out = min(ACT_TILE, L - p0);
  const int tid = threadIdx.x;
  const bool vec = ((L & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  // ---- raw[s] = x[clamp(p0 - 8 + s)], s in [0, n_out + 16)
  const int nraw = n_out + 2 * ACT_HALO;
  if (vec && p0 >= ACT_HALO && p0 + n_out + ACT_HALO <= L) {
    for (int v = tid; v < nraw / 4; v += ACT_THREADS)
      *reinterpret_cast<float4*>(raw + 4 * v) = *reinterpret_cast<const float4*>(xrow + p0 - ACT_HALO + 4 * v);
  } else {
    for (int s = tid; s < nraw; s += ACT_THREADS) raw[s] = xrow[hsp_clampi(p0 - ACT_HALO + s, 0, L - 1)];
  }
  __syncthreads();
  // ---- a2[s] = snake(2 * up[clamp(2*p0 - 5 + s)]), s in [0, 2*n_out + 10)
  const int a2w = 2 * n_out + 10;
  const int mlo = 2 * p0 - 5;
  if (mlo >= 0 && mlo + a2w <= 2 * L) {
    // interior: slot 2i+1 <-> m = 2q (q = p0-2+i), slot 2i+2 <-> m = 2q+1; x[q-3 .. q+3] = raw[i+3 .. i+9].
    // Two adjacent pairs per thread share their raw window (8 LDS reads for 4 outputs).
    const int npair = (a2w - 1) >> 1;
    float hu[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) hu[t] = filt[t];
    for (int i = 2 * tid; i < npair; i += 2 * ACT_THREADS) {
      float xv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) xv[t] = raw[i + 3 + t];
      act_f32x2 u0 = {0.0f, 0.0f}, u1 = {0.0f, 0.0f};
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const act_f32x2 hh = {hu[11 - 2 * t], hu[10 - 2 * t]};
        const act_f32x2 x0 = {xv[t], xv[t + 1]};
        const act_f32x2 x1 = {xv[t + 1], xv[t + 2]};
        u0 = __builtin_elementwise_fma(x0, hh, u0);
        u1 = __builtin_elementwise_fma(x1, hh, u1);
      }
      u0 = u0 * 2.0f;
      u1 = u1 * 2.0f;
      a2[2 * i + 1] = hsp_snake_hw(u0.x, kf, kb);
      a2[2 * i + 2] = hsp_snake_hw(u0.y, kf, kb);
      if (i + 1 < npair) {
        a2[2 * i + 3] = hsp_snake_hw(u1.x, kf, kb);
        a2[2 * i + 4] = hsp_snake_hw(u1.y, kf, kb);
      }
    }
    if (tid < 2) {  // slot 0 (odd m, raw[3..8]) and the last slot (even m)
      const float* xr_ = tid ? raw + n_out + 7 : raw + 3;
      float uu = 0.0f;
#pragma unroll
      for (int t = 0; t < 6; ++t) uu = fmaf(xr_[t], tid ? filt[11 - 2 * t] : filt[10 - 2 * t], uu);
      a2[tid ? a2w - 1 : 0] = hsp_snake_hw(2.0f * uu, kf, kb);
    }
  } else {
    for (int s = tid; s < a2w; s += ACT_THREADS) {
      const int m = hsp_clampi(mlo + s, 0, 2 * L - 1);
      const int q = m >> 1, odd = m & 1;
      const float* xr_ = raw + (q - 3 + odd) - (p0 - ACT_HALO);
      float u = 0.0f;
#pragma unroll
      for (int t = 0; t < 6; ++t) u = fmaf(xr_[t], odd ? filt[10 - 2 * t] : filt[11 - 2 * t], u);
      a2[s] = hsp_snake_hw(2.0f * u, kf, kb);
    }
  }
  __syncthreads();
  // ---- y[p0 + s] = sum_k hd[k] * a2[2s + k]
  float hd[12];
#pragma unroll
  for (int t = 0; t < 12; ++t) hd[t] = filt[12 + t];
  if (vec && (n_out & 3) == 0) {
    // four consecutive outputs per thread: their 12-tap windows overlap in 18 a2 values
    for (int v = tid; v < n_out / 4; v += ACT_THREADS) {
      const act_f32x2* ar = reinterpret_cast<const act_f32x2*>(a2 + 8 * v);
      act_f32x2 w[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) w[k] = ar[k];
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        act_f32x2 acc = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const act_f32x2 hh = {hd[2 * k], hd[2 * k + 1]};
          acc = __builtin_elementwise_fma(w[q + k], hh, acc);
        }
        o[q] = acc.x + acc.y;
      }
      *reinterpret_cast<float4*>(yrow + p0 + 4 * v) = make_float4(o[0], o[1], o[2], o[3]);
    }
  } else {
    for (int s = tid; s < n_out; s += ACT_THREADS) {
      const act_f32x2* ar = reinterpret_cast<const act_f32x2*>(a2 + 2 * s);
      act_f32x2 acc = {0.0f, 0.0f};
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const act_f32x2 hh = {hd[2 * k], hd[2 * k + 1]};
        acc = __builtin_elementwise_fma(ar[k], hh, acc);
      }
      yrow[p0 + s] = acc.x + acc.y;
    }
  }
}

// Wave-per-segment form for 16-B addressable rows with L % 4 == 0 (every layer of the vocoder).
// One WAVE owns ACT2_SEG outputs of one (b, c) row and its own LDS slice: no workgroup barrier,
// a wave's LDS instructions retire in order, so phases are separated by compiler fences only.
// Every LDS access is contiguous across lanes (b64 / b128, lane stride = access width), so there
// are no bank conflicts -- the workgroup-tile kernel above reads with lane strides of 2, 4 and 8
// floats and is LDS-conflict bound at 1.7 TB/s.
//   raw[j]  = x[clamp(p0 - 8 + j)]                       j in [0, 512)
//   a2[j]   = a[clamp(2 p0 - 5 + j, 0, 2L - 1)]          j in [0, 1024): slot 4l of lane l is the
//             first tap of outputs (p0 + 2l, p0 + 2l + 1) -> three b128 + one b64 read per pair.
//   phase B: lane pair pi owns a[2q+1 .. 2q+4], q = p0 - 3 + 2 pi (odd(q), even(q+1), odd(q+1),
//            even(q+2)): one b128 write at slot 4 pi; its inputs x[q-2 .. q+4] sit in four aligned
//            b64 reads.  The x2 gain is folded into the taps (exact).
constexpr int ACT2_SEG = 496;   // raw window = SEG + 16 = 512 floats = two float4 per lane
constexpr int ACT2_WAVES = 4;
constexpr int ACT2_ITEMS = 4;   // consecutive segments per wave (software-pipelined loads)
constexpr int ACT2_RAW = 512;
constexpr int ACT2_A2 = 1040;

__device__ __forceinline__ void act2_compiler_fence() { asm volatile("" ::: "memory"); }

// one work item = one segment [p0, p0 + n_out) of one (b, c) row; p0 - 8 and L are multiples of 4, so a
// float4 of the raw window lies wholly inside the row or wholly outside it
struct Act2Item {
  int p0, n_out, c;
  int64_t row_off;
};

__device__ __forceinline__ Act2Item act2_item(unsigned w, int nseg, int C, int L) {
  const unsigned row = w / (unsigned)nseg;  // b * C + c
  const int seg = (int)(w - row * (unsigned)nseg);
  Act2Item it;
  it.p0 = seg * ACT2_SEG;
  it.n_out = min(ACT2_SEG, L - it.p0);  // multiple of 4
  it.c = (int)(row % (unsigned)C);
  it.row_off = (int64_t)row * L;
  return it;
}

// Two unconditional float4 loads per lane (clamped address, so always legal and branch-free: a
// load inside a divergent block gets an s_waitcnt vmcnt(0) right behind it and the round trips
// serialise); the replicate padding is applied when the registers are written to LDS.
__device__ __forceinline__ void act2_load(const float* __restrict__ x, const Act2Item& it, int L, int lane,
                                          float4 (&rv)[2]) {
  const float* xrow = x + it.row_off;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = it.p0 - 8 + 4 * (lane + 64 * k);
    rv[k] = *reinterpret_cast<const float4*>(xrow + hsp_clampi(idx, 0, L - 4));
  }
}

__global__ __launch_bounds__(64 * ACT2_WAVES) void act1d_seg_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                    int C, int L, const float* __restrict__ alpha_exp,
                                                                    const float* __restrict__ beta_inv,
                                                                    const float* __restrict__ filt, int nseg,
                                                                    unsigned nwork) {
  __shared__ __attribute__((aligned(16))) float raw_all[ACT2_WAVES][ACT2_RAW];
  __shared__ __attribute__((aligned(16))) float a2_all[ACT2_WAVES][ACT2_A2];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const unsigned w0 = (blockIdx.x * ACT2_WAVES + wave) * ACT2_ITEMS;
  if (w0 >= nwork) return;  // wave-uniform; the kernel has no workgroup barrier
  const unsigned w1 = min(w0 + ACT2_ITEMS, nwork);
  float* raw = raw_all[wave];
  float* a2 = a2_all[wave];
  act_f32x2 hu[6], hd[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    hu[i] = act_f32x2{2.0f * filt[10 - 2 * i], 2.0f * filt[11 - 2 * i]};
    hd[i] = act_f32x2{filt[12 + 2 * i], filt[13 + 2 * i]};
  }

  // A wave walks ACT2_ITEMS consecutive segments; the raw window of the next one is in flight
  // (registers) while the current one is computed, so HBM latency is off the critical path.
  // The outputs of an item stay in registers and are stored one iteration later, BEFORE the next
  // prefetch is issued: vmcnt retires in order, so the wait for a prefetch would otherwise also wait
  // for the stores issued just before it (a full write round trip per item).
  Act2Item cur = act2_item(w0, nseg, C, L);
  float4 rv[2];
  act2_load(x, cur, L, lane, rv);
  constexpr int NC = (ACT2_SEG / 2 + 63) / 64;  // phase-C iterations of a full segment
  float2 yo[NC];
  float* yprev = nullptr;
  int nprev = 0;
  auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NC; ++it) {
      const int l2 = lane + 64 * it;
      // non-temporal: the write stream does not displace the read stream's lines in L2 on its way out
      // (round 3, same box: 4.8-5.0 -> 5.5-5.8 TB/s at the vocoder's shapes, step 81.5 -> 81.15 ms)
      if (2 * l2 < nprev) __builtin_nontemporal_store(act_f32x2{yo[it].x, yo[it].y}, reinterpret_cast<act_f32x2*>(yprev + 2 * l2));
    }
  };
  for (unsigned w = w0; w < w1; ++w) {
    const int p0 = cur.p0, n_out = cur.n_out;
    float* yrow = y + cur.row_off;
    const float kf = alpha_exp[cur.c] * 0.318309886183790672f, kb = 0.5f * beta_inv[cur.c];
    // ---- phase A: registers -> LDS, then start the next item's loads
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int v = lane + 64 * k;
      const int idx = p0 - 8 + 4 * v;
      float4 t = rv[k];
      t = idx < 0 ? make_float4(t.x, t.x, t.x, t.x) : t;    // clamped address was 0:     x[0]
      t = idx >= L ? make_float4(t.w, t.w, t.w, t.w) : t;   // clamped address was L - 4: x[L-1]
      *reinterpret_cast<float4*>(raw + 4 * v) = t;
    }
    act2_compiler_fence();  // keeps the deferred stores behind the wait for rv (the scheduler hoists them into the latch otherwise)
    if (yprev) flush();     // previous item's outputs: ahead of the prefetch in vmcnt order
    if (w + 1 < w1) {
      cur = act2_item(w + 1, nseg, C, L);
      act2_load(x, cur, L, lane, rv);
    }
    act2_compiler_fence();

    // ---- phase B
    const int npairs = (2 * n_out + 13) >> 2;  // slots [0, 2 n_out + 10)
#pragma unroll 2
    for (int pi = lane; pi < npairs; pi += 64) {
      const act_f32x2* rp = reinterpret_cast<const act_f32x2*>(raw + 2 * pi + 2);
      const act_f32x2 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
      const float xv[7] = {r0.y, r1.x, r1.y, r2.x, r2.y, r3.x, r3.y};  // x[q-2 .. q+4]
      act_f32x2 u0 = {0.0f, 0.0f}, u1 = {0.0f, 0.0f};
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        u0 = __builtin_elementwise_fma(act_f32x2{xv[i], xv[i]}, hu[i], u0);          // a[2q+1], a[2q+2]
        u1 = __builtin_elementwise_fma(act_f32x2{xv[i + 1], xv[i + 1]}, hu[i], u1);  // a[2q+3], a[2q+4]
      }
      *reinterpret_cast<float4*>(a2 + 4 * pi) = make_float4(hsp_snake_hw(u0.x, kf, kb), hsp_snake_hw(u0.y, kf, kb),
                                                            hsp_snake_hw(u1.x, kf, kb), hsp_snake_hw(u1.y, kf, kb));
    }
    act2_compiler_fence();
    // replicate padding of the 2x-rate signal: a[-5 .. -1] = a[0], a[2L .. 2L+4] = a[2L-1]
    if (p0 == 0 || p0 + n_out == L) {
      if (p0 == 0 && lane < 5) a2[lane] = a2[5];
      if (p0 + n_out == L && lane >= 8 && lane < 13) a2[2 * n_out + lane - 3] = a2[2 * n_out + 4];
      act2_compiler_fence();
    }

    // ---- phase C: y[p0 + s] = sum_k hd[k] * a2[2 s + k], two outputs per lane
#pragma unroll
    for (int it = 0; it < NC; ++it) {
      const int l2 = lane + 64 * it;
      if (2 * l2 >= n_out) continue;
      const float4* ap = reinterpret_cast<const float4*>(a2 + 4 * l2);
      const float4 q0 = ap[0], q1 = ap[1], q2 = ap[2];
      const act_f32x2 q3 = *reinterpret_cast<const act_f32x2*>(a2 + 4 * l2 + 12);
      const act_f32x2 aw[7] = {{q0.x, q0.y}, {q0.z, q0.w}, {q1.x, q1.y}, {q1.z, q1.w}, {q2.x, q2.y}, {q2.z, q2.w}, q3};
      act_f32x2 s0 = {0.0f, 0.0f}, s1 = {0.0f, 0.0f};
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        s0 = __builtin_elementwise_fma(aw[k], hd[k], s0);
        s1 = __builtin_elementwise_fma(aw[k + 1], hd[k], s1);
      }
      yo[it] = make_float2(s0.x + s0.y, s1.x + s1.y);
    }
    yprev = yrow + p0;
    nprev = n_out;
    act2_compiler_fence();  // raw / a2 are rewritten by the next item
  }
  flush();
}

__global__ void snake_consts_kernel(const float* al, const float* bl, float* ea, float* binv, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) {
    ea[i] = expf(al[i]);
    binv[i] = 1.0f / (expf(bl[i]) + 1e-9f);
  }
}

// ---------------------------------------------------------------- weight prep
// one wave per row: ||v_r||_2 by a butterfly reduction, then scale
__global__ __launch_bounds__(256) void fold_weight_norm_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                               float* __restrict__ w, int rows, int cols) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* vr = v + (int64_t)row * cols;
  float ss = 0.0f;
  for (int i = lane; i < cols; i += 64) ss = fmaf(vr[i], vr[i], ss);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float sc = g[row] / sqrtf(ss);
  float* wr = w + (int64_t)row * cols;
  for (int i = lane; i < cols; i += 64) wr[i] = vr[i] * sc;
}

__global__ void gather_kernel(const float* __restrict__ src, const int32_t* __restrict__ map, float* __restrict__ dst,
                              int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t m = map[i];
    dst[i] = m >= 0 ? src[m] : 0.0f;
  }
}

// ---------------------------------------------------------------- small ops
__global__ void sequence_mask_kernel(const int64_t* length, float* mask, int B, int T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * T) mask[i] = (i % T) < length[i / T] ? 1.0f : 0.0f;
}

__global__ void flip_channels_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int T) {
  const int64_t n = (int64_t)B * C * T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int64_t r = i / T;
    const int c = (int)(r % C);
    const int64_t b = r / C;
    y[i] = x[(b * C + (C - 1 - c)) * T + t];
  }
}

__global__ void sample_prior_kernel(const float* __restrict__ stats, const float* __restrict__ noise,
                                    const float* __restrict__ mask, float* __restrict__ z, int B, int C, int T,
                                    float noise_scale) {
  const int64_t n = (int64_t)B * C * T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int64_t r = i / T;
    const int c = (int)(r % C);
    const int64_t b = r / C;
    const float m = stats[(b * 2 * C + c) * T + t];
    const float logs = stats[(b * 2 * C + C + c) * T + t];
    z[i] = (m + noise[i] * expf(logs) * noise_scale) * mask[b * T + t];
  }
}

__global__ void mask_mul_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ y,
                                int B, int C, int T) {
  const int64_t n = (int64_t)B * C * T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int64_t b = i / ((int64_t)C * T);
    y[i] = x[i] * mask[b * T + t];
  }
}

// F.interpolate(mode='linear', align_corners=False) along T.  torch's CPU kernel evaluates the
// source index in fp32 with ONE rounding, src = fmaf(scale, i + 0.5, -0.5), clamped at 0; a
// separate mul + sub is off by 2e-3 at L = 64000 because ulp(src) ~ 4e-3 there.
__global__ void linear_interp_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int Lin,
                                     int Lout, float scale) {
  const int64_t n = rows * Lout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % Lout);
    const int64_t r = i / Lout;
    float src = fmaf(scale, (float)t + 0.5f, -0.5f);
    src = src < 0.0f ? 0.0f : src;
    const int i0 = (int)src;
    const int i1 = i0 + (i0 < Lin - 1 ? 1 : 0);
    const float l1 = src - (float)i0, l0 = 1.0f - l1;
    const float* xr = x + r * Lin;
    y[i] = l0 * xr[i0] + l1 * xr[i1];
  }
}

__global__ void act_kernel(const float* x, float* y, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = hsp_apply_act(x[i], act);
}

__global__ void reflect_pad_kernel(const float* x, int64_t x_bs, float* y, int B, int L, int pad) {
  const int Lo = L + 2 * pad;
  const int64_t n = (int64_t)B * Lo;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / Lo), t = (int)(i - (int64_t)b * Lo) - pad;
    const int s = t < 0 ? -t : (t >= L ? 2 * (L - 1) - t : t);
    y[i] = x[(int64_t)b * x_bs + s];
  }
}

// one workgroup: voiced statistics of both tracks in double (the reference's numpy float32 mean / std differ from the
// exact values by ~1e-7 relative; double keeps this side at the exact ones), then the conversion
__global__ __launch_bounds__(256) void f0_convert_kernel(const float* src, int ns, const float* trg, int nt, float* out) {
  __shared__ double red[4][256];
  __shared__ double stat[4];
  double s1 = 0, c1 = 0, s2 = 0, c2 = 0;
  for (int i = threadIdx.x; i < ns; i += 256) if (src[i] != 0.0f) { s1 += src[i]; c1 += 1; }
  for (int i = threadIdx.x; i < nt; i += 256) if (trg[i] != 0.0f) { s2 += trg[i]; c2 += 1; }
  red[0][threadIdx.x] = s1; red[1][threadIdx.x] = c1; red[2][threadIdx.x] = s2; red[3][threadIdx.x] = c2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  const double m1 = red[1][0] > 0 ? red[0][0] / red[1][0] : 0.0, m2 = red[3][0] > 0 ? red[2][0] / red[3][0] : 0.0;
  const double n1 = red[1][0], n2 = red[3][0];
  __syncthreads();
  double v1 = 0, v2 = 0;
  for (int i = threadIdx.x; i < ns; i += 256) if (src[i] != 0.0f) { const double d = src[i] - m1; v1 += d * d; }
  for (int i = threadIdx.x; i < nt; i += 256) if (trg[i] != 0.0f) { const double d = trg[i] - m2; v2 += d * d; }
  red[0][threadIdx.x] = v1; red[2][threadIdx.x] = v2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[2][threadIdx.x] += red[2][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    stat[0] = m1; stat[1] = n1 > 0 ? sqrt(red[0][0] / n1) : 1.0; stat[2] = m2; stat[3] = n2 > 0 ? sqrt(red[2][0] / n2) : 0.0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ns; i += 256) {
    float o = 0.0f;
    if (src[i] != 0.0f) {
      // the reference's steps, in float64 like its numpy arrays (get_yaapt_f0 works on float64), then the float32 cast
      // of torch.FloatTensor(f0 + 1) and a float32 log
      const double z = ((double)src[i] - stat[0]) / stat[1];
      const double f = fmax(z * stat[3] + stat[2], 0.0);
      o = logf((float)(f + 1.0));
    }
    out[i] = o;
  }
}

__global__ void axpby_kernel(const float* x, const float* z, float* y, float a, float b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = a * x[i] + b * z[i];
}

// LayerNorm over channels of a (B, C, T) tensor.  Block = 64 time steps x 4 channel
// groups; lanes run along T (coalesced), the 4 waves split C and combine through LDS.
// Two passes (mean, then centred variance) like torch's CPU kernel.
__global__ __launch_bounds__(256) void layernorm_mod_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                            int T, float eps, const float* __restrict__ mask,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ scale, int64_t mod_bs,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int n_tt) {
  __shared__ float red[4][64];
  const int tt = blockIdx.x % n_tt, b = blockIdx.x / n_tt;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int t = tt * 64 + lane;
  const bool live = t < T;
  const float* xb = x + (int64_t)b * C * T;
  float* yb = y + (int64_t)b * C * T;
  float s = 0.0f;
  if (live)
    for (int c = grp; c < C; c += 4) s += xb[(int64_t)c * T + t];
  red[grp][lane] = s;
  __syncthreads();
  const float mean = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / (float)C;
  __syncthreads();
  float ss = 0.0f;
  if (live)
    for (int c = grp; c < C; c += 4) {
      const float d = xb[(int64_t)c * T + t] - mean;
      ss = fmaf(d, d, ss);
    }
  red[grp][lane] = ss;
  __syncthreads();
  const float var = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / (float)C;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (!live) return;
  const float mk = mask ? mask[(int64_t)b * T + t] : 1.0f;
  for (int c = grp; c < C; c += 4) {
    float v = (xb[(int64_t)c * T + t] - mean) * rstd;
    if (gamma) v = v * gamma[c] + beta[c];
    v *= mk;
    if (scale) v = v * (1.0f + scale[(int64_t)b * mod_bs + c]) + shift[(int64_t)b * mod_bs + c];
    yb[(int64_t)c * T + t] = v;
  }
}

// Register-resident variant for C <= 16 * LN_R = 512: a workgroup owns 16 time columns, its 256 threads
// are 16 columns x 16 channel groups and every thread keeps its <= LN_R channel values in
// registers, so the tile is read once with all loads in flight (the loop kernel above chases
// C/4 dependent loads three times: ~40 us for C = 276 whatever T is).  Short sequences still
// give B * T/16 workgroups.
constexpr int LN_R = 32;
__global__ __launch_bounds__(256) void layernorm_reg_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                            int T, float eps, const float* __restrict__ mask,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ scale, int64_t mod_bs,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int n_tt) {
  __shared__ float red[16][17];
  const int tt = blockIdx.x % n_tt, b = blockIdx.x / n_tt;
  const int lt = threadIdx.x & 15, cg = threadIdx.x >> 4;
  const int t = tt * 16 + lt;
  const bool live = t < T;
  const float* xb = x + (int64_t)b * C * T + t;
  float* yb = y + (int64_t)b * C * T + t;
  float v[LN_R];
#pragma unroll
  for (int r = 0; r < LN_R; ++r) {
    const int c = cg + 16 * r;
    v[r] = (live && c < C) ? xb[(int64_t)c * T] : 0.0f;
  }
  float s = 0.0f;
#pragma unroll
  for (int r = 0; r < LN_R; ++r) s += v[r];
  red[cg][lt] = s;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int g = 0; g < 16; ++g) tot += red[g][lt];
  const float mean = tot / (float)C;
  __syncthreads();
  float ss = 0.0f;
#pragma unroll
  for (int r = 0; r < LN_R; ++r) {
    const float d = (cg + 16 * r < C) ? v[r] - mean : 0.0f;
    ss = fmaf(d, d, ss);
  }
  red[cg][lt] = ss;
  __syncthreads();
  tot = 0.0f;
#pragma unroll
  for (int g = 0; g < 16; ++g) tot += red[g][lt];
  const float rstd = 1.0f / sqrtf(tot / (float)C + eps);
  if (!live) return;
  const float mk = mask ? mask[(int64_t)b * T + t] : 1.0f;
#pragma unroll
  for (int r = 0; r < LN_R; ++r) {
    const int c = cg + 16 * r;
    if (c >= C) break;
    float o = (v[r] - mean) * rstd;
    if (gamma) o = o * gamma[c] + beta[c];
    o *= mk;
    if (scale) o = o * (1.0f + scale[(int64_t)b * mod_bs + c]) + shift[(int64_t)b * mod_bs + c];
    yb[(int64_t)c * T] = o;
  }
}

// out[b, c] = sum_t x[b, c, t] / sum_t mask[b, t]; one wave per (b, c)
__global__ __launch_bounds__(256) void masked_mean_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                          float* __restrict__ out, int B, int C, int T) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= B * C) return;
  const int b = row / C;
  float s = 0.0f, m = 0.0f;
  for (int t = lane; t < T; t += 64) {
    s += x[(int64_t)row * T + t];
    m += mask[(int64_t)b * T + t];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    m += __shfl_xor(m, o, 64);
  }
  if (lane == 0) out[row] = s / m;
}

}  // namespace

#define HSP_STREAM static_cast<hipStream_t>(stream)

extern "C" int hsp_version(void) { return HSP_VERSION; }
extern "C" const char* hsp_arch(void) { return "gfx950"; }

extern "C" int hsp_act1d_snakebeta_f32(const float* x, float* y, int32_t B, int32_t C, int32_t L,
                                       const float* alpha_exp, const float* beta_inv, const float* filt, void* stream) {
  if (!x || !y || !alpha_exp || !beta_inv || !filt || B <= 0 || C <= 0 || L <= 0) return HSP_EINVAL;
  if ((L & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
    const int nseg = (L + ACT2_SEG - 1) / ACT2_SEG;
    const int64_t nwork = (int64_t)nseg * B * C;
    const int64_t blocks = (nwork + ACT2_WAVES * ACT2_ITEMS - 1) / (ACT2_WAVES * ACT2_ITEMS);
    if (nwork > 0x7fffffff - ACT2_WAVES * ACT2_ITEMS) return HSP_EINVAL;
    hipLaunchKernelGGL(act1d_seg_kernel, dim3((unsigned)blocks), dim3(64 * ACT2_WAVES), 0, HSP_STREAM, x, y, C, L,
                       alpha_exp, beta_inv, filt, nseg, (unsigned)nwork);
    return (int)hipGetLastError();
  }
  const int n_tiles = (L + ACT_TILE - 1) / ACT_TILE;
  const int64_t blocks = (int64_t)n_tiles * B * C;
  if (blocks > 0x7fffffff) return HSP_EINVAL;
  hipLaunchKernelGGL(act1d_kernel, dim3((unsigned)blocks), dim3(ACT_THREADS), 0, HSP_STREAM, x, y, C, L, alpha_exp,
                     beta_inv, filt, n_tiles);
  return (int)hipGetLastError();
}

extern "C" int hsp_snake_consts_f32(const float* al, const float* bl, float* ea, float* binv, int32_t C, void* stream) {
  if (!al || !bl || !ea || !binv || C <= 0) return HSP_EINVAL;
  hipLaunchKernelGGL(snake_consts_kernel, dim3((C + 255) / 256), dim3(256), 0, HSP_STREAM, al, bl, ea, binv, C);
  return (int)hipGetLastError();
}

extern "C" int hsp_fold_weight_norm_f32(const float* v, const float* g, float* w, int32_t rows, int32_t cols,
                                        void* stream) {
  if (!v || !g || !w || rows <= 0 || cols <= 0) return HSP_EINVAL;
  hipLaunchKernelGGL(fold_weight_norm_kernel, dim3((rows + 3) / 4), dim3(256), 0, HSP_STREAM, v, g, w, rows, cols);
  return (int)hipGetLastError();
}

extern "C" int hsp_gather_f32(const float* src, const int32_t* map, float* dst, int64_t n, void* stream) {
  if (!src || !map || !dst || n <= 0) return HSP_EINVAL;
  hipLaunchKernelGGL(gather_kernel, dim3(grid_for(n, 256)), dim3(256), 0, HSP_STREAM, src, map, dst, n);
  return (int)hipGetLastError();
}

extern "C" int hsp_sequence_mask_f32(const int64_t* length, float* mask, int32_t B, int32_t T, void* stream) {
  if (!length || !mask || B <= 0 || T <= 0) return HSP_EINVAL;
  hipLaunchKernelGGL(sequence_mask_kernel, dim3((B * T + 255) / 256), dim3(256), 0, HSP_STREAM, length, mask, B, T);
  return (int)hipGetLastError();
}

extern "C" int hsp_flip_channels_f32(const float* x, float* y, int32_t B, int32_t C, int32_t T, void* stream) {
  if (!x || !y || x == y || B <= 0 || C <= 0 || T <= 0) return HSP_EINVAL;
  hipLaunchKernelGGL(flip_channels_kernel, dim3(grid_for((int64_t)B * C * T, 256)), dim3(256), 0, HSP_STREAM, x, y, B,
                     C, T);
  return (int)hipGetLastError();
}

extern "C" int hsp_sample_prior_f32(const float* stats, const float* noise, const float* mask, float* z, int32_t B,
                                    int32_t C, int32_t T, float noise_scale, void* stream) {
  if (!stats || !noise || !mask || !z || B <= 0 || C <= 0 || T <= 0) return HSP_EINVAL;
  hipLaunchKernelGGL(sample_prior_kernel, dim3(grid_for((int64_t)B * C * T, 256)), dim3(256), 0, HSP_STREAM, stats,
                     noise, mask, z, B, C, T, noise_scale);
  return (int)hipGetLastError();
}

extern "C" int hsp_mask_mul_f32(const float* x, const float* mask, float* y, int32_t B, int32_t C, int32_t T,
                                void* stream) {
  if (!x || !mask || !y || B <= 0 || C <= 0 || T <= 0) return HSP_EINVAL;
  hipLaunchKernelGGL(mask_mul_kernel, dim3(grid_for((int64_t)B * C * T, 256)), dim3(256), 0, HSP_STREAM, x, mask, y, B,
                     C, T);
  return (int)hipGetLastError();
}

extern "C" int hsp_linear_interp_f32(const float* x, float* y, int32_t B, int32_t C, int32_t Lin, int32_t Lout,
                                    void* stream) {
  if (!x || !y || B <= 0 || C <= 0 || Lin <= 0 || Lout <= 0) return HSP_EINVAL;
  const float scale = (float)Lin / (float)Lout;  // torch: area_pixel_compute_scale, align_corners=False
  hipLaunchKernelGGL(linear_interp_kernel, dim3(grid_for((int64_t)B * C * Lout, 256)), dim3(256), 0, HSP_STREAM, x, y,
                     (int64_t)B * C, Lin, Lout, scale);
  return (int)hipGetLastError();
}

extern "C" int hsp_act_f32(const float* x, float* y, int64_t n, int32_t act, void* stream) {
  if (!x || !y || n <= 0 || act < 0 || act > HSP_ACT_GELU_ERF) return HSP_EINVAL;
  hipLaunchKernelGGL(act_kernel, dim3(grid_for(n, 256)), dim3(256), 0, HSP_STREAM, x, y, n, act);
  return (int)hipGetLastError();
}

extern "C" int hsp_reflect_pad_f32(const float* x, int64_t x_bs, float* y, int32_t B, int32_t L, int32_t pad, void* stream) {
  if (!x || !y || B <= 0 || L <= 0 || pad < 0 || pad >= L) return HSP_EINVAL;
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(grid_for((int64_t)B * (L + 2 * pad), 256)), dim3(256), 0, HSP_STREAM, x, x_bs,
                     y, B, L, pad);
  return (int)hipGetLastError();
}

extern "C" int hsp_f0_convert_f32(const float* f0_src, int32_t n_src, const float* f0_trg, int32_t n_trg, float* out,
                                  void* stream) {
  if (!f0_src || !f0_trg || !out || n_src <= 0 || n_trg <= 0) return HSP_EINVAL;
  hipLaunchKernelGGL(f0_convert_kernel, dim3(1), dim3(256), 0, HSP_STREAM, f0_src, n_src, f0_trg, n_trg, out);
  return (int)hipGetLastError();
}

extern "C" int hsp_axpby_f32(const float* x, const float* z, float* y, float a, float b, int64_t n, void* stream) {
  if (!x || !z || !y || n <= 0) return HSP_EINVAL;
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n, 256)), dim3(256), 0, HSP_STREAM, x, z, y, a, b, n);
  return (int)hipGetLastError();
}

extern "C" int hsp_layernorm_mod_f32(const float* x, float* y, int32_t B, int32_t C, int32_t T, float eps,
                                     const float* mask, const float* shift, const float* scale, int64_t mod_bs,
                                     const float* gamma, const float* beta, void* stream) {
  if (!x || !y || B <= 0 || C <= 0 || T <= 0) return HSP_EINVAL;
  if ((shift == nullptr) != (scale == nullptr) || (gamma == nullptr) != (beta == nullptr)) return HSP_EINVAL;
  if (C <= 16 * LN_R) {
    const int n_t16 = (T + 15) / 16;
    hipLaunchKernelGGL(layernorm_reg_kernel, dim3((unsigned)(n_t16 * B)), dim3(256), 0, HSP_STREAM, x, y, C, T, eps,
                       mask, shift, scale, mod_bs, gamma, beta, n_t16);
    return (int)hipGetLastError();
  }
  const int n_tt = (T + 63) / 64;
  hipLaunchKernelGGL(layernorm_mod_kernel, dim3((unsigned)(n_tt * B)), dim3(256), 0, HSP_STREAM, x, y, C, T, eps, mask,
                     shift, scale, mod_bs, gamma, beta, n_tt);
  return (int)hipGetLastError();
}

extern "C" int hsp_masked_mean_f32(const float* x, const float* mask, float* out, int32_t B, int32_t C, int32_t T,
                                   void* stream) {
  if (!x || !mask || !out || B <= 0 || C <= 0 || T <= 0) return HSP_EINVAL;
  hipLaunchKernelGGL(masked_mean_kernel, dim3((B * C + 3) / 4), dim3(256), 0, HSP_STREAM, x, mask, out, B, C, T);
  return (int)hipGetLastError();
}
