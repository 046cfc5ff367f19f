// Implicit-GEMM convolution on the gfx950 matrix cores for the wide layers of the CIFAR ResNet
// (reference workload: example-cifar100/src/main/scala/lamp/example/cifar/cnn.scala:89-137 - res3/res4
// blocks: 3x3 stride-1 pad-1 and 1x1 convolutions on 8x8 maps with 16/100/128 channels, which are
// 97 % of the network's FLOPs; reference op: ops.scala:1547-1651).
//
// Scope of the fast path: bf16, NCHW, H = W = 8, kernel 3x3 (pad 1) or 1x1 (pad 0), stride 1,
// dilation 1, groups 1, Cin <= 128, Cout <= 128.  Everything else stays on the direct kernels.
//
// fprop and dgrad share ONE kernel (dgrad = fprop of dy with the weights transposed and the taps
// mirrored):   Out[co][p] = sum_{r,s,ci} Wp[rs][co][ci] * X[ci][p + (r,s) - pad]
//   M = output channels (128 per workgroup, zero padded), N = 128 pixels = two images,
//   K = taps * Cin; v_mfma_f32_16x16x32_bf16, fp32 accumulation.
//   * the two input images live in LDS channel-LAST ([10x10 padded pixel][ci]) so that a B fragment
//     (8 consecutive ci of one shifted pixel) is one ds_read_b128; the NCHW -> channel-last
//     transposition happens once per image while staging (coalesced 16-byte global loads).
//   * weights are pre-packed by a tiny kernel to [tap][co][ci] (ci contiguous) and streamed
//     through a double-buffered LDS tile, one (tap, 64-channel chunk) per stage.
//   * both LDS images are XOR-swizzled on 16-byte chunks so every fragment read is conflict-free
//     (see the lane -> pixel permutation in px_of_col).
// wgrad is a plain NT GEMM per tap, dW[rs][co][ci] = sum_{n,p} dY[n][co][p] * Xshift_rs[n][ci][p]:
//   the K dimension runs over pixels of many images; the tap shift is applied while staging X
//   (row select + a 16-bit funnel shift inside the 16-byte row), partial sums of the image
//   splits go to an fp32 workspace and a small kernel reduces them into dW[co][ci][r][s].
#include "device_utils.h"
#include "conv_geom.h"

namespace lamp {

typedef short s8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int IG_M = 128;          // output channels per workgroup (padded)
constexpr int IG_WTILE = 128 * 64 * 2;   // one weight stage: 128 rows x 64 k, bf16

// [128 rows][64 k] K-contiguous tile, 128-byte rows, chunk' = chunk ^ (row & 7)  (same image as gemm.hip)
__device__ __forceinline__ int ig_kc_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ void ig_stage_load_rows(uint4 (&r)[4], const bf16_t* __restrict__ base, int64_t ld, int64_t k0, int64_t rows,
                                                   int64_t K, int tid) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = tid + i * 256;
    const int64_t gr = c >> 3, gk = k0 + ((c & 7) << 3);
    if (gr < rows && gk + 8 <= K) r[i] = *reinterpret_cast<const uint4*>(base + gr * ld + gk);
    else r[i] = make_uint4(0, 0, 0, 0);
  }
}
__device__ __forceinline__ void ig_stage_store_rows(const uint4 (&r)[4], char* lds, int tid) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = tid + i * 256;
    *reinterpret_cast<uint4*>(lds + ig_kc_off(c >> 3, c & 7)) = r[i];
  }
}
__device__ __forceinline__ bf8v ig_frag_rows(const char* lds, int row0, int s, int lane) {
  s8v v = *reinterpret_cast<const s8v*>(lds + ig_kc_off(row0 + (lane & 15), s * 4 + (lane >> 4)));
  return __builtin_bit_cast(bf8v, v);
}

// MFMA column -> pixel inside a 16-pixel tile (two image rows): the permutation that makes the
// ds_read_b128 lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} hit 16 different 16-byte slots
__device__ __forceinline__ void px_of_col(int c, int& rowsel, int& w) {
  if (c < 4) { rowsel = 0; w = c; }
  else if (c >= 12) { rowsel = 0; w = c - 8; }
  else { rowsel = 1; w = c - 4; }
}
__device__ __forceinline__ int x_swz(int hp, int wp, int nchunk_mask) { return (((hp & 1) << 3) | (wp & 7)) & nchunk_mask; }

// ---- weight packing ---------------------------------------------------------------------------------
// fprop: wp[rs][co][ci] = W[co][ci][r][s]            (rows = Cout, k = Cin)
// dgrad: wp[rs][ci][co] = W[co][ci][kh-1-r][kw-1-s]  (rows = Cin,  k = Cout)
__global__ void ig_pack_weights_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wp, int Cout, int Cin, int KS, int KP, int dgrad) {
  const int RS = KS * KS;
  const int total = RS * IG_M * KP;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int k = e % KP, row = (e / KP) % IG_M, rs = e / (KP * IG_M);
    const int r = rs / KS, s = rs % KS;
    bf16_t v; v.bits = 0;
    if (!dgrad) { if (row < Cout && k < Cin) v = w[((row * Cin + k) * KS + r) * KS + s]; }
    else { if (row < Cin && k < Cout) v = w[((k * Cin + row) * KS + (KS - 1 - r)) * KS + (KS - 1 - s)]; }
    wp[e] = v;
  }
}

// ---- fprop / dgrad -----------------------------------------------------------------------------------
// x [N][CI][64], wp [RS][128][KP], y [N][CO][64].  KP = padded K per tap (32, 64 or 128).
template <int KS, int NW>
__global__ __launch_bounds__(NW * 128) void ig_conv8_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, const bf16_t* __restrict__ bias,
                                                            bf16_t* __restrict__ y, int N, int CI, int KP, int CO) {
  // NW images per workgroup, 2 * NW waves: wave (wr, wc) owns output channels [64 wr, 64 wr + 64) of image wc.
  // One weight stage feeds NW images, so a wider workgroup halves the L2 weight traffic and the barriers per MFMA.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = KS * KS;
  constexpr int PAD = (KS - 1) / 2;
  constexpr int NT = NW * 128;
  constexpr int LPT = 1024 / NT;            // 16-byte weight chunks per thread and stage
  const int RB = KP * 2;                    // bytes per pixel row of the channel-last image
  const int XIMG = 100 * RB;                // one padded 10x10 image
  char* Xl = smem;                          // [NW][100][KP]
  char* Wl = smem + NW * XIMG;              // 2 x IG_WTILE
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid / NW, wc = wid % NW;
  const int n0 = blockIdx.x * NW;
  const int cmask = (KP >> 3) - 1;

  // zero the image tiles (borders and padded channels must read as 0)
  for (int o = tid * 16; o < NW * XIMG; o += NT * 16) *reinterpret_cast<uint4*>(Xl + o) = make_uint4(0, 0, 0, 0);
  __syncthreads();
  // NCHW -> channel-last: one 16-byte global load = one image row (8 pixels) of one channel
  for (int img = 0; img < NW; img++) {
    const int n = n0 + img;
    if (n >= N) break;
    const bf16_t* xp = x + (int64_t)n * CI * 64;
    char* xi = Xl + img * XIMG;
    for (int e = tid; e < CI * 8; e += NT) {
      const int ci = e >> 3, h = e & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(xp + ci * 64 + h * 8);
      const unsigned int words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int wq = 0; wq < 8; wq++) {
        const unsigned short el = (unsigned short)(words[wq >> 1] >> ((wq & 1) * 16));
        const int hp = h + 1, wpx = wq + 1;
        const int off = (hp * 10 + wpx) * RB + ((((ci >> 3) ^ x_swz(hp, wpx, cmask))) << 4) + ((ci & 7) << 1);
        *reinterpret_cast<unsigned short*>(xi + off) = el;
      }
    }
  }

  f4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};

  const int KW = KP < 64 ? KP : 64;         // k per stage
  const int CC = KP / KW;                   // chunks per tap
  const int T = RS * CC;
  uint4 rw[LPT];
  auto stage_load = [&](const bf16_t* base, int k0) {
#pragma unroll
    for (int i = 0; i < LPT; i++) {
      const int c = tid + i * NT;
      const int gr = c >> 3, gk = k0 + ((c & 7) << 3);
      rw[i] = (gk + 8 <= KP) ? *reinterpret_cast<const uint4*>(base + gr * KP + gk) : make_uint4(0, 0, 0, 0);
    }
  };
  auto stage_store = [&](char* lds) {
#pragma unroll
    for (int i = 0; i < LPT; i++) {
      const int c = tid + i * NT;
      *reinterpret_cast<uint4*>(lds + ig_kc_off(c >> 3, c & 7)) = rw[i];
    }
  };
  stage_load(wp, 0);
  stage_store(Wl);
  __syncthreads();

  // per-lane pixel of each of this wave's 4 n-tiles (image wc, image rows 2j, 2j+1)
  int rowsel, wpix;
  px_of_col(lane & 15, rowsel, wpix);
  const char* ximg = Xl + wc * XIMG;

  for (int t = 0; t < T; t++) {
    const int cur = t & 1;
    const int rs = t / CC, cc = t - rs * CC;
    if (t + 1 < T) {
      const int rs1 = (t + 1) / CC, cc1 = (t + 1) - rs1 * CC;
      stage_load(wp + (int64_t)rs1 * IG_M * KP, cc1 * KW);
    }
    const int r = rs / KS, s = rs - r * KS;
    const char* wl = Wl + cur * IG_WTILE;
    const int ksteps = KW >> 5;
#pragma unroll 2
    for (int ks = 0; ks < ksteps; ks++) {
      bf8v fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; i++) fa[i] = ig_frag_rows(wl, wr * 64 + i * 16, ks, lane);
      const int chunk = cc * (KW >> 3) + ks * 4 + (lane >> 4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int hp = 2 * j + rowsel + r + (1 - PAD), wpx = wpix + s + (1 - PAD);
        const int off = (hp * 10 + wpx) * RB + ((chunk ^ x_swz(hp, wpx, cmask)) << 4);
        s8v v = *reinterpret_cast<const s8v*>(ximg + off);
        fb[j] = __builtin_bit_cast(bf8v, v);
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < T) stage_store(Wl + (cur ^ 1) * IG_WTILE);
    __syncthreads();
  }

  // epilogue: D rows = output channel, D cols = pixels
  const int n = n0 + wc;
  if (n < N) {
    bf16_t* yp = y + (int64_t)n * CO * 64;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int co = wr * 64 + i * 16 + (lane >> 4) * 4 + rr;
        if (co < CO) {
          const float b = bias ? (float)bias[co] : 0.f;
#pragma unroll
          for (int j = 0; j < 4; j++) yp[co * 64 + (2 * j + rowsel) * 8 + wpix] = bf16_t(acc[i][j][rr] + b);
        }
      }
    }
  }
}

// ---- wgrad --------------------------------------------------------------------------------------------
// partial[(split * RS + rs)][128][128] (fp32) = sum over the split's images of dY[n] (128 x 64) . Xshift_rs[n]^T (64 x 128)
template <int KS>
__global__ __launch_bounds__(256) void ig_wgrad8_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ partial,
                                                        int N, int CO, int CI, int images_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = KS * KS;
  constexpr int PAD = (KS - 1) / 2;
  const int rs = blockIdx.x % RS, split = blockIdx.x / RS;
  const int r = rs / KS, s = rs - r * KS;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int nbeg = split * images_per_split;
  const int nend = min(nbeg + images_per_split, N);

  auto load_x = [&](uint4 (&rx)[4], int n) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = tid + i * 256;
      const int ci = c >> 3, h = c & 7;           // k chunk = image row h
      const int hs = h + r - PAD;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ci < CI && hs >= 0 && hs < 8) {
        v = *reinterpret_cast<const uint4*>(x + ((int64_t)n * CI + ci) * 64 + hs * 8);
        const int sh = s - PAD;                    // out[w] = in[w + sh]
        if (sh < 0) v = make_uint4(v.x << 16, (v.y << 16) | (v.x >> 16), (v.z << 16) | (v.y >> 16), (v.w << 16) | (v.z >> 16));
        else if (sh > 0) v = make_uint4((v.x >> 16) | (v.y << 16), (v.y >> 16) | (v.z << 16), (v.z >> 16) | (v.w << 16), v.w >> 16);
      }
      rx[i] = v;
    }
  };

  f4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  if (nbeg < nend) {
    ig_stage_load_rows(ra, dy + (int64_t)nbeg * CO * 64, 64, 0, CO, 64, tid);
    load_x(rb, nbeg);
    ig_stage_store_rows(ra, smem, tid);
    ig_stage_store_rows(rb, smem + IG_WTILE, tid);
  }
  __syncthreads();
  for (int n = nbeg; n < nend; n++) {
    const int cur = (n - nbeg) & 1;
    if (n + 1 < nend) {
      ig_stage_load_rows(ra, dy + (int64_t)(n + 1) * CO * 64, 64, 0, CO, 64, tid);
      load_x(rb, n + 1);
    }
    const char* al = smem + cur * 2 * IG_WTILE;
    const char* bl = al + IG_WTILE;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf8v fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; i++) fa[i] = ig_frag_rows(al, wr * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; j++) fb[j] = ig_frag_rows(bl, wc * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (n + 1 < nend) {
      ig_stage_store_rows(ra, smem + (cur ^ 1) * 2 * IG_WTILE, tid);
      ig_stage_store_rows(rb, smem + (cur ^ 1) * 2 * IG_WTILE + IG_WTILE, tid);
    }
    __syncthreads();
  }
  float* out = partial + (int64_t)blockIdx.x * IG_M * IG_M;   // blockIdx.x = split * RS + rs
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int co = wr * 64 + i * 16 + (lane >> 4) * 4 + rr, ci = wc * 64 + j * 16 + (lane & 15);
        out[co * IG_M + ci] = acc[i][j][rr];
      }
}
// dw[co][ci][r][s] = sum_split partial[split][rs][co][ci]; threads run along ci (the contiguous index of the partials)
__global__ void ig_wgrad_reduce_kernel(const float* __restrict__ partial, bf16_t* __restrict__ dw, int CO, int CI, int RS, int nsplit) {
  const int total = RS * CO * CI;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int ci = e % CI, co = (e / CI) % CO, rs = e / (CI * CO);
    float a = 0.f;
    for (int sp = 0; sp < nsplit; sp++) a += partial[((int64_t)(sp * RS + rs) * IG_M + co) * IG_M + ci];
    dw[((int64_t)co * CI + ci) * RS + rs] = bf16_t(a);
  }
}

// ---- wgrad v2: all taps in one workgroup ----------------------------------------------------------------
// Workgroup = (32-channel slice of Cin, image range); it owns dW[all taps][128 co][32 ci] in registers (9 x 4 MFMA
// tiles per wave).  Per image: dY[128 co][64 px] is staged K-contiguous (swizzled 128-B rows) and X[32 ci][8x8] is
// staged as THREE horizontally shifted copies with a zero row above and below ([s][ci][10 rows][8 px], 176-B channel
// stride => conflict-free ds_read_b128), so the B fragment of tap (r, s) for image rows 4ks..4ks+3 is one aligned
// 16-byte read.  A fragments are shared by the 9 taps: 26 LDS reads feed 72 MFMAs per wave and image.
constexpr int WG_CI = 32;                         // input channels per workgroup
constexpr int WG_XCH = 176;                       // bytes per channel in one shifted copy (11 x 16)
constexpr int WG_XCOPY = WG_CI * WG_XCH;          // 5632
constexpr int WG_STAGE = IG_WTILE + 3 * WG_XCOPY + 512;   // dY tile + 3 copies (+ pad to keep stages 16-B aligned) = 33,792
template <int KS>
__global__ __launch_bounds__(256) void ig_wgrad8v2_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ partial,
                                                          int N, int CO, int CI, int CIP, int images_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = KS * KS;
  constexpr int PAD = (KS - 1) / 2;
  const int tile = blockIdx.x, split = blockIdx.y;
  const int ci0 = tile * WG_CI;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int nbeg = split * images_per_split, nend = min(nbeg + images_per_split, N);
  // zero both stages once: the padding rows of the shifted copies are never written again
  for (int o = tid * 16; o < 2 * WG_STAGE; o += 256 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
  __syncthreads();

  const int xci = tid >> 3, xh = tid & 7;          // this thread's (channel, image row) of the X tile
  auto load_x = [&](int n) -> uint4 {
    if (ci0 + xci < CI) return *reinterpret_cast<const uint4*>(x + ((int64_t)n * CI + ci0 + xci) * 64 + xh * 8);
    return make_uint4(0, 0, 0, 0);
  };
  auto store_x = [&](char* stage, uint4 v) {
    char* xb = stage + IG_WTILE + xci * WG_XCH + (xh + 1) * 16;
    if (KS == 1) { *reinterpret_cast<uint4*>(xb) = v; return; }
    // copy s holds out[w] = in[w + s - 1]
    *reinterpret_cast<uint4*>(xb) = make_uint4(v.x << 16, (v.y << 16) | (v.x >> 16), (v.z << 16) | (v.y >> 16), (v.w << 16) | (v.z >> 16));
    *reinterpret_cast<uint4*>(xb + WG_XCOPY) = v;
    *reinterpret_cast<uint4*>(xb + 2 * WG_XCOPY) = make_uint4((v.x >> 16) | (v.y << 16), (v.y >> 16) | (v.z << 16), (v.z >> 16) | (v.w << 16), v.w >> 16);
  };

  f4v acc[RS][4];
#pragma unroll
  for (int t = 0; t < RS; t++)
#pragma unroll
    for (int i = 0; i < 4; i++) acc[t][i] = f4v{0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rx;
  if (nbeg < nend) {
    ig_stage_load_rows(ra, dy + (int64_t)nbeg * CO * 64, 64, 0, CO, 64, tid);
    rx = load_x(nbeg);
    ig_stage_store_rows(ra, smem, tid);
    store_x(smem, rx);
  }
  __syncthreads();
  for (int n = nbeg; n < nend; n++) {
    const int cur = (n - nbeg) & 1;
    if (n + 1 < nend) {
      ig_stage_load_rows(ra, dy + (int64_t)(n + 1) * CO * 64, 64, 0, CO, 64, tid);
      rx = load_x(n + 1);
    }
    const char* st = smem + cur * WG_STAGE;
    const char* xl = st + IG_WTILE + (wc * 16 + (lane & 15)) * WG_XCH;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf8v fa[4];
#pragma unroll
      for (int i = 0; i < 4; i++) fa[i] = ig_frag_rows(st, wr * 64 + i * 16, ks, lane);
      const int h = 4 * ks + (lane >> 4);
#pragma unroll
      for (int t = 0; t < RS; t++) {
        const int r = t / KS, s = t % KS;
        s8v v = *reinterpret_cast<const s8v*>(xl + s * WG_XCOPY + (h + r + (1 - PAD)) * 16);
        const bf8v fb = __builtin_bit_cast(bf8v, v);
#pragma unroll
        for (int i = 0; i < 4; i++) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[t][i], 0, 0, 0);
      }
    }
    if (n + 1 < nend) {
      char* nx = smem + (cur ^ 1) * WG_STAGE;
      ig_stage_store_rows(ra, nx, tid);
      store_x(nx, rx);
    }
    __syncthreads();
  }
  // partial[(split * RS + t)][co][CIP]
#pragma unroll
  for (int t = 0; t < RS; t++) {
    float* out = partial + (int64_t)(split * RS + t) * IG_M * CIP;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int co = wr * 64 + i * 16 + (lane >> 4) * 4 + rr, ci = ci0 + wc * 16 + (lane & 15);
        out[co * CIP + ci] = acc[t][i][rr];
      }
  }
}
// Split reduction: thread (q, sg) sums float4 column q of this block over the splits sg, sg+8, ... (independent 16-byte
// loads in flight), the 8 split groups are then combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void ig_wgrad_reduce_v2_kernel(const float* __restrict__ partial, bf16_t* __restrict__ dw, int CO, int CI, int CIP, int RS,
                                                                 int nsplit) {
  __shared__ float4 red[8][32];
  const int q = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int64_t per_split = (int64_t)RS * IG_M * CIP / 4;   // float4 elements of one split
  const int64_t col = (int64_t)blockIdx.x * 32 + q;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < per_split) {
    const float4* p4 = reinterpret_cast<const float4*>(partial) + col;
#pragma unroll 4
    for (int sp = sg; sp < nsplit; sp += 8) {
      const float4 v = p4[(int64_t)sp * per_split];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[sg][q] = a;
  __syncthreads();
  if (sg == 0 && col < per_split) {
#pragma unroll
    for (int g = 1; g < 8; g++) { const float4 v = red[g][q]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    const int c4 = CIP / 4;
    const int ci = (int)(col % c4) * 4, co = (int)((col / c4) % IG_M), rs = (int)(col / ((int64_t)c4 * IG_M));
    if (co < CO) {
      const float r[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (ci + k < CI) dw[((int64_t)co * CI + ci + k) * RS + rs] = bf16_t(r[k]);
    }
  }
}

// ---- host ---------------------------------------------------------------------------------------------
static bool ig_qualifies(const ConvGeom& g, int dtype) {
  if (dtype != kBF16) return false;
  if (g.groups != 1 || g.transposed) return false;
  if (g.H != 8 || g.W != 8 || g.Ho != 8 || g.Wo != 8) return false;
  if (g.sh != 1 || g.sw != 1 || g.dh != 1 || g.dw != 1) return false;
  if (!((g.kh == 3 && g.kw == 3 && g.ph == 1 && g.pw == 1) || (g.kh == 1 && g.kw == 1 && g.ph == 0 && g.pw == 0))) return false;
  if (g.Cin > 128 || g.Cout > 128 || g.Cin < 8 || g.Cout < 8) return false;
  if (g.N < 1) return false;
  return true;
}
static int pad_k(int64_t c) { return c <= 32 ? 32 : (c <= 64 ? 64 : 128); }

static void run_conv8(const Tensor* in, const Tensor* w, const Tensor* bias, Tensor* out, const ConvGeom& g, bool dgrad, hipStream_t st) {
  const int KS = g.kh, RS = KS * KS;
  const int CI = (int)(dgrad ? g.Cout : g.Cin), CO = (int)(dgrad ? g.Cin : g.Cout);
  const int KP = pad_k(CI);
  int64_t ps[1] = {(int64_t)RS * IG_M * KP};
  Hold wp(new_tensor(ps, 1, kBF16, in->device()));
  hipLaunchKernelGGL(ig_pack_weights_kernel, dim3(grid_for(ps[0], 256)), dim3(256), 0, st, w->ptr<bf16_t>(), wp->ptr<bf16_t>(), (int)g.Cout,
                     (int)g.Cin, KS, KP, dgrad ? 1 : 0);
  LAMP_LAUNCH_CHECK();
  const int NW = g.N >= 1024 ? 4 : 2;        // images per workgroup (keep >= 256 workgroups before widening)
  const size_t lds = (size_t)NW * 100 * KP * 2 + 2 * IG_WTILE;
  const int blocks = (int)((g.N + NW - 1) / NW);
  // fprop and dgrad are the SAME kernel (ig_conv8_kernel), so they share one timer class
  KernelTimer kt("conv_igemm_fprop_dgrad", conv_flops(g), conv_bytes(g, 2), st);
  const bf16_t* bp = bias ? bias->ptr<bf16_t>() : (const bf16_t*)nullptr;
#define IG_LAUNCH(KS_, NW_)                                                                                                        \
  do {                                                                                                                             \
    static bool attr = false;                                                                                                      \
    if (!attr) {                                                                                                                   \
      HIP_CHECK(hipFuncSetAttribute((const void*)ig_conv8_kernel<KS_, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      attr = true;                                                                                                                 \
    }                                                                                                                              \
    hipLaunchKernelGGL((ig_conv8_kernel<KS_, NW_>), dim3(blocks), dim3(NW_ * 128), lds, st, in->ptr<bf16_t>(), wp->ptr<bf16_t>(), bp, \
                       out->ptr<bf16_t>(), (int)g.N, CI, KP, CO);                                                                  \
  } while (0)
  if (KS == 3) { if (NW == 4) IG_LAUNCH(3, 4); else IG_LAUNCH(3, 2); }
  else { if (NW == 4) IG_LAUNCH(1, 4); else IG_LAUNCH(1, 2); }
#undef IG_LAUNCH
  LAMP_LAUNCH_CHECK();
}

bool igemm_conv_fwd(const Tensor* x, const Tensor* w, const Tensor* bias, Tensor* y, const ConvGeom& g, hipStream_t st) {
  if (!ig_qualifies(g, x->dtype)) return false;
  run_conv8(x, w, bias, y, g, false, st);
  return true;
}
bool igemm_conv_dgrad(const Tensor* dy, const Tensor* w, Tensor* dx, const ConvGeom& g, hipStream_t st) {
  if (!ig_qualifies(g, dy->dtype)) return false;
  run_conv8(dy, w, nullptr, dx, g, true, st);
  return true;
}
bool igemm_conv_wgrad(const Tensor* dy, const Tensor* x, Tensor* dw, const ConvGeom& g, hipStream_t st) {
  if (!ig_qualifies(g, x->dtype)) return false;
  const int KS = g.kh, RS = KS * KS;
  {
    // v2: workgroup = (32-channel slice of Cin, image range), all taps in registers
    const int ntile = (int)((g.Cin + WG_CI - 1) / WG_CI);
    const int CIP = ntile * WG_CI;
    static const int wgs_per_cu = [] { const char* e = getenv("LAMP_WGRAD_WGS_PER_CU"); return e ? std::max(1, atoi(e)) : 2; }();
    int target = std::max(1, (num_cus() * wgs_per_cu) / ntile);      // ~2 workgroups per CU
    int ips = (int)std::max<int64_t>(1, (g.N + target - 1) / target);
    if (ips < 8 && g.N >= 8) ips = 8;
    const int nsplit = (int)((g.N + ips - 1) / ips);
    int64_t ps[1] = {(int64_t)nsplit * RS * IG_M * CIP};
    Hold partial(new_tensor(ps, 1, kF32, x->device()));
    const size_t lds = 2 * WG_STAGE;
    {
      KernelTimer kt("conv_wgrad_igemm", conv_flops(g), conv_bytes(g, 2), st);
      static bool a3 = false, a1 = false;
      if (KS == 3) {
        if (!a3) { HIP_CHECK(hipFuncSetAttribute((const void*)ig_wgrad8v2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); a3 = true; }
        hipLaunchKernelGGL((ig_wgrad8v2_kernel<3>), dim3(ntile, nsplit), dim3(256), lds, st, dy->ptr<bf16_t>(), x->ptr<bf16_t>(), partial->ptr<float>(),
                           (int)g.N, (int)g.Cout, (int)g.Cin, CIP, ips);
      } else {
        if (!a1) { HIP_CHECK(hipFuncSetAttribute((const void*)ig_wgrad8v2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); a1 = true; }
        hipLaunchKernelGGL((ig_wgrad8v2_kernel<1>), dim3(ntile, nsplit), dim3(256), lds, st, dy->ptr<bf16_t>(), x->ptr<bf16_t>(), partial->ptr<float>(),
                           (int)g.N, (int)g.Cout, (int)g.Cin, CIP, ips);
      }
      LAMP_LAUNCH_CHECK();
    }
    const int64_t cols = (int64_t)RS * IG_M * CIP / 4;
    hipLaunchKernelGGL(ig_wgrad_reduce_v2_kernel, dim3((unsigned)((cols + 31) / 32)), dim3(256), 0, st, partial->ptr<float>(), dw->ptr<bf16_t>(), (int)g.Cout,
                       (int)g.Cin, CIP, RS, nsplit);
    LAMP_LAUNCH_CHECK();
    return true;
  }
  // (v1, kept for reference: one tap per workgroup) enough workgroups to fill 256 CUs: RS taps x nsplit image ranges
  int target_splits = (2 * num_cus() + RS - 1) / RS;
  int ips = (int)std::max<int64_t>(1, (g.N + target_splits - 1) / target_splits);
  if (ips < 8 && g.N >= 8) ips = 8;
  const int nsplit = (int)((g.N + ips - 1) / ips);
  int64_t ps[1] = {(int64_t)nsplit * RS * IG_M * IG_M};
  Hold partial(new_tensor(ps, 1, kF32, x->device()));
  const size_t lds = 4 * IG_WTILE;
  {
    KernelTimer kt("conv_wgrad_igemm", conv_flops(g), conv_bytes(g, 2), st);
    if (KS == 3) hipLaunchKernelGGL((ig_wgrad8_kernel<3>), dim3(nsplit * RS), dim3(256), lds, st, dy->ptr<bf16_t>(), x->ptr<bf16_t>(),
                                    partial->ptr<float>(), (int)g.N, (int)g.Cout, (int)g.Cin, ips);
    else hipLaunchKernelGGL((ig_wgrad8_kernel<1>), dim3(nsplit * RS), dim3(256), lds, st, dy->ptr<bf16_t>(), x->ptr<bf16_t>(),
                            partial->ptr<float>(), (int)g.N, (int)g.Cout, (int)g.Cin, ips);
    LAMP_LAUNCH_CHECK();
  }
  const int total = (int)(g.Cout * g.Cin * RS);
  hipLaunchKernelGGL(ig_wgrad_reduce_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, partial->ptr<float>(), dw->ptr<bf16_t>(), (int)g.Cout,
                     (int)g.Cin, RS, nsplit);
  LAMP_LAUNCH_CHECK();
  return true;
}

}  // namespace lamp
