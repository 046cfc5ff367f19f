// Implicit-GEMM convolution in f32 and f64 on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64: exact IEEE fused
// multiply-adds, the result is a k-ordered fma chain) for the wide layers of the CIFAR ResNet - the precisions the reference's example
// runs in (example-cifar100/src/main/scala/lamp/example/cifar/cifar100.scala:127-129: DoublePrecision unless --single; model
// cnn.scala:89-137, operator ops.scala:1547-1651).
//
// Scope: f32 / f64, NCHW, H = W = 8, kernel 3x3 (pad 1) or 1x1 (pad 0), stride 1, dilation 1, groups 1, Cin and Cout <= 128 and multiples of 4.
//
// The f32 matrix pipe runs at 1/16 of the bf16 rate (256 FLOP per clock and CU; f64: 128), so these kernels are bound by MFMA issue and by
// nothing else: a 128 -> 128 3x3 layer needs 74 k matrix cycles per image against 64 KiB of activations.  The design therefore spends nothing
// on data movement cleverness and everything on keeping the pipe busy:
//  * fprop and dgrad are ONE kernel (dgrad = fprop of dY with the weights transposed and the taps mirrored).  A workgroup owns four images
//    (f64: two); wave (image, part) multiplies the 64 pixels of its image with one part of the output-channel tiles (f32: halves, 4 + 4 or
//    4 + 3 for 100 channels; f64: quarters) - the two waves of a SIMD fill each other's gaps.  D rows = pixels, D columns = output channels.
//  * The images sit in LDS exactly as they sit in memory (NCHW rows of 64 values), brought in by LDS-DMA with no register staging and no
//    transposition: the pixel operand of an MFMA is ONE value per lane (A[pixel = lane & 15][k = lane >> 4]), i.e. a ds_read_b32 / _b64,
//    which needs no more alignment than the element: the tap shift is a byte offset, pixels outside the image are zeroed in registers.
//    A DMA piece holds four channels (k-step j of a 16-channel chunk = piece j, lane group q = its row q); pieces are 16 elements further
//    apart than they are long: the spare bytes take the reads past the last row.
//  * K runs over 16-channel chunks (outer) and taps (inner).  Every wave fetches its own weight fragments straight from L2 - packed
//    [tap][chunk][128 rows][16], so the values a lane needs (k-steps 4 j + q of its row) are contiguous - one stage ahead into a second
//    register set: no LDS ring and NO BARRIER in the main loop (with a ring shared by the workgroup a stage took 4700 - 4800 cycles for
//    4096 of matrix work; now 4140).  A last chunk that is only partly filled (Cin = 100: 4 of 16) runs only the k-steps that hold channels.
//  * wgrad: dW[tap][co][ci] = sum over images and pixels of dY[co][p] X[ci][p + shift(tap)], K = pixels.  Workgroup = (16-channel tile
//    of Cin, image range), wave = 16-channel tile of Cout x all nine taps (36 accumulator values per lane).  dY arrives by LDS-DMA ([co][64 px],
//    the 4-pixel groups XOR-swizzled by the row), X through registers into rows of 65 values.  Per image a lane reads four dY fragments and
//    the 34 X values its k-slots can meet under any tap (X[ci][16q - 9 .. 16q + 24]) ONCE; every MFMA then takes its operands straight
//    from those registers.  Taps whose column falls outside the image for a whole k-step are skipped (132 instead of 144 MFMAs per image).
//    The partial sums per image range are reduced by the batched kernel of wgrad_reduce.hip (f32) or by a small kernel here (f64).
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include "device_utils.h"
#include "conv_geom.h"
#include "wgrad_reduce.h"

namespace lamp {

typedef float f4v __attribute__((ext_vector_type(4)));
typedef double d4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char_t;
typedef const __attribute__((address_space(1))) char glb_char_t;

constexpr int F_ROWS = 128;             // rows of a packed weight image (output channels, zero padded)

// per element type: images per workgroup, parts the output-channel tiles of an image are split into (8 waves = images x parts), the MFMA
template <class T> struct IgT;
template <> struct IgT<float> {
  using acc = f4v;
  static constexpr int NI = 4, NPART = 2;
  static __device__ __forceinline__ acc mfma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int drow(int q, int r) { return 4 * q + r; }      // D row of accumulator register r in lane group q
};
template <> struct IgT<double> {
  using acc = d4v;
  static constexpr int NI = 2, NPART = 4;
  static __device__ __forceinline__ acc mfma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int drow(int q, int r) { return q + 4 * r; }      // the f64 MFMA's own D layout
};
template <class T> constexpr int ig_piece_bytes() { return 4 * 64 * (int)sizeof(T); }                 // four channels of one image
template <class T> constexpr int ig_pstr() { return ig_piece_bytes<T>() + 16 * (int)sizeof(T); }     // + room for nine pixels past a row

// ---- weight packing ---------------------------------------------------------------------------------
// fprop: wp[rs][kc][co][16] <- W[co][ci][r][s]            (rows = Cout, k = Cin in chunks of 16, KPf = round16(Cin))
// dgrad: wp[rs][kc][ci][16] <- W[co][ci][kh-1-r][kw-1-s]  (rows = Cin,  k = Cout, KPd = round16(Cout))
constexpr int F_PACK_MAX = 16;
template <class T> struct PackManyT { const T* w[F_PACK_MAX]; T* wp[F_PACK_MAX]; int Cout[F_PACK_MAX], Cin[F_PACK_MAX], KS[F_PACK_MAX], KPf[F_PACK_MAX], KPd[F_PACK_MAX]; };
template <class T>
__global__ void ig32_pack_weights_many_kernel(PackManyT<T> a) {
  const int t = blockIdx.y;
  const T* __restrict__ w = a.w[t];
  T* __restrict__ wp = a.wp[t];
  const int Cout = a.Cout[t], Cin = a.Cin[t], KS = a.KS[t], KPf = a.KPf[t], KPd = a.KPd[t];
  const int RS = KS * KS;
  const int nf = RS * F_ROWS * KPf, total = nf + RS * F_ROWS * KPd;
  for (int e0 = blockIdx.x * blockDim.x + threadIdx.x; e0 < total; e0 += gridDim.x * blockDim.x) {
    const int dgrad = e0 >= nf;
    const int e = dgrad ? e0 - nf : e0, KP = dgrad ? KPd : KPf;
    // [tap][16-channel chunk][128 rows][16]: the 16 x 16 tile one lane group of one wave multiplies is contiguous memory
    const int pos = e & 15, row = (e >> 4) % F_ROWS, kc = (e / (16 * F_ROWS)) % (KP >> 4), rs = e / (KP * F_ROWS);
    // position 4 q + j of a chunk holds channel 4 j + q: a lane's load (q) then delivers, as k-step j, a channel of [4 j, 4 j + 4) -
    // the k-steps of a partly filled last chunk that hold no channel at all can be skipped
    const int k = kc * 16 + 4 * (pos & 3) + (pos >> 2);
    const int r = rs / KS, s = rs % KS;
    T v = T(0);
    if (!dgrad) { if (row < Cout && k < Cin) v = w[((row * Cin + k) * KS + r) * KS + s]; }
    else { if (row < Cin && k < Cout) v = w[((k * Cin + row) * KS + (KS - 1 - r)) * KS + (KS - 1 - s)]; }
    wp[e0] = v;
  }
}

// ---- fprop / dgrad -----------------------------------------------------------------------------------
// x [N][CI][64], wp [RS][KP / 16][128][16], y [N][CO][64]; KP = round16(CI).  NCT = 16-channel tiles of the output; SPLITPX: the waves of
// an image split its PIXELS instead of the output channels (one tile of output channels: dgrad into a 16-channel layer).
// diagnostic build only (-DIG32_STAMP): shader-clock and 100 MHz stamps of wave 0 of every workgroup (scripts/conv_f32_stamp_probe.py)
#ifdef IG32_STAMP
__device__ unsigned long long ig32_stamps[8 * 1024];
__device__ unsigned long long ig32_rt_stamps[8 * 1024];
#define IG32_STAMP_AT(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) { ig32_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
                                                                           ig32_rt_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
extern "C" int lamp_debug_ig32_stamps(unsigned long long* out, unsigned long long* out_rt) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ig32_stamps), sizeof(ig32_stamps)) != hipSuccess) return 1;
  return hipMemcpyFromSymbol(out_rt, HIP_SYMBOL(ig32_rt_stamps), sizeof(ig32_rt_stamps)) == hipSuccess ? 0 : 1;
}
#else
#define IG32_STAMP_AT(k) do { } while (0)
#endif

// Welford triple (count, mean, M2) of the 16 output values a lane holds for one output channel of its image, merged over the four lane
// groups that share the channel: the statistics of the image's 64 pixels of that channel, from the values the batch norm will read
// (as conv_igemm.hip's epilogue publishes them: norm.hip merges the per-image triples and skips its statistics pass)
__device__ __forceinline__ void ig32_stats_wave(const float (&v)[16], float& n, float& mean, float& m2) {
  const float sh = v[0];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < 16; k++) { const float d = v[k] - sh; s1 += d; s2 += d * d; }
  n = 16.f; mean = sh + s1 * (1.f / 16.f); m2 = s2 - s1 * s1 * (1.f / 16.f);
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const float n2 = __shfl_xor(n, off, 64), mean2 = __shfl_xor(mean, off, 64), m22 = __shfl_xor(m2, off, 64);
    const float nt = n + n2, d = mean2 - mean, f = n2 / nt;
    mean = mean + d * f;
    m2 = m2 + m22 + d * d * n * f;
    n = nt;
  }
}

template <class T, int KS, int NCT, bool SPLITPX>
__global__ __launch_bounds__(512) void ig32_conv8_kernel(const T* __restrict__ x, const T* __restrict__ wp, const T* __restrict__ bias,
                                                         T* __restrict__ y, int N, int CI, int KP, int CO, const T* __restrict__ addend,
                                                         float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using TR = IgT<T>;
  using acc_v = typename TR::acc;
  constexpr int ES = (int)sizeof(T);
  constexpr int NI = TR::NI, NPART = TR::NPART;
  constexpr int PSTR = ig_pstr<T>(), PB = ig_piece_bytes<T>();
  constexpr int RS = KS * KS;
  constexpr int PAD = (KS - 1) / 2;
  constexpr int CT0 = SPLITPX ? NCT : (NCT + NPART - 1) / NPART;   // channel tiles of a wave (the loop bound of every part)
  constexpr int PT = SPLITPX ? 4 / NPART : 4;                      // pixel tiles (two image rows each) per wave
  constexpr bool ALLFULL = SPLITPX || NCT % NPART == 0;            // every wave has CT0 real tiles: no test in the MFMA chain
  const int KC = KP >> 4;                               // 16-channel chunks of K
  const int NP = KC * 4;                                // 4-channel pieces per image
  const int XIMG = NP * PSTR;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = wid % NI, part = wid / NI;
  char* Xl = smem + 16 * ES;                            // [NI images][NP pieces][PSTR]; taps outside the image read up to nine pixels before /
                                                        // after a row (spare bytes, neighbouring row, padding) and are zeroed in registers
  const int n0 = blockIdx.x * NI;
  const int NT = KC * RS;
  const int m = lane & 15, q = lane >> 4;
  const int nj_last = (CI - 16 * (KC - 1) + 3) >> 2;    // k-steps of the last chunk that hold channels

  // this wave's tiles
  const int ct_first = SPLITPX ? 0 : part * CT0;
  const int ct_count = SPLITPX ? NCT : max(0, min(CT0, NCT - part * CT0));
  const int jt0 = SPLITPX ? part * PT : 0;

  IG32_STAMP_AT(0);
  // images: piece pid = (image, 4 channels) is contiguous memory; pieces of channels that do not exist (CI < KP) are zeros
  for (int pid = wid; pid < NI * NP; pid += 8) {
    const int im = pid / NP, pc = pid - im * NP;
    char* dst = Xl + im * XIMG + pc * PSTR;
    if (pc * 4 < CI) {
      const int n = min(n0 + im, N - 1);                 // images beyond the batch: a copy of the last one, never stored
      const char* src = reinterpret_cast<const char*>(x + ((int64_t)n * CI + pc * 4) * 64) + lane * 16;
#pragma unroll
      for (int u = 0; u < PB / 1024; u++) __builtin_amdgcn_global_load_lds((glb_char_t*)(src + u * 1024), (lds_char_t*)(dst + u * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < PB / 1024; u++) *reinterpret_cast<uint4*>(dst + u * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
    }
  }

  acc_v acc[CT0][PT];
#pragma unroll
  for (int i = 0; i < CT0; i++)
#pragma unroll
    for (int j = 0; j < PT; j++) acc[i][j] = acc_v{0, 0, 0, 0};

  // pixel operand: lane (m, q) of pixel tile jt, k-step j, tap (r, s) reads channel 16 kc + 4 j + q (row q of piece 4 kc + j) at pixel
  // 16 jt + m + 8 (r - PAD) + (s - PAD)
  const char* xq = Xl + img * XIMG + q * 64 * ES + (jt0 * 16 + m) * ES;
  const bool col_lo = (m & 7) == 0, col_hi = (m & 7) == 7, row_lo = m < 8, row_hi = m >= 8;
  // weight operand, straight from memory (L2: every workgroup reads the same 0.6 MB): the four values of row 16 (tile) + m, lane group q, of
  // stage (tap, chunk) - k-step j of them is channel 4 j + q (ig32_pack_weights_many_kernel).  No LDS ring, no barrier in the main loop:
  // the waves of a workgroup share nothing but the read-only images
  const char* wq = reinterpret_cast<const char*>(wp) + ((ct_first * 16 + m) * 16 + q * 4) * ES;

  T fx[2][PT][4];
  acc_v fw[2][CT0];
  auto load_frags = [&](int kc1, int rs, int set) {
    const int r = rs / KS, s = rs - r * KS;
    const char* wb = wq + (int64_t)(rs * KC + kc1) * (F_ROWS * 16 * ES);
#pragma unroll
    for (int i = 0; i < CT0; i++) fw[set][i] = *reinterpret_cast<const acc_v*>(wb + i * (256 * ES));
    const char* xb = xq + kc1 * 4 * PSTR;
#pragma unroll
    for (int jt = 0; jt < PT; jt++)
#pragma unroll
      for (int j = 0; j < 4; j++)
        fx[set][jt][j] = *reinterpret_cast<const T*>(xb + j * PSTR + (16 * jt + 8 * (r - PAD) + (s - PAD)) * ES);
  };
  auto zero_edges = [&](int rs, int set) {
    if (KS != 3) return;
    const int r = rs / KS, s = rs - r * KS;
    const bool colout = (s == 0 && col_lo) || (s == 2 && col_hi);
#pragma unroll
    for (int jt = 0; jt < PT; jt++) {
      const bool out = colout || (r == 0 && jt0 + jt == 0 && row_lo) || (r == 2 && jt0 + jt == 3 && row_hi);
      if (s != 1 || r != 1) {
#pragma unroll
        for (int j = 0; j < 4; j++) fx[set][jt][j] = out ? T(0) : fx[set][jt][j];
      }
    }
  };

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                        // the images are in LDS: the only barrier of the kernel
  load_frags(0, 0, 0);
  IG32_STAMP_AT(1);

  // Stage t = (kc, rs) on register set t & 1 (RS is odd: t = kc RS + rs has the parity of kc + rs): request the fragments of stage t + 1
  // into the other set (weights from L2, pixels from LDS), select the out-of-image lanes of this stage's pixels to zero, multiply.  Every
  // wave runs on its own: while one waits for its loads or selects, the other wave of its SIMD feeds the matrix pipe.
  // (Measured, f32 128 -> 128 3x3, B = 2048: with the weights in a two-slot LDS-DMA ring shared by the workgroup - one or two barriers per
  //  stage, the halves in step or half a stage apart - the main loop took 4700 - 4800 cycles per stage against 4096 of matrix work.)
  auto tap_loop = [&](int kc, auto par0c) {
    constexpr int par0 = decltype(par0c)::value;
    const int nj = kc == KC - 1 ? nj_last : 4;
#pragma unroll
    for (int rs = 0; rs < RS; rs++) {
      const int cur = (par0 + rs) & 1;
      const int t = kc * RS + rs;
      const int rs1 = (rs + 1) % RS;
      const int kc1 = kc + (rs + 1) / RS;
      if (t + 1 < NT) load_frags(kc1, rs1, cur ^ 1);
      zero_edges(rs, cur);
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (j < nj) {
#pragma unroll
          for (int i = 0; i < CT0; i++)
            if (ALLFULL || i < ct_count) {
#pragma unroll
              for (int jt = 0; jt < PT; jt++) acc[i][jt] = TR::mfma(fx[cur][jt][j], fw[cur][i][j], acc[i][jt]);
            }
        }
    }
  };
  {
    int kc = 0;
    for (; kc + 1 < KC; kc += 2) {
      tap_loop(kc, std::integral_constant<int, 0>{});
      tap_loop(kc + 1, std::integral_constant<int, 1>{});
    }
    if (kc < KC) tap_loop(kc, std::integral_constant<int, 0>{});
  }
  IG32_STAMP_AT(2);

  // epilogue: lane (m, q) holds output channel 16 tile + m and, in accumulator register r, pixel 16 jt + drow(q, r) of its image (f32: four
  // consecutive pixels, one 16-byte store).  Every tile's bias value is requested up front: one memory round trip instead of one per tile
  const int n = n0 + img;
  if (n < N) {
    T* yp = y + (int64_t)n * CO * 64;
    const T* ap = addend ? addend + (int64_t)n * CO * 64 : nullptr;
    T bv[CT0];
#pragma unroll
    for (int i = 0; i < CT0; i++) {
      const int co = (ct_first + i) * 16 + m;
      bv[i] = (bias && co < CO) ? bias[co] : T(0);
    }
#pragma unroll
    for (int i = 0; i < CT0; i++) {
      const int co = (ct_first + i) * 16 + m;
      if (i < ct_count && co < CO) {
        if constexpr (std::is_same<T, float>::value) {
          f4v av[PT];
          if (ap) {
#pragma unroll
            for (int jt = 0; jt < PT; jt++) av[jt] = *reinterpret_cast<const f4v*>(ap + co * 64 + (jt0 + jt) * 16 + q * 4);
          }
          float vals[16];
#pragma unroll
          for (int jt = 0; jt < PT; jt++) {
            const int off = co * 64 + (jt0 + jt) * 16 + q * 4;
            f4v v = acc[i][jt] + f4v{bv[i], bv[i], bv[i], bv[i]};
            if (ap) v += av[jt];
            *reinterpret_cast<f4v*>(yp + off) = v;
            if (!SPLITPX) { vals[4 * jt + 0] = v[0]; vals[4 * jt + 1] = v[1]; vals[4 * jt + 2] = v[2]; vals[4 * jt + 3] = v[3]; }
          }
          if (!SPLITPX && stats) {                // one partial per image and channel, [channel][image][3]
            float wn, wm, w2;
            ig32_stats_wave(vals, wn, wm, w2);
            if (q == 0) { float* sp = stats + ((int64_t)co * N + n) * 3; sp[0] = wn; sp[1] = wm; sp[2] = w2; }
          }
        } else {
#pragma unroll
          for (int jt = 0; jt < PT; jt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int off = co * 64 + (jt0 + jt) * 16 + TR::drow(q, r);
              T v = acc[i][jt][r] + bv[i];
              if (ap) v += ap[off];
              yp[off] = v;
            }
        }
      }
    }
  }
  IG32_STAMP_AT(3);
#ifdef IG32_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  IG32_STAMP_AT(4);
#endif
}

// ---- wgrad ---------------------------------------------------------------------------------------------
template <class T> constexpr int fw_dy_bytes() { return 128 * 64 * (int)sizeof(T); }                  // dY tile of one image: [128 co][64 px]
template <class T> constexpr int fw_xrow() { return 65 * (int)sizeof(T); }                             // X row: 64 px + one element (lanes = channels: distinct banks)
template <class T> constexpr int fw_x_bytes() { return 16 * fw_xrow<T>() + 16 * (int)sizeof(T); }      // X tile (16 channels) + room for the reads beyond the last row
template <class T> constexpr int fw_stage_bytes() { return fw_dy_bytes<T>() + fw_x_bytes<T>(); }       // f32 36,992 bytes, f64 73,984

template <class T, int KS>
__global__ __launch_bounds__(512) void ig32_wgrad8_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ partial,
                                                          int N, int CO, int CI, int CIP, int images_per_split, int ntile) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using TR = IgT<T>;
  using acc_v = typename TR::acc;
  constexpr int ES = (int)sizeof(T);
  constexpr int RS = KS * KS, PAD = (KS - 1) / 2;
  constexpr int NXV = KS == 3 ? 34 : 16;                      // X values a lane can meet: offsets -9 .. 24 (3x3) or 0 .. 15 (1x1)
  constexpr int XOFF = KS == 3 ? 9 : 0;
  constexpr int FW_DY = fw_dy_bytes<T>(), FW_XROW = fw_xrow<T>(), FW_X = fw_x_bytes<T>(), FW_STAGE = fw_stage_bytes<T>();
  constexpr int RB = 64 * ES;                                 // bytes of a dY row
  constexpr int CPR = RB / 16;                                // 16-byte chunks per row (f32 16, f64 32)
  constexpr int RPP = 1024 / RB;                              // rows per 1 KiB DMA piece
  constexpr int NPIECE = FW_DY / 1024 / 8;                    // DMA pieces per wave and image
  constexpr int GSH = ES == 4 ? 0 : 1;                        // log2 of the chunks in a 4-pixel group: a lane's fragment
  const int nsplit = gridDim.x / ntile;
  int tile, split;
  {                                                           // XCD-aware: the tiles of one image range share an L2 (dY is read ntile times)
    const int b = blockIdx.x;
    if ((nsplit & 7) == 0) { const int xcd = b & 7, slot = b >> 3; tile = slot % ntile; split = xcd + 8 * (slot / ntile); }
    else { tile = b % ntile; split = b / ntile; }
  }
  const int ci0 = tile * 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // = this wave's tile of output channels
  const int m = lane & 15, q = lane >> 4;
  const int nbeg = split * images_per_split, nend = min(nbeg + images_per_split, N);
  const bool active = wid * 16 < CO;                          // tiles of padding are not multiplied

  // dY by LDS-DMA: LDS position (row, group') holds the source group group' ^ (row & 15), so the 16 rows a fragment read touches land on
  // different banks
  auto dma_dy = [&](int n, char* stage) {
    const char* base = reinterpret_cast<const char*>(dy + (int64_t)n * CO * 64);
#pragma unroll
    for (int i = 0; i < NPIECE; i++) {
      const int piece = wid * NPIECE + i;
      const int row = piece * RPP + lane / CPR;
      const int rowc = min(row, CO - 1);                      // rows of padding: copies of the last real row (their products are never stored)
      const int cpos = lane % CPR;                            // chunk position inside the LDS row
      const int chunk = (((cpos >> GSH) ^ (row & 15)) << GSH) | (cpos & ((1 << GSH) - 1));
      __builtin_amdgcn_global_load_lds((glb_char_t*)(base + rowc * RB + chunk * 16), (lds_char_t*)(stage + piece * 1024), 16, 0, 0);
    }
  };
  // X through registers: 16 channels x 64 pixels, 16 bytes per thread (f32: threads 0..255, f64: all 512)
  constexpr int XT = 16 * CPR;                                // threads that carry a packet
  const bool xthread = tid < XT;
  const int xc = (tid % XT) / CPR, xp = tid % CPR;
  auto load_x = [&](int n) -> uint4 {
    if (xthread && ci0 + xc < CI) return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x + ((int64_t)n * CI + ci0 + xc) * 64) + xp * 16);
    return make_uint4(0, 0, 0, 0);
  };
  auto store_x = [&](char* stage, uint4 v) {
    if (!xthread) return;
    char* d = stage + FW_DY + xc * FW_XROW + xp * 16;
    if constexpr (ES == 4) { unsigned* u = reinterpret_cast<unsigned*>(d); u[0] = v.x; u[1] = v.y; u[2] = v.z; u[3] = v.w; }
    else { uint2* u = reinterpret_cast<uint2*>(d); u[0] = make_uint2(v.x, v.y); u[1] = make_uint2(v.z, v.w); }
  };

  acc_v acc[RS];
#pragma unroll
  for (int t = 0; t < RS; t++) acc[t] = acc_v{0, 0, 0, 0};

  auto compute = [&](const char* st) {
    // k-slot (m4, j) of lane group q is pixel 4 m4 + 16 q + j
    acc_v fa[4];
#pragma unroll
    for (int m4 = 0; m4 < 4; m4++) fa[m4] = *reinterpret_cast<const acc_v*>(st + (wid * 16 + m) * RB + (((m4 + 4 * q) ^ m) * (4 * ES)));
    T xv[NXV];
    const char* xs = st + FW_DY + m * FW_XROW + (16 * q - XOFF) * ES;
#pragma unroll
    for (int k = 0; k < NXV; k++) xv[k] = *reinterpret_cast<const T*>(xs + k * ES);
    if (KS == 3) {
      // offsets below 0 / above 15 are the previous / next two image rows: outside the image for the first / last lane group
#pragma unroll
      for (int k = 0; k < 9; k++) xv[k] = q == 0 ? T(0) : xv[k];
#pragma unroll
      for (int k = 25; k < 34; k++) xv[k] = q == 3 ? T(0) : xv[k];
    }
#pragma unroll
    for (int m4 = 0; m4 < 4; m4++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int t = 0; t < RS; t++) {
          const int r = t / KS, s = t % KS;
          // the column of this k-step's pixels is w0 = 4 (m4 & 1) + j for every lane: taps that leave the image row add nothing
          const int w0 = 4 * (m4 & 1) + j + (s - PAD);
          if (w0 < 0 || w0 > 7) continue;
          const int off = 4 * m4 + j + 8 * (r - PAD) + (s - PAD) + XOFF;
          acc[t] = TR::mfma(fa[m4][j], xv[off], acc[t]);
        }
  };

  uint4 xr = make_uint4(0, 0, 0, 0);
  // zero the X regions once (the spare bytes behind the last row are read, then discarded: keep them finite anyway)
  for (int o = tid * 4; o < FW_X; o += 512 * 4) { *reinterpret_cast<unsigned*>(smem + FW_DY + o) = 0u; *reinterpret_cast<unsigned*>(smem + FW_STAGE + FW_DY + o) = 0u; }
  __syncthreads();
  if (nbeg < nend) {
    dma_dy(nbeg, smem);
    xr = load_x(nbeg);
    store_x(smem, xr);
    if (nbeg + 1 < nend) { dma_dy(nbeg + 1, smem + FW_STAGE); xr = load_x(nbeg + 1); }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int cur = 0;
  for (int n = nbeg; n < nend; n++, cur ^= 1) {
    char* st = smem + cur * FW_STAGE;
    char* nx = smem + (cur ^ 1) * FW_STAGE;
    if (active) compute(st);
    if (n + 1 < nend) store_x(nx, xr);                        // requested one image ago
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // stage cur is free, stage cur ^ 1 is complete
    if (n + 2 < nend) { dma_dy(n + 2, st); xr = load_x(n + 2); }
  }
  // partial[(split * RS + t)][128][CIP]: lane (m, q) holds, in register r, row (output channel) drow(q, r) of column (input channel) m
  if (active) {
#pragma unroll
    for (int t = 0; t < RS; t++) {
      T* out = partial + (int64_t)(split * RS + t) * F_ROWS * CIP;
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int co = wid * 16 + TR::drow(q, rr), ci = ci0 + m;
        if (co < CO && ci < CI) out[co * CIP + ci] = acc[t][rr];
      }
    }
  }
}
// f64 partial sums [split][tap][128][CIP] -> dW[co][ci][r][s], splits summed in order (the f32 ones go through wgrad_reduce.hip).  Threads walk
// the PARTIAL's layout (input channel fastest: coalesced reads of the nsplit x 1.2 MB), the 8-byte writes of dW are the scattered side
__global__ void ig64_wgrad_reduce_kernel(const double* __restrict__ partial, double* __restrict__ dw, int CO, int CI, int CIP, int RS, int nsplit) {
  const int64_t slice = (int64_t)RS * F_ROWS * CIP;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < slice; e += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(e % CIP), co = (int)((e / CIP) % F_ROWS), rs = (int)(e / ((int64_t)CIP * F_ROWS));
    if (ci >= CI || co >= CO) continue;
    // (the loads do not depend on the sum: sixteen in flight per thread - with one at a time the launch was 29 us of memory latency)
    double a = 0.0;
    int sp = 0;
    for (; sp + 16 <= nsplit; sp += 16) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = __builtin_nontemporal_load(partial + (int64_t)(sp + u) * slice + e);
#pragma unroll
      for (int u = 0; u < 16; u++) a += v[u];
    }
    for (; sp < nsplit; sp++) a += partial[(int64_t)sp * slice + e];
    dw[((int64_t)co * CI + ci) * RS + rs] = a;
  }
}

// ---- host ---------------------------------------------------------------------------------------------
static bool ig32_qualifies(const ConvGeom& g, int dtype) {
  static const bool on = [] { const char* e = getenv("LAMP_IGEMM_F32"); return !(e && e[0] == '0'); }();
  static const bool on64 = [] { const char* e = getenv("LAMP_IGEMM_F64"); return !(e && e[0] == '0'); }();
  if (!((on && dtype == kF32) || (on64 && dtype == kF64))) return false;
  if (g.groups != 1 || g.transposed) return false;
  if (g.H != 8 || g.W != 8 || g.Ho != 8 || g.Wo != 8) return false;
  if (g.sh != 1 || g.sw != 1 || g.dh != 1 || g.dw != 1) return false;
  if (!((g.kh == 3 && g.kw == 3 && g.ph == 1 && g.pw == 1) || (g.kh == 1 && g.kw == 1 && g.ph == 0 && g.pw == 0))) return false;
  if (g.Cin > 128 || g.Cout > 128 || g.Cin < 4 || g.Cout < 4 || (g.Cin & 3) || (g.Cout & 3)) return false;
  if (g.Cin <= 16 && g.Cout <= 16) return false;             // the narrow layers stay on the image-per-workgroup kernels (conv_small.hip)
  if (g.N < 1) return false;
  return true;
}
static int pad16(int64_t c) { return (int)((c + 15) / 16) * 16; }

// packed images [fprop | dgrad], cached per (weight storage, view, stream) while the storage's version is unchanged and re-packed in
// place by the optimiser (see conv_igemm.hip: the same discipline, a separate cache because the element type differs)
namespace {
struct PackKey32 {
  uint64_t uid; int64_t offset; int KS, Cout, Cin, dtype; hipStream_t st;
  bool operator<(const PackKey32& o) const { return std::tie(uid, offset, KS, Cout, Cin, dtype, st) < std::tie(o.uid, o.offset, o.KS, o.Cout, o.Cin, o.dtype, o.st); }
};
struct PackVal32 { uint64_t version; Tensor* packed; uint64_t tick;  bool pinned = false; };
std::mutex g_pack32_mu;
std::map<PackKey32, PackVal32> g_pack32_cache;
uint64_t g_pack32_tick = 0;
}  // namespace

template <class T> static void launch_pack32(const T* w, T* wp, int Cout, int Cin, int KS, hipStream_t st) {
  PackManyT<T> a;
  a.w[0] = w; a.wp[0] = wp; a.Cout[0] = Cout; a.Cin[0] = Cin; a.KS[0] = KS; a.KPf[0] = pad16(Cin); a.KPd[0] = pad16(Cout);
  const int total = KS * KS * F_ROWS * (a.KPf[0] + a.KPd[0]);
  hipLaunchKernelGGL((ig32_pack_weights_many_kernel<T>), dim3((unsigned)std::min(512, (total + 255) / 256), 1u), dim3(256), 0, st, a);
  LAMP_LAUNCH_CHECK();
}

static Tensor* packed_weights32(const Tensor* w, const ConvGeom& g, int KS, hipStream_t st, int64_t* dgrad_offset) {
  const int RS = KS * KS;
  const int KPf = pad16(g.Cin), KPd = pad16(g.Cout);
  const int64_t nf = (int64_t)RS * F_ROWS * KPf, nd = (int64_t)RS * F_ROWS * KPd;
  *dgrad_offset = nf;
  static const bool cache_on = [] { const char* e = getenv("LAMP_PACK_CACHE"); return !(e && e[0] == '0'); }();
  const bool cacheable = cache_on && w->st->owned && !w->st->scratch;
  const PackKey32 key{w->st->uid, w->offset, KS, (int)g.Cout, (int)g.Cin, w->dtype, st};
  const uint64_t ver = w->st->version.load(std::memory_order_relaxed);
  if (cacheable) {
    std::lock_guard<std::mutex> lk(g_pack32_mu);
    auto it = g_pack32_cache.find(key);
    if (it != g_pack32_cache.end() && it->second.version == ver) {
      it->second.tick = ++g_pack32_tick;
      if (allocator_capturing()) it->second.pinned = true;
      return retain(it->second.packed);
    }
  }
  int64_t ps[1] = {nf + nd};
  Hold wp(new_tensor(ps, 1, w->dtype, w->device()));
  if (w->dtype == kF32) launch_pack32<float>(w->ptr<float>(), wp->ptr<float>(), (int)g.Cout, (int)g.Cin, KS, st);
  else launch_pack32<double>(w->ptr<double>(), wp->ptr<double>(), (int)g.Cout, (int)g.Cin, KS, st);
  if (cacheable) {
    std::lock_guard<std::mutex> lk(g_pack32_mu);
    auto it = g_pack32_cache.find(key);
    if (it != g_pack32_cache.end()) { release(it->second.packed); g_pack32_cache.erase(it); }
    if (g_pack32_cache.size() >= 256) {         // least recently used entry that no captured graph reads
      auto victim = g_pack32_cache.end();
      for (auto i = g_pack32_cache.begin(); i != g_pack32_cache.end(); ++i)
        if (!i->second.pinned && (victim == g_pack32_cache.end() || i->second.tick < victim->second.tick)) victim = i;
      if (victim != g_pack32_cache.end()) { release(victim->second.packed); g_pack32_cache.erase(victim); }
    }
    g_pack32_cache[key] = PackVal32{ver, retain(wp.get()), ++g_pack32_tick, allocator_capturing()};
  }
  return wp.take();
}

// the optimisers' hook (optim.hip): re-pack, in place and in one launch per element type, every f32 / f64 weight whose packed images are
// cached on this stream (the caller holds g_pack32_mu)
template <class T> static void repack_cached_t(lamp_tensor* const* params, int n, hipStream_t st, int dtype) {
  PackManyT<T> a;
  int cnt = 0, maxtotal = 0;
  std::vector<std::pair<PackKey32, uint64_t>> done;
  auto flush = [&] {                                    // one launch per F_PACK_MAX images; the loop goes on (it used to stop)
    if (cnt == 0) return;
    hipLaunchKernelGGL((ig32_pack_weights_many_kernel<T>), dim3((unsigned)std::min(512, (maxtotal + 255) / 256), (unsigned)cnt), dim3(256), 0, st, a);
    LAMP_LAUNCH_CHECK();
    cnt = 0; maxtotal = 0;
  };
  for (int i = 0; i < n; i++) {
    if (cnt == F_PACK_MAX) flush();
    const Tensor* w = params[i];
    if (!w || !w->is_device() || w->dtype != dtype || w->ndim != 4 || !w->st->owned || !w->is_contiguous()) continue;
    for (auto& kv : g_pack32_cache) {
      if (kv.first.uid != w->st->uid || kv.first.offset != w->offset || kv.first.st != st || kv.first.dtype != dtype) continue;
      if (kv.first.Cout != (int)w->sizes[0] || kv.first.Cin != (int)w->sizes[1] || kv.first.KS != (int)w->sizes[2]) continue;
      const int KS = kv.first.KS, RS = KS * KS, KPf = pad16(kv.first.Cin), KPd = pad16(kv.first.Cout);
      const int total = RS * F_ROWS * (KPf + KPd);
      if (kv.second.packed->numel() != total) continue;
      a.w[cnt] = w->ptr<T>(); a.Cout[cnt] = kv.first.Cout; a.Cin[cnt] = kv.first.Cin; a.KS[cnt] = KS; a.KPf[cnt] = KPf; a.KPd[cnt] = KPd;
      a.wp[cnt] = static_cast<T*>(kv.second.packed->raw());
      done.push_back({kv.first, w->st->version.load(std::memory_order_relaxed)});
      maxtotal = std::max(maxtotal, total);
      cnt++;
      break;
    }
  }
  flush();
  for (auto& d : done) {
    auto it = g_pack32_cache.find(d.first);
    if (it != g_pack32_cache.end()) { it->second.version = d.second; it->second.tick = ++g_pack32_tick; }
  }
}
void igemm32_repack_cached(lamp_tensor* const* params, int n, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_pack32_mu);
  if (g_pack32_cache.empty()) return;
  repack_cached_t<float>(params, n, st, kF32);
  repack_cached_t<double>(params, n, st, kF64);
}

template <class T>
static void run_conv8_t(const Tensor* in, const Tensor* w, const Tensor* bias, Tensor* out, const ConvGeom& g, bool dgrad, hipStream_t st, const Tensor* addend) {
  const int KS = g.kh;
  const int CI = (int)(dgrad ? g.Cout : g.Cin), CO = (int)(dgrad ? g.Cin : g.Cout);
  const int KP = pad16(CI);
  int64_t dgrad_off = 0;
  Hold wpk(packed_weights32(w, g, KS, st, &dgrad_off));
  const T* wpp = static_cast<const Tensor*>(wpk.get())->ptr<T>() + (dgrad ? dgrad_off : 0);
  const int nct = (CO + 15) / 16;
  constexpr int NI = IgT<T>::NI;
  const int blocks = (int)((g.N + NI - 1) / NI);
  const size_t lds = 16 * sizeof(T) + (size_t)NI * (KP / 4) * ig_pstr<T>();
  // fprop in f32: per-image batch-norm statistics of the output from the epilogue, handed to the batch norm that follows (LAMP_CONV_BN_STATS=0: off)
  static const bool bn_stats = [] { const char* e = getenv("LAMP_CONV_BN_STATS"); return !(e && e[0] == '0'); }();
  Hold statt;
  float* statp = nullptr;
  if (std::is_same<T, float>::value && bn_stats && !dgrad && g.N >= 2 && nct > 1) {
    int64_t ps[1] = {(int64_t)g.N * CO * 3};
    statt = Hold(new_tensor(ps, 1, kF32, in->device()));
    statp = statt->ptr<float>();
  }
  struct Publish { Hold& t; const Tensor* y; int P; ~Publish() { if (t.get()) conv_stats_publish(y, t.get(), P); } } publish{statt, out, (int)g.N};
  KernelTimer kt(std::is_same<T, float>::value ? "conv_igemm_fprop_dgrad_f32" : "conv_igemm_fprop_dgrad_f64", conv_flops(g), conv_bytes(g, sizeof(T)), st);
  const T* bp = bias ? bias->ptr<T>() : (const T*)nullptr;
  const T* ap = addend ? addend->ptr<T>() : (const T*)nullptr;
#define IG32_LAUNCH(KS_, NCT_, SP_)                                                                                                  \
  do {                                                                                                                               \
    allow_big_lds((const void*)ig32_conv8_kernel<T, KS_, NCT_, SP_>);                                                                \
    hipLaunchKernelGGL((ig32_conv8_kernel<T, KS_, NCT_, SP_>), dim3(blocks), dim3(512), lds, st, in->ptr<T>(), wpp, bp,              \
                       out->ptr<T>(), (int)g.N, CI, KP, CO, ap, statp);                                                              \
  } while (0)
#define IG32_BY_NCT(KS_)                                                                                                             \
  do {                                                                                                                               \
    if (nct <= 1) IG32_LAUNCH(KS_, 1, true);                                                                                         \
    else if (nct <= 2) IG32_LAUNCH(KS_, 2, false);                                                                                   \
    else if (nct <= 4) IG32_LAUNCH(KS_, 4, false);                                                                                   \
    else if (nct <= 6) IG32_LAUNCH(KS_, 6, false);                                                                                   \
    else if (nct <= 7) IG32_LAUNCH(KS_, 7, false);                                                                                   \
    else IG32_LAUNCH(KS_, 8, false);                                                                                                 \
  } while (0)
  if (KS == 3) IG32_BY_NCT(3); else IG32_BY_NCT(1);
#undef IG32_BY_NCT
#undef IG32_LAUNCH
  LAMP_LAUNCH_CHECK();
}

bool igemm32_conv_fwd(const Tensor* x, const Tensor* w, const Tensor* bias, Tensor* y, const ConvGeom& g, hipStream_t st) {
  if (!ig32_qualifies(g, x->dtype)) return false;
  if (x->dtype == kF32) run_conv8_t<float>(x, w, bias, y, g, false, st, nullptr);
  else run_conv8_t<double>(x, w, bias, y, g, false, st, nullptr);
  return true;
}
bool igemm32_conv_dgrad(const Tensor* dy, const Tensor* w, Tensor* dx, const ConvGeom& g, hipStream_t st, const Tensor* addend, bool* addend_fused) {
  if (addend_fused) *addend_fused = false;
  if (!ig32_qualifies(g, dy->dtype)) return false;
  if (dy->dtype == kF32) run_conv8_t<float>(dy, w, nullptr, dx, g, true, st, addend);
  else run_conv8_t<double>(dy, w, nullptr, dx, g, true, st, addend);
  if (addend_fused) *addend_fused = addend != nullptr;
  return true;
}
template <class T>
static void run_wgrad_t(const Tensor* dy, const Tensor* x, Tensor* dw, const ConvGeom& g, hipStream_t st) {
  const int KS = g.kh, RS = KS * KS;
  const int ntile = (int)((g.Cin + 15) / 16);
  const int CIP = ntile * 16;
  const int target = std::max(1, num_cus() / ntile);
  int ips = (int)std::max<int64_t>(1, (g.N + target - 1) / target);
  if (ips < 4 && g.N >= 4) ips = 4;
  const int nsplit = (int)((g.N + ips - 1) / ips);
  int64_t ps[1] = {(int64_t)nsplit * RS * F_ROWS * CIP};
  Hold partial(new_tensor(ps, 1, dw->dtype, x->device()));
  const size_t lds = 2 * (size_t)fw_stage_bytes<T>();
  {
    KernelTimer kt(std::is_same<T, float>::value ? "conv_wgrad_igemm_f32" : "conv_wgrad_igemm_f64", conv_flops(g), conv_bytes(g, sizeof(T)), st);
    if (KS == 3) {
      allow_big_lds((const void*)ig32_wgrad8_kernel<T, 3>);
      hipLaunchKernelGGL((ig32_wgrad8_kernel<T, 3>), dim3(ntile * nsplit), dim3(512), lds, st, dy->ptr<T>(), x->ptr<T>(), partial->ptr<T>(), (int)g.N,
                         (int)g.Cout, (int)g.Cin, CIP, ips, ntile);
    } else {
      allow_big_lds((const void*)ig32_wgrad8_kernel<T, 1>);
      hipLaunchKernelGGL((ig32_wgrad8_kernel<T, 1>), dim3(ntile * nsplit), dim3(512), lds, st, dy->ptr<T>(), x->ptr<T>(), partial->ptr<T>(), (int)g.N,
                         (int)g.Cout, (int)g.Cin, CIP, ips, ntile);
    }
    LAMP_LAUNCH_CHECK();
  }
  if constexpr (std::is_same<T, float>::value) {
    const int64_t cols = (int64_t)RS * F_ROWS * CIP / 4;
    WgradReduceArgs ra{};
    ra.kind = 0; ra.CO = (int)g.Cout; ra.CI = (int)g.Cin; ra.CIP = CIP; ra.COP = F_ROWS; ra.RS = RS; ra.nsplit = nsplit; ra.blocks = (int)((cols + 31) / 32);
    ra.dw_f32 = 1;
    wgrad_reduce_enqueue(ra, partial.get(), dw, st);
  } else {
    const int64_t total = (int64_t)RS * F_ROWS * CIP;
    hipLaunchKernelGGL(ig64_wgrad_reduce_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, partial->ptr<double>(), dw->ptr<double>(), (int)g.Cout, (int)g.Cin, CIP, RS,
                       nsplit);
    LAMP_LAUNCH_CHECK();
  }
}
bool igemm32_conv_wgrad(const Tensor* dy, const Tensor* x, Tensor* dw, const ConvGeom& g, hipStream_t st) {
  if (!ig32_qualifies(g, x->dtype)) return false;
  if (x->dtype == kF32) run_wgrad_t<float>(dy, x, dw, g, st); else run_wgrad_t<double>(dy, x, dw, g, st);
  return true;
}

}  // namespace lamp
