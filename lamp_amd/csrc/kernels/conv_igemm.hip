// Implicit-GEMM convolution on the gfx950 matrix cores for the wide layers of the CIFAR ResNet
// (reference workload: example-cifar100/src/main/scala/lamp/example/cifar/cnn.scala:89-137 - res3/res4
// blocks: 3x3 stride-1 pad-1 and 1x1 convolutions on 8x8 maps with 16/100/128 channels, which are
// 97 % of the network's FLOPs; reference op: ops.scala:1547-1651).
//
// Scope of the fast path: bf16, NCHW, H = W = 8, kernel 3x3 (pad 1) or 1x1 (pad 0), stride 1,
// dilation 1, groups 1, Cin <= 128, Cout <= 128.  Everything else stays on the direct kernels.
//
// fprop and dgrad share ONE kernel (dgrad = fprop of dy with the weights transposed and the taps
// mirrored):   Out[p][co] = sum_{r,s,ci} X[ci][p + (r,s) - pad] * Wp[rs][co][ci]
//   M = pixels (256 = four images per workgroup), N = output channels (128, zero padded),
//   K = taps * Cin; v_mfma_f32_16x16x32_bf16, fp32 accumulation; 8 waves, wave tile 64 px x 64 co.
//   * the input images live in LDS channel-LAST ([10x10 padded pixel][ci]) so that a pixel fragment
//     (8 consecutive ci of one shifted pixel) is one ds_read_b128; the NCHW -> channel-last
//     transposition is an 8x8 register transpose per thread between coalesced 16-byte global loads
//     and 16-byte LDS writes.
//   * weights are pre-packed by a tiny kernel to [tap][co][ci] (ci contiguous) and streamed by LDS-DMA
//     (global_load_lds_dwordx4) through a three-slot ring, one (tap, 64-channel chunk) per stage.
//   * both LDS images are XOR-swizzled on 16-byte chunks so every fragment read is conflict-free
//     (see the lane -> pixel permutation in px_of_col); the same permutation makes a lane's four
//     accumulator rows four consecutive pixels, so the epilogue stores 8 bytes per lane.
//   * the two waves of every SIMD run the READ and MFMA phases of a stage in ping-pong (raw s_barrier,
//     counted vmcnt), see the main loop.
// wgrad is a plain NT GEMM per tap, dW[rs][co][ci] = sum_{n,p} dY[n][co][p] * Xshift_rs[n][ci][p]:
//   the K dimension runs over pixels of many images; the tap shift is applied while staging X
//   (row select + a 16-bit funnel shift inside the 16-byte row), partial sums of the image
//   splits go to an fp32 workspace and a small kernel reduces them into dW[co][ci][r][s].
#include <map>
#include <mutex>
#include <tuple>
#include "device_utils.h"
#include "conv_geom.h"
#include "wgrad_reduce.h"
#include <thread>
#include "conv_narrow_pack.h"

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
// one launch writes BOTH layouts: [fprop image (KPf) | dgrad image (KPd) | fprop K-tail | dgrad K-tail]
// K-TAILS (round 6, 3x3 filters whose K side has 1 .. 8 channels beyond its last whole 32-channel chunk: the ResNet's 100-channel layers).  In
// the images above the last chunk of such a filter is nine stages of 32 k with at most 8 real ones.  A tail image holds the same channels as
// THREE stages of FOUR taps: tail[g][row][8 q + c] = the filter's value for tap 4 g + q, channel 32 (K / 32) + c (zero for tap >= 9 or a channel
// beyond the filter) - ig_conv8d_kernel reads the lane group q of such a stage from the pixel tap 4 g + q needs.  30 stages instead of 36.
constexpr int IG_TAIL_STAGES = 3;
constexpr int IG_TAIL_ELEMS = IG_TAIL_STAGES * IG_M * 32;
__host__ __device__ inline bool ig_tail_ok(int K, int KS) { return KS == 3 && K > 32 && (K & 31) >= 1 && (K & 31) <= 8; }
__device__ __forceinline__ void ig_pack_body(const bf16_t* __restrict__ w, bf16_t* __restrict__ wp, int Cout, int Cin, int KS, int KPf, int KPd, int first, int stride) {
  const int RS = KS * KS;
  const int lf = KPf == 128 ? 7 : 6, ld = KPd == 128 ? 7 : 6;            // (pad_k: 64 or 128)
  const int nf = RS * IG_M * KPf, nfd = nf + RS * IG_M * KPd;
  const int ntf = ig_tail_ok(Cin, KS) ? IG_TAIL_ELEMS : 0, total = nfd + ntf + (ig_tail_ok(Cout, KS) ? IG_TAIL_ELEMS : 0);
  for (int e0 = first; e0 < total; e0 += stride) {
    bf16_t v; v.bits = 0;
    if (e0 < nfd) {
      const int dgrad = e0 >= nf;
      const int e = dgrad ? e0 - nf : e0, l2 = dgrad ? ld : lf;
      const int k = e & ((1 << l2) - 1), row = (e >> l2) & (IG_M - 1), rs = e >> (l2 + 7);
      const int r = KS == 3 ? (rs * 11) >> 5 : 0, s = rs - r * KS;          // (rs / 3 for rs < 9)
      if (!dgrad) { if (row < Cout && k < Cin) v = w[((row * Cin + k) * KS + r) * KS + s]; }
      else { if (row < Cin && k < Cout) v = w[((k * Cin + row) * KS + (KS - 1 - r)) * KS + (KS - 1 - s)]; }
    } else {
      const int dgrad = e0 >= nfd + ntf;
      const int e = e0 - nfd - (dgrad ? ntf : 0);
      const int c = e & 7, tap = 4 * (e >> 12) + ((e >> 3) & 3), row = (e >> 5) & (IG_M - 1);
      const int r = (tap * 11) >> 5, s = tap - r * 3;
      if (tap < 9) {
        if (!dgrad) { const int k = (Cin & ~31) + c; if (row < Cout && k < Cin) v = w[((row * Cin + k) * 3 + r) * 3 + s]; }
        else { const int k = (Cout & ~31) + c; if (row < Cin && k < Cout) v = w[((k * Cin + row) * 3 + (2 - r)) * 3 + (2 - s)]; }
      }
    }
    wp[e0] = v;
  }
}
__global__ void ig_pack_weights_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wp, int Cout, int Cin, int KS, int KPf, int KPd) {
  ig_pack_body(w, wp, Cout, Cin, KS, KPf, KPd, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// the same for up to IG_PACK_MAX weight tensors in ONE launch (blockIdx.y = tensor): the optimiser re-packs every weight it has just
// updated (igemm_repack_cached), so a training step pays one pack launch instead of one per convolution layer
constexpr int IG_PACK_MAX = 16;
struct PackMany { const bf16_t* w[IG_PACK_MAX]; bf16_t* wp[IG_PACK_MAX]; int Cout[IG_PACK_MAX], Cin[IG_PACK_MAX], KS[IG_PACK_MAX], KPf[IG_PACK_MAX], KPd[IG_PACK_MAX]; };
__global__ void ig_pack_weights_many_kernel(PackMany a) {
  const int t = blockIdx.y;
  ig_pack_body(a.w[t], a.wp[t], a.Cout[t], a.Cin[t], a.KS[t], a.KPf[t], a.KPd[t], blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// ... and the narrow convolutions' fragment images (conv_narrow_pack.h) in the same launch: blockIdx.y >= nig belongs to them
__global__ void ig_ncv_pack_many_kernel(PackMany a, NcvPackMany b, int nig, int nncv) {
  if ((int)blockIdx.y >= nig) {                              // ONE extra row of the grid: block x = (narrow image x / 4, quarter x % 4) - host: 4 nncv <= gridDim.x
    const int e = (int)blockIdx.x >> 2;
    if (e < nncv) ncv_pack_body(b, e, (int)((blockIdx.x & 3) * blockDim.x + threadIdx.x));
    return;
  }
  const int t = blockIdx.y;
  ig_pack_body(a.w[t], a.wp[t], a.Cout[t], a.Cin[t], a.KS[t], a.KPf[t], a.KPd[t], blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
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
  char* Wl = smem + NW * XIMG;              // 3 x IG_WTILE (ring)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: guards raw s_barriers below
  const int wr = wid / NW, wc = wid % NW;
  const int n0 = blockIdx.x * NW;
  const int cmask = (KP >> 3) - 1;

  // halo pixels of the padded 10x10 images must read as 0 (a 1x1 kernel never leaves the interior)
  if (KS > 1) {
    const int cpp = RB >> 4;                                   // 16-byte chunks per pixel
    for (int o = tid; o < NW * 36 * cpp; o += NT) {
      const int ch = o % cpp, hidx = (o / cpp) % 36, img = o / (cpp * 36);
      int hp, wpx;
      if (hidx < 10) { hp = 0; wpx = hidx; }
      else if (hidx < 20) { hp = 9; wpx = hidx - 10; }
      else { hp = 1 + ((hidx - 20) >> 1); wpx = ((hidx - 20) & 1) * 9; }
      *reinterpret_cast<uint4*>(Xl + img * XIMG + (hp * 10 + wpx) * RB + (ch << 4)) = make_uint4(0, 0, 0, 0);
    }
  }
  // NCHW -> channel-last.  A thread takes 8 channels x one image row: eight coalesced 16-byte loads (8 pixels of one
  // channel each), an 8x8 transposition of the 16-bit elements in registers, eight 16-byte LDS writes (8 channels of one
  // pixel each).  Lanes run along the channel groups, so the 8 lanes of a ds_write_b128 group hit 8 different chunks.
  {
    const int ncgp = KP >> 3;
    for (int e = tid; e < NW * 8 * ncgp; e += NT) {
      const int cg = e % ncgp, h = (e / ncgp) & 7, img = e / (ncgp * 8);
      const int n = n0 + img;
      unsigned int w[8][4];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int c = cg * 8 + k;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < CI && n < N) v = *reinterpret_cast<const uint4*>(x + ((int64_t)n * CI + c) * 64 + h * 8);
        w[k][0] = v.x; w[k][1] = v.y; w[k][2] = v.z; w[k][3] = v.w;
      }
      char* xi = Xl + img * XIMG;
#pragma unroll
      for (int p = 0; p < 8; p++) {
        unsigned int d[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const unsigned int lo = w[2 * j][p >> 1], hi = w[2 * j + 1][p >> 1];
          d[j] = (p & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
        }
        const int hp = h + 1, wpx = p + 1;
        *reinterpret_cast<uint4*>(xi + (hp * 10 + wpx) * RB + ((cg ^ x_swz(hp, wpx, cmask)) << 4)) = make_uint4(d[0], d[1], d[2], d[3]);
      }
    }
  }

  f4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};

  constexpr int KW = 64;                    // k per stage (KP is 64 or 128)
  const int CC = KP / KW;                   // chunks per tap
  const int T = RS * CC;
  // Weight stages arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write) into a ring of THREE
  // slots, two stages ahead of their use.  The DMA destination is lane-linear (wave base + lane * 16), so the XOR swizzle
  // of the tile is applied to the per-lane SOURCE address; ig_kc_off() applies the same involution on the read side.
  typedef __attribute__((address_space(3))) char lds_char_t;
  typedef const __attribute__((address_space(1))) char glb_char_t;
  auto stage_dma = [&](int t, int slot) {
    const int rs1 = t / CC, cc1 = t - rs1 * CC;
    const bf16_t* base = wp + (int64_t)rs1 * IG_M * KP + cc1 * KW;
#pragma unroll
    for (int i = 0; i < LPT; i++) {
      const int piece = wid * LPT + i;                 // 1 KiB piece of the 16 KiB slot written by this wave-instruction
      const int p = piece * 64 + lane;                 // 16-byte position inside the slot
      const int row = p >> 3, chunk = (p & 7) ^ (row & 7);
      __builtin_amdgcn_global_load_lds((glb_char_t*)(base + row * KP + chunk * 8), (lds_char_t*)(Wl + slot * IG_WTILE + piece * 1024), 16, 0, 0);
    }
  };
  stage_dma(0, 0);
  if (T > 1) stage_dma(1, 1);

  // B-fragment addresses.  For tap (r, s) and n-tile j a lane reads pixel (2j + rowsel + r, wpix + s) of image wc, chunk
  // (k-chunk ^ swizzle).  k-chunk = uniform part (cc, ks: bits 2..) ^ lane part (lane >> 4: bits 0..1), and XOR is
  // bitwise, so everything except the uniform part is folded into ONE precomputed byte offset per (tap, j); the hot loop
  // then needs a single v_xor per ds_read instead of ~10 integer instructions.
  int rowsel, wpix;
  px_of_col(lane & 15, rowsel, wpix);
  int pre[RS][4];
#pragma unroll
  for (int rs = 0; rs < RS; rs++) {
    const int r = rs / KS, s = rs - r * KS;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int hp = 2 * j + rowsel + r + (1 - PAD), wpx = wpix + s + (1 - PAD);
      pre[rs][j] = wc * XIMG + (hp * 10 + wpx) * RB + ((((lane >> 4) & cmask) ^ x_swz(hp, wpx, cmask)) << 4);
    }
  }
  // A-fragment (weight tile) addresses: row = wr*64 + i*16 + (lane & 15), chunk = (ks*4 + (lane >> 4)) ^ (row & 7)
  const int a_row = (wr * 64 + (lane & 15)) * 128;
  const int a_ch0 = ((lane >> 4) ^ (lane & 7)) << 4, a_ch1 = ((4 + (lane >> 4)) ^ (lane & 7)) << 4;

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // image tiles and weight stages 0, 1 are in LDS

  // Two-phase ping-pong.  A stage is a READ phase (16 ds_read_b128: both k-steps' fragments, plus the DMA of stage t+2)
  // and an MFMA phase (32 MFMAs), each closed by a raw s_barrier.  The waves with wr == 1 - the second wave of every
  // SIMD - run one phase behind the waves with wr == 0, so while one wave of a SIMD feeds the matrix core the other one
  // reads LDS.  Ordering of the DMA ring (slot = stage % 3):
  //   * DMA(t+2) is issued in READ(t); its slot was last read in READ(t-1), and every read is retired (lgkmcnt(0))
  //     before the barrier that closes its phase;
  //   * each wave retires its own DMA(t+1) with a counted vmcnt before the barrier closing READ(t) (DMA(t+2) stays in
  //     flight), which for both groups is passed before anyone starts READ(t+1).
  bf8v fa0[4], fb0[4], fa1[4], fb1[4];
  if (wr == 1) __builtin_amdgcn_s_barrier();          // measured in one process: 8 % faster than no stagger, 13 % than odd/even
  int t = 0, slot = 0;                               // slot = t % 3
#pragma unroll
  for (int rs = 0; rs < RS; rs++) {
    for (int cc = 0; cc < CC; cc++, t++) {
      const char* wl = Wl + slot * IG_WTILE + a_row;
      const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
      // ---- READ(t)
      if (t + 2 < T) stage_dma(t + 2, slot2);
      const int u = ((cc * (KW >> 3)) & cmask) << 4;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        fa0[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + i * 16 * 128 + a_ch0));
        fa1[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + i * 16 * 128 + a_ch1));
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        fb0[j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(Xl + (pre[rs][j] ^ u)));
        fb1[j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(Xl + (pre[rs][j] ^ u ^ (4 << 4))));
      }
      if (t + 2 < T) {
        if (LPT == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      // ---- MFMA(t)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      slot = slot1;
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();

  // epilogue.  The pixels are the A operand, so D rows = pixels and D cols = output channels: a lane holds, per (i, j),
  // output channel i*16 + (lane & 15) and the four MFMA rows 4q..4q+3 (q = lane >> 4), which px_of_col maps to four
  // CONSECUTIVE pixels of one image row -> one 8-byte store.
  const int n = n0 + wc;
  if (n < N) {
    bf16_t* yp = y + (int64_t)n * CO * 64;
    const int q = lane >> 4;
    const int qrow = (q == 1 || q == 2) ? 1 : 0, qw = (q >= 2) ? 4 : 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int co = wr * 64 + i * 16 + (lane & 15);
      if (co < CO) {
        const float b = bias ? (float)bias[co] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const bf16_t o0(acc[i][j][0] + b), o1(acc[i][j][1] + b), o2(acc[i][j][2] + b), o3(acc[i][j][3] + b);
          uint2 pk;
          pk.x = (unsigned)o0.bits | ((unsigned)o1.bits << 16);
          pk.y = (unsigned)o2.bits | ((unsigned)o3.bits << 16);
          *reinterpret_cast<uint2*>(yp + co * 64 + (2 * j + qrow) * 8 + qw) = pk;
        }
      }
    }
  }
}

// Welford triple (count, mean, M2) of the 16 output values a lane holds for one output channel, merged over the four lanes that share
// the channel (xor 16, 32): the statistics of the wave's 64 pixels of that channel, from the bf16-ROUNDED values the batch norm reads.
__device__ __forceinline__ void ig_stats_merge(float& n, float& mean, float& m2, float n2, float mean2, float m22) {
  const float nt = n + n2, d = mean2 - mean, f = n2 / nt;
  mean = mean + d * f;
  m2 = m2 + m22 + d * d * n * f;
  n = nt;
}
__device__ __forceinline__ void ig_stats_wave(const float (&v)[16], float& n, float& mean, float& m2) {
  const float sh = v[0];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < 16; k++) { const float d = v[k] - sh; s1 += d; s2 += d * d; }
  n = 16.f; mean = sh + s1 * (1.f / 16.f); m2 = s2 - s1 * s1 * (1.f / 16.f);
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) ig_stats_merge(n, mean, m2, __shfl_xor(n, off, 64), __shfl_xor(mean, off, 64), __shfl_xor(m2, off, 64));
}

// The SIBLING of a 3x3 convolution (SIB instantiations): a 1x1 convolution of the SAME input with as many output channels - the shortcut of
// lamp's residual block, whose two branches both start with a Conv2D on the block's input (cnn.scala:16-20, 38-78).  Its product is a second,
// 1/9-length main loop over the images that are already in LDS (centre tap only), run BEFORE the 3x3 product because the 3x3 epilogue
// overwrites the images; it saves the 1x1 launch with its own image burst (33.5 MB for res4), prologue and ramp.
// SIBM = 2 (round 5, dgrad): a SECOND SOURCE instead - the output gradient x2 of that 1x1 sibling, whose input gradient is the other contribution
// to the block's input gradient (autograd.scala:66-84 accumulates the two).  Its product (centre tap, the sibling's transposed filter) runs first
// on its own images, then the 3x3's images replace them in LDS and the 3x3 product continues in the SAME accumulators: one rounding, one
// epilogue, and neither the 1x1's 33.5 MB result nor its re-read as the addend.
// wtail (round 6, ig_conv8d_kernel only; not a sibling's): the K-tail image of the launch's OWN filter (ig_pack_body), or null
struct IgSibling { const bf16_t* wp; const bf16_t* bias; bf16_t* y; float* stats; const bf16_t* x2; const bf16_t* wtail; };

// DEFAULT variant, TWO co-resident workgroups per CU (LAMP_IG_VARIANT=a selects the 4-image kernel above): two images and four waves per
// workgroup, the images WITHOUT halo (taps outside the image read a shared zero pixel), two weight slots: 32 + 0.25 + 32 KiB of
// LDS.  The two workgroups of a CU are independent, so the prologue / epilogue of one overlaps the main loop of the other and
// their READ / MFMA phases interleave without an explicit stagger.
// NWV = 4: waves = 2 images x 2 halves of 64 output channels (two such workgroups per CU at large batches).  NWV = 8 (round 5, batches of at
// most 2 x CUs images, where a CU holds ONE workgroup and the kernel is the latency chain of one wave: 18 stages x 32 dependent MFMAs behind
// 16 fragment reads each - 17 us per 128-channel 3x3 launch at B = 32 ... 256 whatever the batch): 2 images x 4 quarters of 32 channels,
// half the MFMAs and 12 instead of 16 fragment reads per wave and stage.
// SIBM (round 5, the small-batch forms of ig_conv8d_kernel's): 1 = a sibling 1x1 convolution of the same input (fprop) runs first on the staged
// images (centre tap, KP / 64 stages of its own weights) and leaves through the same epilogue into its own tensor; 2 = (dgrad) the sibling's output
// gradient x2 is staged first, multiplied with the sibling's transposed filter, then the 3x3's images replace it and the accumulators go on.
// NW = 1 (round 5, batches of at most one image per CU): ONE image per workgroup, eight waves x 16 output channels - at B <= 256 the two-image
// form fills half the CUs or fewer, and a stage of it is bound by its 96 fragment reads (768 LDS cycles against 512 matrix cycles).
template <int KS, int NWV = 4, int SIBM = 0, int NW = 2>
__global__ __launch_bounds__(NWV * 64) void ig_conv8b_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, const bf16_t* __restrict__ bias,
                                                        bf16_t* __restrict__ y, int N, int CI, int KP, int CO, float* __restrict__ stats,
                                                        const bf16_t* addend, const IgSibling sib) {
  static_assert(SIBM == 0 || KS == 3, "the sibling product is the centre tap of a 3x3 staging");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = KS * KS;
  constexpr int PAD = (KS - 1) / 2;
  constexpr int NT = NWV * 64, LPT = 16 / NWV;              // 16 one-KiB pieces per weight stage, LPT per wave
  constexpr int NTI = 8 * NW / NWV;                          // output-channel tiles of 16 per wave: 4 (64 channels), 2 (32) or 1
  static_assert(NW == 2 || (NW == 1 && NWV == 8), "one image per workgroup: eight waves");
  // weight slots: two (the next stage requested while this one is multiplied), or - one image per workgroup, where a stage is too short to
  // cover an L2 round trip - four, requested three stages ahead behind a counted vmcnt (as ig_conv8c_kernel)
  constexpr int NSLOT = NW == 1 ? 4 : 2;
  const int RB = KP * 2;
  const int XIMG = 64 * RB;                 // 8x8 pixels, no halo
  char* Xl = smem;                          // [2][64][KP] + one zero pixel
  const int ZOFF = NW * XIMG;
  char* Wl = smem + NW * XIMG + RB;         // NSLOT x IG_WTILE
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = NW == 2 ? wid >> 1 : wid, wc = NW == 2 ? wid & 1 : 0;
  const int n0 = blockIdx.x * NW;
  const int cmask = (KP >> 3) - 1;

  // the first weight stage is requested before the image loads: both latencies overlap
  constexpr int KW = 64;
  const int CC = KP / KW;
  const int T = RS * CC;
  typedef __attribute__((address_space(3))) char lds_char_t;
  typedef const __attribute__((address_space(1))) char glb_char_t;
  // stage g of the launch: the sibling's KP / 64 centre-tap stages first (SIBM), then the T stages of this convolution; slot = g % NSLOT
  const int CS = SIBM ? CC : 0, TT = CS + T;
  auto stage_dma = [&](int g) {
    const bf16_t* wimg = (SIBM && g < CS) ? sib.wp : wp;
    const int t = (SIBM && g < CS) ? g : g - CS;
    const int rs1 = t / CC, cc1 = t - rs1 * CC;
    const bf16_t* base = wimg + (int64_t)rs1 * IG_M * KP + cc1 * KW;
#pragma unroll
    for (int i = 0; i < LPT; i++) {
      const int piece = wid * LPT + i;
      const int p = piece * 64 + lane;
      const int row = p >> 3, chunk = (p & 7) ^ (row & 7);
      __builtin_amdgcn_global_load_lds((glb_char_t*)(base + row * KP + chunk * 8), (lds_char_t*)(Wl + (g & (NSLOT - 1)) * IG_WTILE + piece * 1024), 16, 0, 0);
    }
  };
  stage_dma(0);
#pragma unroll
  for (int d = 1; d < NSLOT - 1; d++)
    if (d < TT) stage_dma(d);
  // the wait that ends stage g: stage g + 1 has landed, the (at most NSLOT - 2) stages requested behind it may stay in flight
  auto stage_wait = [&](int g) {
    if (NSLOT == 4 && g + 3 < TT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
    else if (NSLOT == 4 && g + 2 < TT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  if (tid * 16 < RB) *reinterpret_cast<uint4*>(Xl + ZOFF + tid * 16) = make_uint4(0, 0, 0, 0);
  auto load_images = [&](const bf16_t* __restrict__ x) {
    const int ncgp = KP >> 3;
    for (int e = tid; e < NW * 8 * ncgp; e += NT) {
      const int cg = e % ncgp, h = (e / ncgp) & 7, img = e / (ncgp * 8);
      const int n = n0 + img;
      unsigned int w[8][4];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int c = cg * 8 + k;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < CI && n < N) v = *reinterpret_cast<const uint4*>(x + ((int64_t)n * CI + c) * 64 + h * 8);
        w[k][0] = v.x; w[k][1] = v.y; w[k][2] = v.z; w[k][3] = v.w;
      }
      char* xi = Xl + img * XIMG;
#pragma unroll
      for (int p = 0; p < 8; p++) {
        unsigned int d[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const unsigned int lo = w[2 * j][p >> 1], hi = w[2 * j + 1][p >> 1];
          d[j] = (p & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
        }
        *reinterpret_cast<uint4*>(xi + (h * 8 + p) * RB + ((cg ^ x_swz(h + 1, p + 1, cmask)) << 4)) = make_uint4(d[0], d[1], d[2], d[3]);
      }
    }
  };
  load_images(SIBM == 2 ? sib.x2 : x);

  f4v acc[NTI][4];
#pragma unroll
  for (int i = 0; i < NTI; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};


  int rowsel, wpix;
  px_of_col(lane & 15, rowsel, wpix);
  int pre[RS][4];
#pragma unroll
  for (int rs = 0; rs < RS; rs++) {
    const int r = rs / KS, s = rs - r * KS;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int h = 2 * j + rowsel + r - PAD, w = wpix + s - PAD;
      const bool inside = h >= 0 && h < 8 && w >= 0 && w < 8;
      pre[rs][j] = inside ? wc * XIMG + (h * 8 + w) * RB + ((((lane >> 4) & cmask) ^ x_swz(h + 1, w + 1, cmask)) << 4)
                          : ZOFF + (((lane >> 4) & cmask) << 4);
    }
  }
  const int a_row = (wr * (NTI * 16) + (lane & 15)) * 128;
  const int a_ch0 = ((lane >> 4) ^ (lane & 7)) << 4, a_ch1 = ((4 + (lane >> 4)) ^ (lane & 7)) << 4;

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  bf8v fa0[NTI], fb0[4], fa1[NTI], fb1[4];
  // epilogue: + bias, round, [+ addend], store, statistics of the rounded values
  auto epilogue = [&](bf16_t* __restrict__ y, const bf16_t* __restrict__ bias, float* __restrict__ stats, const bf16_t* addend) {
    const int n = n0 + wc;
    if (n >= N) return;
    bf16_t* yp = y + (int64_t)n * CO * 64;
    const int q = lane >> 4;
    const int qrow = (q == 1 || q == 2) ? 1 : 0, qw = (q >= 2) ? 4 : 0;
#pragma unroll
    for (int i = 0; i < NTI; i++) {
      const int co = wr * (NTI * 16) + i * 16 + (lane & 15);
      const float b = (bias && co < CO) ? (float)bias[co] : 0.f;
      float vals[16];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bf16_t o0(acc[i][j][0] + b), o1(acc[i][j][1] + b), o2(acc[i][j][2] + b), o3(acc[i][j][3] + b);
        if (co < CO) {
          uint2 pk;
          pk.x = (unsigned)o0.bits | ((unsigned)o1.bits << 16);
          pk.y = (unsigned)o2.bits | ((unsigned)o3.bits << 16);
          if (addend) {                             // dgrad: y = round(round(conv) + addend), as ig_conv8d_kernel (the statistics are fprop's)
            const uint2 av = *reinterpret_cast<const uint2*>(addend + (int64_t)n * CO * 64 + co * 64 + (2 * j + qrow) * 8 + qw);
            pk.x = add_bf16x2(pk.x, av.x); pk.y = add_bf16x2(pk.y, av.y);
          }
          *reinterpret_cast<uint2*>(yp + co * 64 + (2 * j + qrow) * 8 + qw) = pk;
        }
        vals[4 * j + 0] = __uint_as_float((unsigned)o0.bits << 16); vals[4 * j + 1] = __uint_as_float((unsigned)o1.bits << 16);
        vals[4 * j + 2] = __uint_as_float((unsigned)o2.bits << 16); vals[4 * j + 3] = __uint_as_float((unsigned)o3.bits << 16);
      }
      if (stats) {                              // one partial per image: the batch norm that follows merges them (norm.hip)
        float wn, wm, w2;
        ig_stats_wave(vals, wn, wm, w2);
        if (q == 0 && co < CO) { float* sp = stats + ((int64_t)co * N + n) * 3; /* [channel][image][3] */ sp[0] = wn; sp[1] = wm; sp[2] = w2; }
      }
    }
  };
  if constexpr (SIBM != 0) {
    // ---- the sibling's product: KP / 64 stages of the centre tap
    for (int cc = 0; cc < CC; cc++) {
      const char* wl = Wl + (cc & (NSLOT - 1)) * IG_WTILE + a_row;
      if (cc + NSLOT - 1 < TT) stage_dma(cc + NSLOT - 1);   // (the stages behind the sibling's last one are this convolution's first)
      const int u = ((cc * (KW >> 3)) & cmask) << 4;
#pragma unroll
      for (int i = 0; i < NTI; i++) {
        fa0[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + i * 16 * 128 + a_ch0));
        fa1[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + i * 16 * 128 + a_ch1));
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        fb0[j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(Xl + (pre[RS / 2][j] ^ u)));
        fb1[j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(Xl + (pre[RS / 2][j] ^ u ^ (4 << 4))));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < NTI; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NTI; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      stage_wait(cc);
      __builtin_amdgcn_s_barrier();
    }
    if constexpr (SIBM == 1) {
      epilogue(sib.y, sib.bias, sib.stats, nullptr);
#pragma unroll
      for (int i = 0; i < NTI; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};
    } else {
      load_images(x);                                  // (every wave is past its last read of the sibling's images)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  int t = 0;
#pragma unroll
  for (int rs = 0; rs < RS; rs++) {
    for (int cc = 0; cc < CC; cc++, t++) {
      const char* wl = Wl + ((CS + t) & (NSLOT - 1)) * IG_WTILE + a_row;
      if (CS + t + NSLOT - 1 < TT) stage_dma(CS + t + NSLOT - 1);
      const int u = ((cc * (KW >> 3)) & cmask) << 4;
#pragma unroll
      for (int i = 0; i < NTI; i++) {
        fa0[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + i * 16 * 128 + a_ch0));
        fa1[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + i * 16 * 128 + a_ch1));
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        fb0[j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(Xl + (pre[rs][j] ^ u)));
        fb1[j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(Xl + (pre[rs][j] ^ u ^ (4 << 4))));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < NTI; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NTI; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      stage_wait(CS + t);
      __builtin_amdgcn_s_barrier();
    }
  }

  epilogue(y, bias, stats, addend);
}

// Variant for at most 64 output channels (every igemm layer of the CIFAR ResNet): as ig_conv8b_kernel, but
//  * a weight stage holds only the 64 real rows (8 KiB), so no wave multiplies zero rows: waves = 2 images x 2 halves of 32 channels;
//  * FOUR weight slots, stages requested three ahead behind a counted vmcnt (variant b has one stage in flight and pays an L2 round
//    trip per stage);
//  * 16 + 0.1 + 32 KiB of LDS: three workgroups fit a CU.
// SIBM = 2 (dgrad): a sibling's output gradient as a first set of images, as in ig_conv8b_kernel.
template <int KS, int SIBM = 0>
__global__ __launch_bounds__(256) void ig_conv8c_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, const bf16_t* __restrict__ bias,
                                                        bf16_t* __restrict__ y, int N, int CI, int KP, int CO, float* __restrict__ stats,
                                                        const bf16_t* addend, const IgSibling sib) {
  static_assert(SIBM == 0 || (SIBM == 2 && KS == 3), "second-source input gradients of a 3x3 only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = KS * KS;
  constexpr int PAD = (KS - 1) / 2;
  constexpr int NW = 2, NT = 256, LPT = 2;
  constexpr int WT = 64 * 64 * 2, NSLOT = 4;   // one weight stage: 64 rows x 64 k
  const int RB = KP * 2;
  const int XIMG = 64 * RB;                 // 8x8 pixels, no halo
  char* Xl = smem;                          // [2][64][KP] + one zero pixel
  const int ZOFF = NW * XIMG;
  char* Wl = smem + NW * XIMG + RB;         // NSLOT x WT
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int n0 = blockIdx.x * NW;
  const int cmask = (KP >> 3) - 1;

  // the first weight stage is requested before the image loads: both latencies overlap
  constexpr int KW = 64;
  const int CC = KP / KW;
  const int T = RS * CC;
  typedef __attribute__((address_space(3))) char lds_char_t;
  typedef const __attribute__((address_space(1))) char glb_char_t;
  // stage g of the launch: the sibling's KP / 64 centre-tap stages first (SIBM), then the T stages of this convolution; slot = g & 3
  const int CS = SIBM ? CC : 0, TT = CS + T;
  auto stage_dma = [&](int g) {
    const bf16_t* wimg = (SIBM && g < CS) ? sib.wp : wp;
    const int t = (SIBM && g < CS) ? g : g - CS;
    const int rs1 = t / CC, cc1 = t - rs1 * CC;
    const bf16_t* base = wimg + (int64_t)rs1 * IG_M * KP + cc1 * KW;
#pragma unroll
    for (int i = 0; i < LPT; i++) {
      const int piece = wid * LPT + i;
      const int p = piece * 64 + lane;
      const int row = p >> 3, chunk = (p & 7) ^ (row & 7);
      __builtin_amdgcn_global_load_lds((glb_char_t*)(base + row * KP + chunk * 8), (lds_char_t*)(Wl + (g & 3) * WT + piece * 1024), 16, 0, 0);
    }
  };
  stage_dma(0);
  if (1 < TT) stage_dma(1);
  if (2 < TT) stage_dma(2);
  if (tid * 16 < RB) *reinterpret_cast<uint4*>(Xl + ZOFF + tid * 16) = make_uint4(0, 0, 0, 0);
  auto load_images = [&](const bf16_t* __restrict__ x) {
    const int ncgp = KP >> 3;
    for (int e = tid; e < NW * 8 * ncgp; e += NT) {
      const int cg = e % ncgp, h = (e / ncgp) & 7, img = e / (ncgp * 8);
      const int n = n0 + img;
      unsigned int w[8][4];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int c = cg * 8 + k;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < CI && n < N) v = *reinterpret_cast<const uint4*>(x + ((int64_t)n * CI + c) * 64 + h * 8);
        w[k][0] = v.x; w[k][1] = v.y; w[k][2] = v.z; w[k][3] = v.w;
      }
      char* xi = Xl + img * XIMG;
#pragma unroll
      for (int p = 0; p < 8; p++) {
        unsigned int d[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const unsigned int lo = w[2 * j][p >> 1], hi = w[2 * j + 1][p >> 1];
          d[j] = (p & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
        }
        *reinterpret_cast<uint4*>(xi + (h * 8 + p) * RB + ((cg ^ x_swz(h + 1, p + 1, cmask)) << 4)) = make_uint4(d[0], d[1], d[2], d[3]);
      }
    }
  };
  load_images(SIBM == 2 ? sib.x2 : x);

  f4v acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};


  int rowsel, wpix;
  px_of_col(lane & 15, rowsel, wpix);
  int pre[RS][4];
#pragma unroll
  for (int rs = 0; rs < RS; rs++) {
    const int r = rs / KS, s = rs - r * KS;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int h = 2 * j + rowsel + r - PAD, w = wpix + s - PAD;
      const bool inside = h >= 0 && h < 8 && w >= 0 && w < 8;
      pre[rs][j] = inside ? wc * XIMG + (h * 8 + w) * RB + ((((lane >> 4) & cmask) ^ x_swz(h + 1, w + 1, cmask)) << 4)
                          : ZOFF + (((lane >> 4) & cmask) << 4);
    }
  }
  const int a_row = (wr * 32 + (lane & 15)) * 128;
  const int a_ch0 = ((lane >> 4) ^ (lane & 7)) << 4, a_ch1 = ((4 + (lane >> 4)) ^ (lane & 7)) << 4;

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  bf8v fa0[2], fb0[4], fa1[2], fb1[4];
  if constexpr (SIBM == 2) {
    // ---- the sibling's product (centre tap), then this convolution's images take the place of the sibling's
    for (int g = 0; g < CS; g++) {
      const char* wl = Wl + (g & 3) * WT + a_row;
      if (g + 3 < TT) stage_dma(g + 3);
      const int u = ((g * (KW >> 3)) & cmask) << 4;
#pragma unroll
      for (int i = 0; i < 2; i++) {
        fa0[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + i * 16 * 128 + a_ch0));
        fa1[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + i * 16 * 128 + a_ch1));
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        fb0[j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(Xl + (pre[RS / 2][j] ^ u)));
        fb1[j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(Xl + (pre[RS / 2][j] ^ u ^ (4 << 4))));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      if (g + 3 < TT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (g + 2 < TT) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    load_images(x);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  int t = 0;
#pragma unroll
  for (int rs = 0; rs < RS; rs++) {
    for (int cc = 0; cc < CC; cc++, t++) {
      const int g = CS + t;
      const char* wl = Wl + (g & 3) * WT + a_row;
      if (g + 3 < TT) stage_dma(g + 3);
      const int u = ((cc * (KW >> 3)) & cmask) << 4;
#pragma unroll
      for (int i = 0; i < 2; i++) {
        fa0[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + i * 16 * 128 + a_ch0));
        fa1[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + i * 16 * 128 + a_ch1));
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        fb0[j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(Xl + (pre[rs][j] ^ u)));
        fb1[j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(Xl + (pre[rs][j] ^ u ^ (4 << 4))));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      // stage g + 1 must have landed; stages g + 2 and g + 3 (LPT loads each) may stay in flight
      if (g + 3 < TT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (g + 2 < TT) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }

  const int n = n0 + wc;
  if (n < N) {
    bf16_t* yp = y + (int64_t)n * CO * 64;
    const int q = lane >> 4;
    const int qrow = (q == 1 || q == 2) ? 1 : 0, qw = (q >= 2) ? 4 : 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int co = wr * 32 + i * 16 + (lane & 15);
      const float b = (bias && co < CO) ? (float)bias[co] : 0.f;
      float vals[16];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bf16_t o0(acc[i][j][0] + b), o1(acc[i][j][1] + b), o2(acc[i][j][2] + b), o3(acc[i][j][3] + b);
        if (co < CO) {
          uint2 pk;
          pk.x = (unsigned)o0.bits | ((unsigned)o1.bits << 16);
          pk.y = (unsigned)o2.bits | ((unsigned)o3.bits << 16);
          if (addend) {                             // dgrad: y = round(round(conv) + addend), as ig_conv8d_kernel (the statistics are fprop's)
            const uint2 av = *reinterpret_cast<const uint2*>(addend + (int64_t)n * CO * 64 + co * 64 + (2 * j + qrow) * 8 + qw);
            pk.x = add_bf16x2(pk.x, av.x); pk.y = add_bf16x2(pk.y, av.y);
          }
          *reinterpret_cast<uint2*>(yp + co * 64 + (2 * j + qrow) * 8 + qw) = pk;
        }
        vals[4 * j + 0] = __uint_as_float((unsigned)o0.bits << 16); vals[4 * j + 1] = __uint_as_float((unsigned)o1.bits << 16);
        vals[4 * j + 2] = __uint_as_float((unsigned)o2.bits << 16); vals[4 * j + 3] = __uint_as_float((unsigned)o3.bits << 16);
      }
      if (stats) {                              // one partial per image: the batch norm that follows merges them (norm.hip)
        float wn, wm, w2;
        ig_stats_wave(vals, wn, wm, w2);
        if (q == 0 && co < CO) { float* sp = stats + ((int64_t)co * N + n) * 3; /* [channel][image][3] */ sp[0] = wn; sp[1] = wm; sp[2] = w2; }
      }
    }
  }
}

// relu(bn(x)) applied to eight bf16 values of one channel while they are staged (round 3: the batch norm + relu BETWEEN two convolutions of a
// residual block is folded into the consumer's staging, DESIGN "mid-block fold"): exactly bn_apply2_kernel's arithmetic - one fma on
// (x - mean) with the ROUNDED mean / invstd the batch norm saves, rounded to bf16, then the comparison with 0 on the rounded value.
__device__ __forceinline__ uint4 ig_bn_relu_x8(uint4 v, float mu, float scale, float bb) {
  unsigned* w = &v.x;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const float x0 = __uint_as_float(w[q] << 16), x1 = __uint_as_float(w[q] & 0xffff0000u);
    bf16_t o0(__builtin_fmaf(x0 - mu, scale, bb)), o1(__builtin_fmaf(x1 - mu, scale, bb));
    if ((float)o0 < 0.f) o0 = bf16_t(0.f);
    if ((float)o1 < 0.f) o1 = bf16_t(0.f);
    w[q] = (unsigned)o0.bits | ((unsigned)o1.bits << 16);
  }
  return v;
}

// Variant for MORE than 64 output channels at large batch (the 128- and 100-channel layers of res3 / res4: 80 % of the network's FLOPs):
// ONE workgroup per CU, eight images and eight waves, wave w = image w x ALL output channels (64 pixels x 128 channels, 128 f32
// accumulators per lane).
//  * One weight stage feeds eight images instead of two: the weights - 295 KB per 3x3 layer, re-streamed from L2 by every workgroup -
//    cost 75 MB of L2 -> LDS traffic per launch instead of 302 MB (ig_conv8b at B = 2048: 105 FLOP per byte ingested against the
//    ~32 B/clk a CU takes from L2; here 4x that).
//  * A wave reads 8 weight fragments + 4 pixel fragments per 32 MFMAs (ig_conv8b: 16): the LDS array is busy 37 % of the matrix time.
//  * Weight stages are 128 rows x 32 k (8 KiB, one LDS-DMA piece per wave) in a three-slot ring requested two stages ahead; 64-byte
//    rows, 16-byte chunk c of row r at c ^ ((4 - (r >> 2)) & 3): the four 16-lane groups of a ds_read_b128 (MI355X_MICROARCH.md, LDS)
//    each see 16 different slots.
//  * K runs input-channel chunk OUTER, tap inner, only over the chunks that hold real channels (a 16-channel input is one chunk, not
//    two 64-deep stages of mostly zeros); the images sit in LDS as [chunk][image][pixel][32 channels] (64-byte pixel rows).
//  * Measured with in-kernel stamps (scripts/conv_stamp_probe.py, 128 -> 128 3x3, B = 2048; cycles per workgroup): prologue 12.2k
//    (an HBM burst of every CU at once: 11 B/clk/CU), main loop 41.7k (36 stages x 1158; 1024 = matrix pipe alone), epilogue 6.5k +
//    stores 4k.  Tried and dropped: fetching chunk kc + 1 of the wave's own image during chunk kc (two 32-channel chunks in LDS,
//    four-slot ring, the loads queued behind the ring's requests so the in-order vmcnt waits leave them three stages to land): the
//    prologue fell to 4.7k but every round stalled the barrier-coupled waves on HBM latency - main loop 54.7k, 71k in total vs 65k.
//  * READ / MFMA phases in ping-pong between the two waves of a SIMD (waves 4-7 one phase behind), as in ig_conv8_kernel.
//  * Epilogue: the accumulators go through the (now free) LDS as [channel][64 pixels] rows and leave as 16-byte stores, eight lanes
//    per 128-byte row: full cache lines instead of 8-byte pieces (ig_conv8b writes 1.5x its algorithmic bytes to HBM).
#ifdef IG8D_STAMP
__device__ unsigned long long ig8d_stamps[8 * 512];
__device__ unsigned long long ig8d_rt_stamps[8 * 512];      // the constant 100 MHz clock beside the shader clock: in-kernel frequency
#define IG_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 512) { ig8d_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
                                                                     ig8d_rt_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define IG_STAMP(k) do { } while (0)
#endif
// swizzle of the 64-byte pixel rows: 16-byte chunk c of pixel (h, w) at c ^ (((h & 1) << 1) | ((w >> 2) & 1)); with bank(16-byte slot) =
// (4 (w & 3) + chunk') mod 16 the 16 pixels of an MFMA tile (two rows x eight columns) read 16 different slots
__device__ __forceinline__ int ig8d_swz(int h, int w) { return ((h & 1) << 1) | ((w >> 2) & 1); }

// mean and M2 of the 64 bf16-ROUNDED values of one (image, channel) held as pk[4] by the four lanes q = lane >> 4 of the channel: packed f32
// math on sums shifted by the lane's first value, then two equal-count merges over the lanes (xor 16, 32) without divisions.
// 8 waves x 8 tiles of this is pure vector-issue time (measured: 10.7k of the kernel's 69k cycles with the per-value Welford form).
__device__ __forceinline__ void ig8d_tile_stats(const uint2 (&pk)[4], float& mean, float& m2) {
  typedef float f2v __attribute__((ext_vector_type(2)));
  const float sh = __uint_as_float(pk[0].x << 16);
  const f2v sh2 = {sh, sh};
  f2v s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const f2v va = {__uint_as_float(pk[j].x << 16), __uint_as_float(pk[j].x & 0xffff0000u)};
    const f2v vb = {__uint_as_float(pk[j].y << 16), __uint_as_float(pk[j].y & 0xffff0000u)};
    const f2v da = va - sh2, db = vb - sh2;
    s1 += da; s2 = __builtin_elementwise_fma(da, da, s2);
    s1 += db; s2 = __builtin_elementwise_fma(db, db, s2);
  }
  const float t1 = s1[0] + s1[1], t2 = s2[0] + s2[1];
  mean = sh + t1 * (1.f / 16.f); m2 = t2 - t1 * t1 * (1.f / 16.f);
  // merge over the four lanes that hold the channel (xor 16, 32): equal counts n, so mean' = mean + d / 2, M2' = M2a + M2b + d^2 n / 2
  // (v_permlane16/32_swap leave {own, partner} in {a, b} in an order that depends on the lane: the merge is written on (a, b);
  // lanes may differ in the last bit, only lane q == 0 publishes)
  {
    float ma = mean, mb = mean, qa = m2, qb = m2;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(qa), "+v"(qb));
    const float d = mb - ma; mean = ma + d * 0.5f; m2 = qa + qb + d * d * 8.f;
  }
  {
    float ma = mean, mb = mean, qa = m2, qb = m2;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(qa), "+v"(qb));
    const float d = mb - ma; mean = ma + d * 0.5f; m2 = qa + qb + d * d * 16.f;
  }
}

template <int KS, int NCT, int SIBM>
__global__ __launch_bounds__(512) void ig_conv8d_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, const bf16_t* __restrict__ bias,
                                                        bf16_t* y, int N, int CI, int KP, int CO, float* __restrict__ stats, int stats_per_wg,
                                                        const bf16_t* addend, const float4* __restrict__ affine, const IgSibling sib) {
  constexpr bool SIB = SIBM == 1, DG2 = SIBM == 2;
  static_assert(SIBM == 0 || KS == 3, "the sibling product is the centre tap of a 3x3 staging");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = KS * KS;
  constexpr int PAD = (KS - 1) / 2;
  constexpr int NI = 8, NT = 512;
  constexpr int WT = 128 * 32 * 2, NSLOT = 3;   // one weight stage: 128 rows x 32 k
  constexpr int RB = 64;                    // bytes per pixel of one 32-channel chunk
  constexpr int XIMG = 64 * RB;             // one chunk of one image: 4 KiB
  constexpr int XBUF = NI * XIMG;           // one chunk of the workgroup's eight images
  char* Wl = smem;                          // NSLOT x WT
  char* Xl = smem + NSLOT * WT;             // [chunk][8 images][64 pixels][32 channels].  Taps outside the image read whatever lies up
                                            // to 9 pixels before / after it (weight ring, neighbour image, spare bytes at the end) and
                                            // are zeroed in registers
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // = image of this wave
  const int grp = wid >> 2;                 // second wave of its SIMD: runs one phase behind
  const int n0 = blockIdx.x * NI;
  const int KC = (CI + 31) >> 5;            // 32-channel chunks that hold real channels
  // K-tail (host: KS == 3, 1 .. 8 channels in the last chunk, not the whole-tap form): the last chunk's nine stages are three of four taps each
  const bool tailm = KS == 3 && !SIB && sib.wtail != nullptr;     // (not beside a forward sibling: its epilogue state and the tail's six registers spilled)
  const int KCF = tailm ? KC - 1 : KC;      // chunks multiplied tap by tap
  const int T = RS * KCF + (tailm ? IG_TAIL_STAGES : 0);

  typedef __attribute__((address_space(3))) char lds_char_t;
  typedef const __attribute__((address_space(1))) char glb_char_t;
  // stage (kc, rs): rows = output channels, k = input channels [32 kc, 32 kc + 32) of tap rs; one 1 KiB piece (16 rows) per wave
  const int d_row = wid * 16 + (lane >> 2);
  // the lane's BYTE offset inside a stage, unsigned: uniform base + zero-extended 32-bit offset is the saddr + voffset form of the load (one
  // VGPR); as a signed element index the compiler kept one 64-bit address pair per tap in registers (18 VGPRs beside 128 accumulators)
  const unsigned d_src = (unsigned)(d_row * KP + (((lane & 3) ^ ((4 - ((d_row >> 2) & 3)) & 3)) << 3)) * 2u;
  // SIB: waves 0 - 3 request BOTH 1 KiB pieces (wid, wid + 4) of every stage and waves 4 - 7 none - the sibling's output is stored by waves
  // 4 - 7 only, whose vmcnt then never gates a weight stage, so the sibling's stores may drain under the 3x3 main loop.  Measured: no faster
  // than every wave storing its own image behind a vmcnt(0) (1.0903 vs 1.0913 ms per step, EXPERIMENTS (21)) - the pair is bound by the bytes
  // it writes, wherever the wait stands; kept because it is the form that was tested
  const unsigned d_src4 = d_src + 64u * (unsigned)KP * 2u;      // the same lane of piece wid + 4: 64 rows further
  constexpr int NPIECE = SIB ? 2 : 1;                           // this wave's pieces per stage (when it requests any)
  const bool dma_wave = SIB ? wid < 4 : wid < NCT;
  auto stage_dma_of = [&](const bf16_t* wimg, int kc1, int rs1, int slot) {
    const char* base = reinterpret_cast<const char*>(wimg + (int64_t)rs1 * IG_M * KP + kc1 * 32);
    if constexpr (SIB) {
      if (wid < 4) {
        __builtin_amdgcn_global_load_lds((glb_char_t*)(base + d_src), (lds_char_t*)(Wl + slot * WT + wid * 1024), 16, 0, 0);
        // (rows of channel tiles nobody multiplies, NCT < 8, are loaded all the same: the vmcnt arithmetic stays uniform)
        __builtin_amdgcn_global_load_lds((glb_char_t*)(base + d_src4), (lds_char_t*)(Wl + slot * WT + (wid + 4) * 1024), 16, 0, 0);
      }
    } else {
      if (wid < NCT)                          // rows of channel tiles nobody multiplies (NCT < 8) stay unloaded
        __builtin_amdgcn_global_load_lds((glb_char_t*)(base + d_src), (lds_char_t*)(Wl + slot * WT + wid * 1024), 16, 0, 0);
    }
  };
  // (stage t1 >= RS KCF: stage t1 - RS KCF of the K-tail image [3][128 rows][32 k] - the same pieces and swizzle at a row pitch of 32)
  const unsigned d_srct = (unsigned)(d_row * 32 + (((lane & 3) ^ ((4 - ((d_row >> 2) & 3)) & 3)) << 3)) * 2u;
  auto stage_dma = [&](int kc1, int rs1, int slot) {
    if (kc1 < KCF) { stage_dma_of(wp, kc1, rs1, slot); return; }
    if constexpr (!SIB) {
      const char* base = reinterpret_cast<const char*>(sib.wtail + ((kc1 - KCF) * RS + rs1) * (IG_M * 32));
      if (wid < NCT) __builtin_amdgcn_global_load_lds((glb_char_t*)(base + d_srct), (lds_char_t*)(Wl + slot * WT + wid * 1024), 16, 0, 0);
    }
  };
  // WIDE stages (round 5, one channel tile per wave: NCT = 1, not with a fprop sibling): a stage of the narrow form is 16 rows x 32 k - four
  // MFMAs between two barriers, and a 128-channel input costs 36 (+ 4) of them: the 16-channel input gradient of res3 took 26 us for 67 MB of
  // gradients.  Here a stage is a whole TAP, 16 rows x all KC chunks (wave kc requests chunk kc's 1 KiB piece): RS (+ 1) stages.
  constexpr bool WIDE = NCT == 1 && !SIB;
  const unsigned d_srcw = (unsigned)((lane >> 2) * KP + wid * 32 + (((lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3)) << 3)) * 2u;
  auto stage_dma_w = [&](const bf16_t* wimg, int rs1, int slot) {
    if (wid < KC)
      __builtin_amdgcn_global_load_lds((glb_char_t*)(reinterpret_cast<const char*>(wimg + (int64_t)rs1 * IG_M * KP) + d_srcw),
                                       (lds_char_t*)(Wl + slot * WT + wid * 1024), 16, 0, 0);
  };
  // ... and with ONE chunk (at most 32 input channels: the 16 -> 16 layers) a stage is FOUR taps (wave w requests tap 4 g + w): three stages
  // instead of nine for a 3x3
  const bool taps4 = WIDE && !DG2 && KC == 1 && RS > 1;
  auto stage_dma_t = [&](int g, int slot) {
    const int tap = 4 * g + wid;
    if (wid < 4 && tap < RS)
      __builtin_amdgcn_global_load_lds((glb_char_t*)(reinterpret_cast<const char*>(wp + (int64_t)tap * IG_M * KP) + (d_srcw - (unsigned)(wid * 32) * 2u)),
                                       (lds_char_t*)(Wl + slot * WT + wid * 1024), 16, 0, 0);
  };
  IG_STAMP(0);
  if constexpr (WIDE) {
    if (taps4) { stage_dma_t(0, 0); stage_dma_t(1, 1); }
    else if (DG2) { stage_dma_w(sib.wp, 0, 0); stage_dma_w(wp, 0, 1); }
    else { stage_dma_w(wp, 0, 0); if (RS > 1) stage_dma_w(wp, 1, 1); }
  } else if (SIB || DG2) {                    // the sibling's stages (one tap: stage = 32-channel chunk) come first
    stage_dma_of(sib.wp, 0, 0, 0);
    if (KC > 1) stage_dma_of(sib.wp, 1, 0, 1);
  } else {
    stage_dma(0, 0, 0);
    if (T > 1) stage_dma(1 / RS, 1 % RS, 1);
  }
  IG_STAMP(1);
  unsigned bpair[SIB ? (NCT + 1) / 2 : 1];            // the sibling's bias, tiles 2k and 2k + 1 in one register (its epilogue runs beside 128 live
  if constexpr (SIB) {                                // accumulators); requested with the images so that no wait for it stands behind the ring
#pragma unroll
    for (int k = 0; k < (NCT + 1) / 2; k++) {
      const int c0 = 2 * k * 16 + (lane & 15), c1 = c0 + 16;
      const unsigned b0 = (sib.bias && c0 < CO) ? sib.bias[c0].bits : 0u, b1 = (sib.bias && 2 * k + 1 < NCT && c1 < CO) ? sib.bias[c1].bits : 0u;
      bpair[k] = b0 | (b1 << 16);
    }
  }
  // NCHW -> [pixel][32 channels].  A thread takes 8 channels x one image row: eight coalesced 16-byte loads, an 8x8 transposition of the
  // 16-bit elements in registers, eight 16-byte LDS writes (8 channels of one pixel each).
  auto load_images = [&](const bf16_t* __restrict__ x) {
    for (int e = tid; e < NI * 8 * 4 * KC; e += NT) {
      const int cg = e & 3, h = (e >> 2) & 7, img = (e >> 5) & 7, ch = e >> 8;
      const int n = n0 + img;
      uint4 rw[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int c = ch * 32 + cg * 8 + k;
        rw[k] = make_uint4(0, 0, 0, 0);
        // fprop (statistics wanted): the input is not read again before the backward pass - streamed, the caches are for the output the
        // batch norm reads next; dgrad: dY is what the weight-gradient kernel reads right after this one - cached
        if (c < CI && n < N) {
          const uint4* src = reinterpret_cast<const uint4*>(x + ((int64_t)n * CI + c) * 64 + h * 8);
          rw[k] = stats ? nt_load16(src) : *src;
        }
      }
      if (affine) {
        // the input is a convolution's raw output: relu(bn(.)) of it is what this convolution multiplies (per channel mean, invstd * weight, bias)
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int c = ch * 32 + cg * 8 + k;
          if (c < CI && n < N) { const float4 a = affine[c]; rw[k] = ig_bn_relu_x8(rw[k], a.x, a.y, a.z); }
        }
      }
      char* xi = Xl + ch * XBUF + img * XIMG;
#pragma unroll
      for (int p = 0; p < 8; p++) {
        unsigned int d[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const unsigned int lo = (&rw[2 * j].x)[p >> 1], hi = (&rw[2 * j + 1].x)[p >> 1];
          d[j] = (p & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
        }
        *reinterpret_cast<uint4*>(xi + (h * 8 + p) * RB + ((cg ^ ig8d_swz(h, p)) << 4)) = make_uint4(d[0], d[1], d[2], d[3]);
      }
    }
  };
  load_images(DG2 ? sib.x2 : x);
  f4v acc[NCT][4];
#pragma unroll
  for (int i = 0; i < NCT; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};

  // Pixel-fragment addresses.  Tap (r, s), pixel tile j: pixel (2j + rowsel + r - PAD, wpix + s - PAD) of this wave's image, 16-byte
  // chunk (lane >> 4) ^ swizzle(pixel).  Everything that depends on (r, s, j) linearly - ((r * 8 + s) + 16 j) * RB - is a compile-time
  // immediate of the ds_read; the swizzle depends only on the PARITY of the row (rowsel + r) and on the column (wpix + s), so
  // 2 x KS byte offsets per lane cover all taps and tiles (36 precomputed addresses spilled at 128 accumulators).  Pixels outside the
  // image are read from wherever the address lands (always inside the allocation) and the fragment is zeroed in registers.
  int rowsel, wpix;
  px_of_col(lane & 15, rowsel, wpix);
  const int p0 = wid * XIMG + ((rowsel - PAD) * 8 + (wpix - PAD)) * RB;
  int va[2][KS];
#pragma unroll
  for (int par = 0; par < 2; par++)
#pragma unroll
    for (int s = 0; s < KS; s++)
      va[par][s] = p0 + (((lane >> 4) ^ ig8d_swz(rowsel + par - PAD, wpix + s - PAD)) << 4);
  const bool col_lo = wpix == 0, col_hi = wpix == 7, row_lo = rowsel == 0, row_hi = rowsel == 1;
  // weight fragment of output-channel tile i: row 16 i + (lane & 15), chunk (lane >> 4) ^ swizzle(row)
  const int a_off = (lane & 15) * 64 + ((((lane >> 4) ^ ((4 - ((lane >> 2) & 3)) & 3)) & 3) << 4);

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // images and weight stages 0, 1 are in LDS
  if constexpr (SIB) {                               // (the compiler's own wait for the bias belongs here, where everything has landed)
#pragma unroll
    for (int k = 0; k < (NCT + 1) / 2; k++) asm volatile("" : "+v"(bpair[k]));
  }
  IG_STAMP(2);

  // Two-phase ping-pong, ring discipline as in ig_conv8_kernel: DMA(t+2) is issued in READ(t) into the slot last read in READ(t-1)
  // (every read is retired before the barrier closing its phase); each wave retires its own piece of DMA(t+1) before the barrier
  // closing READ(t), which both groups pass before anyone starts READ(t+1).
  bf8v fa[NCT], fb[4];
  const bf8v zero8 = __builtin_bit_cast(bf8v, s8v{0, 0, 0, 0, 0, 0, 0, 0});
  if constexpr (WIDE) {
    // one stage = one tap over every chunk: READ (all chunks' fragments) | barrier | MFMA | barrier, waves 4 - 7 one phase behind as below.
    // Stage sequence: [the second source's centre tap (DG2)], then the RS taps; slot = stage index mod 3, requested two stages ahead.
    bf8v fw[4], fx[4][4];
    auto read_stage = [&](int slot, int r, int s) {
      const char* wl = Wl + slot * WT + a_off;
#pragma unroll
      for (int kc = 0; kc < 4; kc++)
        if (kc < KC) {
          fw[kc] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + kc * 1024));
          const char* xb = Xl + kc * XBUF + va[r & 1][s];
#pragma unroll
          for (int j = 0; j < 4; j++) fx[kc][j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(xb + ((r * 8 + s) + 16 * j) * RB));
        }
    };
    auto zero_outside = [&](int r, int s) {
      if (KS != 3) return;
      const bool colout = (s == 0 && col_lo) || (s == 2 && col_hi);
#pragma unroll
      for (int kc = 0; kc < 4; kc++)
        if (kc < KC) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const bool out = colout || (r == 0 && j == 0 && row_lo) || (r == 2 && j == 3 && row_hi);
            if (s != 1 || (r == 0 && j == 0) || (r == 2 && j == 3)) fx[kc][j] = out ? zero8 : fx[kc][j];
          }
        }
    };
    auto multiply = [&] {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kc = 0; kc < 4; kc++)
        if (kc < KC) {
#pragma unroll
          for (int j = 0; j < 4; j++) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[kc][j], fw[kc], acc[0][j], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
    };
    int slot = 0;
    if (taps4) {
      // ---- one chunk: stages of four taps
      constexpr int NG = (RS + 3) / 4;
      if (grp == 1) __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int g = 0; g < NG; g++) {
        const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
        if (g + 2 < NG) stage_dma_t(g + 2, slot2);
        const char* wl = Wl + slot * WT + a_off;
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const int tap = 4 * g + p;
          if (tap < RS) {
            const int r = tap / KS, s_ = tap - r * KS;
            fw[p] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + p * 1024));
            const char* xb = Xl + va[r & 1][s_];
#pragma unroll
            for (int j = 0; j < 4; j++) fx[p][j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(xb + ((r * 8 + s_) + 16 * j) * RB));
          }
        }
        if (g + 2 < NG) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (KS == 3) {
#pragma unroll
          for (int p = 0; p < 4; p++) {
            const int tap = 4 * g + p;
            if (tap < RS) {
              const int r = tap / KS, s_ = tap - r * KS;
              const bool colout = (s_ == 0 && col_lo) || (s_ == 2 && col_hi);
#pragma unroll
              for (int j = 0; j < 4; j++) {
                const bool out = colout || (r == 0 && j == 0 && row_lo) || (r == 2 && j == 3 && row_hi);
                if (s_ != 1 || (r == 0 && j == 0) || (r == 2 && j == 3)) fx[p][j] = out ? zero8 : fx[p][j];
              }
            }
          }
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int p = 0; p < 4; p++)
          if (4 * g + p < RS) {
#pragma unroll
            for (int j = 0; j < 4; j++) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[p][j], fw[p], acc[0][j], 0, 0, 0);
          }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        slot = slot1;
      }
      if (grp == 0) __builtin_amdgcn_s_barrier();            // every READ phase of every wave is over: LDS is free
    } else {
    if constexpr (DG2) {
      if (grp == 1) __builtin_amdgcn_s_barrier();
      if (RS > 1) stage_dma_w(wp, 1, 2);                     // (stage 0 of this convolution sits in slot 1 already)
      read_stage(0, PAD, PAD);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      multiply();
      __builtin_amdgcn_s_barrier();
      if (grp == 0) __builtin_amdgcn_s_barrier();            // nobody reads the second source's images any more
      load_images(x);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      slot = 1;
    }
    if (grp == 1) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int rs = 0; rs < RS; rs++) {
      const int r = rs / KS, s = rs - r * KS;
      const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
      if (rs + 2 < RS) stage_dma_w(wp, rs + 2, slot2);
      read_stage(slot, r, s);
      if (rs + 2 < RS) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      zero_outside(r, s);
      __builtin_amdgcn_s_barrier();
      multiply();
      __builtin_amdgcn_s_barrier();
      slot = slot1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();              // every READ phase of every wave is over: LDS is free
    }
  } else {
  if constexpr (SIB || DG2) {
    // ---- sibling product: KC stages of the centre tap, same ring and phases
    if (grp == 1) __builtin_amdgcn_s_barrier();
    int sslot = 0;
    for (int kc = 0; kc < KC; kc++) {
      const char* wl = Wl + sslot * WT + a_off;
      const int slot1 = sslot == 2 ? 0 : sslot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
      if (kc + 2 < KC) stage_dma_of(sib.wp, kc + 2, 0, slot2);
#pragma unroll
      for (int i = 0; i < NCT; i++) fa[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + i * 1024));
      const char* xb = Xl + kc * XBUF + va[PAD & 1][PAD];
#pragma unroll
      for (int j = 0; j < 4; j++) fb[j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(xb + ((PAD * 8 + PAD) + 16 * j) * RB));
      if (kc + 2 < KC) {
        if constexpr (SIB) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");  // (two pieces per stage and requesting wave)
        else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < NCT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      sslot = slot1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();        // every wave has read its last weight stage: the ring is free
    // The 3x3 product's first stage is requested now (slot 0) and lands during this epilogue, which goes through 2 KiB of slots 1 - 2 per
    // wave, one channel tile at a time (the images must stay): every wave writes its [16 channels][64 pixels] tile as the MFMA leaves it,
    // waves 4 - 7 read two images' tiles back as 16-byte chunks and store whole 128-byte rows - the same bytes and statistics as the
    // kernel's own epilogue produces.
    stage_dma(0, 0, 0);
    if constexpr (DG2) {
      // ... or, second source: nobody reads the sibling's images any more; the 3x3's own images take their place, the accumulators stay
      if (T > 1) stage_dma(1 / RS, 1 % RS, 1);
      load_images(x);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                    // images and weight stages 0, 1 are in LDS
    }
    if constexpr (SIB) {
      const int n = n0 + wid;
      char* Sc = Wl + WT + wid * 2048;
      const int q = lane >> 4;
      const int qrow = (q == 1 || q == 2) ? 1 : 0, qw = (q >= 2) ? 4 : 0;
      typedef float f2v __attribute__((ext_vector_type(2)));
      typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
      // statistics of tile i wait in lane group q = i & 3 of register i >> 2 (NCT separate registers beside 128 accumulators spilled);
      // every group gets group 0's bits - the value the kernel's own epilogue publishes (the merges differ in the last bit between lanes)
      float smean[(NCT + 3) / 4], sm2[(NCT + 3) / 4];
#pragma unroll
      for (int k = 0; k < (NCT + 3) / 4; k++) { smean[k] = 0.f; sm2[k] = 0.f; }
      // waves 4 - 7 store the images of waves 2 (wid - 4) and 2 (wid - 4) + 1
      const int src0 = (wid & 3) * 2;
      const char* Sr = Wl + WT + src0 * 2048;
      bf16_t* yp0 = sib.y + (int64_t)(n0 + src0) * CO * 64;
#pragma unroll
      for (int i = 0; i < NCT; i++) {
        const int cl = lane & 15;
        const float b = __uint_as_float((i & 1) ? (bpair[i >> 1] & 0xffff0000u) : (bpair[i >> 1] << 16));
        uint2 pk[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const f2v lo = {acc[i][j][0] + b, acc[i][j][1] + b}, hi = {acc[i][j][2] + b, acc[i][j][3] + b};
          pk[j].x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf2v));
          pk[j].y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf2v));
          const int sl = (2 * j + qrow) * 2 + (qw >> 2);
          *reinterpret_cast<uint2*>(Sc + cl * 128 + ((sl ^ ((cl & 7) << 1)) << 3)) = pk[j];
          acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};         // the 3x3 product starts from zero
        }
        if (sib.stats) {
          float mean, m2;
          ig8d_tile_stats(pk, mean, m2);
          mean = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane & 15) << 2, __builtin_bit_cast(int, mean)));
          m2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane & 15) << 2, __builtin_bit_cast(int, m2)));
          if (q == (i & 3)) { smean[i >> 2] = mean; sm2[i >> 2] = m2; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // every wave's tile i is in the scratch area
        if (wid >= 4) {
#pragma unroll
          for (int u = 0; u < 2; u++) {
#pragma unroll
            for (int it = 0; it < 2; it++) {
              const int idx = it * 64 + lane;
              const int cr = idx >> 3, c = idx & 7, co = i * 16 + cr;
              const uint4 v = *reinterpret_cast<const uint4*>(Sr + u * 2048 + cr * 128 + ((c ^ (cr & 7)) << 4));
              if (co < CO && n0 + src0 + u < N) *reinterpret_cast<uint4*>(yp0 + (int64_t)u * CO * 64 + co * 64 + c * 8) = v;
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                  // ... and has been read: the next tile may overwrite it
      }
      if (sib.stats) {
        if (stats_per_wg) {
          // the eight images' (mean, M2) of every channel meet in the scratch area (1 KiB of float2 per wave), one triple per workgroup leaves
          // (stored by threads of waves 4 - 7, as everything of the sibling)
#pragma unroll
          for (int k = 0; k < (NCT + 3) / 4; k++)
            if (4 * k + q < NCT) reinterpret_cast<float2*>(Sc)[(4 * k + q) * 16 + (lane & 15)] = make_float2(smean[k], sm2[k]);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          const int ch = tid - 256;
          if (ch >= 0 && ch < NCT * 16 && ch < CO) {
            const float2* sl = reinterpret_cast<const float2*>(Wl + WT) + ch;
            const float shift = sl[0].x;
            float s1 = 0.f, s2 = 0.f, sm = 0.f;
#pragma unroll
            for (int w = 0; w < 8; w++) { const float2 v = sl[w * 256]; const float d = v.x - shift; s1 += d; s2 += d * d; sm += v.y; }
            float* sp = sib.stats + ((int64_t)ch * gridDim.x + blockIdx.x) * 3;
            sp[0] = 512.f; sp[1] = shift + s1 * 0.125f; sp[2] = sm + 64.f * (s2 - s1 * s1 * 0.125f);
          }
        } else {
          // one triple per image: through the scratch area as well, so that waves 4 - 7 store them
#pragma unroll
          for (int k = 0; k < (NCT + 3) / 4; k++)
            if (4 * k + q < NCT) reinterpret_cast<float2*>(Sc)[(4 * k + q) * 16 + (lane & 15)] = make_float2(smean[k], sm2[k]);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (wid >= 4) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
              const int img = n0 + src0 + u;
#pragma unroll
              for (int h = 0; h < 2; h++) {
                const int ch = h * 64 + lane;
                const float2 v = reinterpret_cast<const float2*>(Sr + u * 2048)[ch];
                if (ch < NCT * 16 && ch < CO && img < N) { float* sp = sib.stats + ((int64_t)ch * N + img) * 3; sp[0] = 64.f; sp[1] = v.x; sp[2] = v.y; }
              }
            }
          }
        }
      }
    }
    if constexpr (SIB) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                    // nobody reads the scratch area any more: slots 1 - 2 belong to the ring again
      if (T > 1) stage_dma(1 / RS, 1 % RS, 1);
      if (dma_wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stages 0 and 1; waves 4 - 7 have only their stores in flight and do not wait
      __builtin_amdgcn_s_barrier();
    }
  }
  if (grp == 1) __builtin_amdgcn_s_barrier();
  int t = 0, slot = 0;
  for (int kc = 0; kc < KCF; kc++) {
    const char* xk = Xl + kc * XBUF;
#pragma unroll
    for (int rs = 0; rs < RS; rs++, t++) {
      const int r = rs / KS, s = rs - r * KS;
      const char* wl = Wl + slot * WT + a_off;
      const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
      // ---- READ(t)
      if (t + 2 < T) stage_dma(kc + (rs + 2) / RS, (rs + 2) % RS, slot2);
#pragma unroll
      for (int i = 0; i < NCT; i++) fa[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + i * 1024));
      const char* xb = xk + va[r & 1][s];
#pragma unroll
      for (int j = 0; j < 4; j++) fb[j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(xb + ((r * 8 + s) + 16 * j) * RB));
      if constexpr (SIB) {
        if (!dma_wave) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (its outstanding vector-memory operations are the sibling's stores)
        else if (t + 2 < T) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else {
        if (t + 2 < T) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      if (KS == 3) {                                   // taps that fall outside the 8x8 image contribute zeros
        const bool colout = (s == 0 && col_lo) || (s == 2 && col_hi);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const bool out = colout || (r == 0 && j == 0 && row_lo) || (r == 2 && j == 3 && row_hi);
          if (s != 1 || (r == 0 && j == 0) || (r == 2 && j == 3)) fb[j] = out ? zero8 : fb[j];
        }
      }
      __builtin_amdgcn_s_barrier();
      // ---- MFMA(t)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < NCT; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      slot = slot1;
    }
  }
  if constexpr (KS == 3 && !SIB) {
    if (tailm) {
      // ---- K-tail: three stages of four taps over the first 8 channels of chunk KCF.  Lane group q = lane >> 4 is tap 4 g + q: its pixel
      // fragment is the 16-byte piece 0 of the pixel that tap needs (one address per lane and stage, the tile j an immediate), zeroed where
      // the tap leaves the image or does not exist (taps 9 - 11).  Addresses and masks are made HERE, from an opaque copy of the lane
      // number: six registers that must not live beside the accumulators through the loop above.
      int lane2 = lane;
      asm volatile("" : "+v"(lane2));
      int rsel2, wpx2;
      px_of_col(lane2 & 15, rsel2, wpx2);
      const char* xk = Xl + KCF * XBUF + wid * XIMG;
#pragma unroll
      for (int g = 0; g < IG_TAIL_STAGES; g++, t++) {
        const int tap = 4 * g + (lane2 >> 4), tapc = tap < RS ? tap : RS - 1;
        const int r = (tapc * 11) >> 5, s_ = tapc - 3 * r;
        const int hp = rsel2 + r - PAD, wq = wpx2 + s_ - PAD;
        const char* xb = xk + (hp * 8 + wq) * RB + (ig8d_swz(hp, wq) << 4);
        const bool outc = tap >= RS || wq < 0 || wq > 7;
        const bool out0 = outc || hp < 0, out3 = outc || hp + 6 > 7;
        const char* wl = Wl + slot * WT + a_off;
        const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
        if (t + 2 < T) stage_dma(KCF, g + 2, slot2);
#pragma unroll
        for (int i = 0; i < NCT; i++) fa[i] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(wl + i * 1024));
#pragma unroll
        for (int j = 0; j < 4; j++) fb[j] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(xb + 16 * j * RB));
        if (t + 2 < T) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        fb[0] = out0 ? zero8 : fb[0];
        fb[1] = outc ? zero8 : fb[1];
        fb[2] = outc ? zero8 : fb[2];
        fb[3] = out3 ? zero8 : fb[3];
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NCT; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        slot = slot1;
      }
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();          // every READ phase of every wave is over: LDS is free
  }
  IG_STAMP(3);

  // epilogue: + bias, round to bf16, batch-norm statistics from the rounded values (as ig_conv8b), then [channel][64 pixels] rows
  // in this wave's 16 KiB of LDS (8-byte slot s of row c at s ^ ((c & 7) << 1): the 16 channels of a store spread over the banks and
  // 16-byte chunks stay whole) and out as 16-byte stores
  const int n = n0 + wid;
  char* El = smem + wid * 16384;
  const int q = lane >> 4;
  const int qrow = (q == 1 || q == 2) ? 1 : 0, qw = (q >= 2) ? 4 : 0;
  // the statistics (ig8d_tile_stats) are skipped altogether when nobody asked for them (dgrad)
  typedef float f2v __attribute__((ext_vector_type(2)));
  // every tile's bias value requested up front: one memory round trip instead of one per tile (a load inside the tile loop also made
  // every tile wait for the previous tile's statistics stores: vmcnt counts loads and stores in one queue)
  unsigned short braw[NCT];
#pragma unroll
  for (int i = 0; i < NCT; i++) {
    const int co = i * 16 + (lane & 15);
    braw[i] = (bias && co < CO) ? bias[co].bits : (unsigned short)0;
  }
#pragma unroll
  for (int i = 0; i < NCT; i++) {
    const int co = i * 16 + (lane & 15);
    const float b = __uint_as_float((unsigned)braw[i] << 16);
    uint2 pk[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      // two values per v_cvt_pk_bf16_f32 (round to nearest even, as the scalar cast)
      typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
      const f2v lo = {acc[i][j][0] + b, acc[i][j][1] + b}, hi = {acc[i][j][2] + b, acc[i][j][3] + b};
      pk[j].x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf2v));
      pk[j].y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf2v));
      const int sl = (2 * j + qrow) * 2 + (qw >> 2);
      *reinterpret_cast<uint2*>(El + co * 128 + ((sl ^ ((co & 7) << 1)) << 3)) = pk[j];
    }
    if (stats) {                              // one partial per image: the batch norm that follows merges them (norm.hip)
      float mean, m2;
      ig8d_tile_stats(pk, mean, m2);
      if (stats_per_wg) {                       // one triple per WORKGROUP (host: N % 8 == 0): the eight images' triples meet in LDS below
        if (q == 0) { float2* sl = reinterpret_cast<float2*>(smem + 8 * 16384) + wid * 128 + co; *sl = make_float2(mean, m2); }
      } else if (q == 0 && co < CO && n < N) { float* sp = stats + ((int64_t)co * N + n) * 3; sp[0] = 64.f; sp[1] = mean; sp[2] = m2; }
    }
  }
  if (stats && stats_per_wg) {
    // The batch norm that reads this output merges the partial statistics of a channel in every one of its workgroups: with one triple
    // per image that merge (2048 x 12 bytes per channel) was as much load traffic as the workgroup's slice of the activation.  Eight
    // equal-count triples -> one of count 512: shift = mean_0, mean = shift + S1 / 8, M2 = sum M2_s + 64 (S2 - S1^2 / 8), fixed order.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid < NCT * 16 && tid < CO) {
      const float2* sl = reinterpret_cast<const float2*>(smem + 8 * 16384) + tid;
      const float shift = sl[0].x;
      float s1 = 0.f, s2 = 0.f, sm = 0.f;
#pragma unroll
      for (int w = 0; w < 8; w++) { const float2 v = sl[w * 128]; const float d = v.x - shift; s1 += d; s2 += d * d; sm += v.y; }
      float* sp = stats + ((int64_t)tid * gridDim.x + blockIdx.x) * 3;
      sp[0] = 512.f; sp[1] = shift + s1 * 0.125f; sp[2] = sm + 64.f * (s2 - s1 * s1 * 0.125f);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the region is private to this wave
  IG_STAMP(4);
  if (n < N) {
    bf16_t* yp = y + (int64_t)n * CO * 64;
    if (addend) {
      // y = round(round(conv) + addend): the values of the convolution followed by an elementwise add (the gradient of a tensor with
      // two consumers), without the second and third pass over the activation
      const bf16_t* ap = addend + (int64_t)n * CO * 64;
      uint4 av[NCT * 2];
#pragma unroll
      for (int it = 0; it < NCT * 2; it++) {
        const int idx = it * 64 + lane;
        const int co = idx >> 3, c = idx & 7;
        av[it] = make_uint4(0, 0, 0, 0);
        if (co < CO) av[it] = *reinterpret_cast<const uint4*>(ap + co * 64 + c * 8);
      }
#pragma unroll
      for (int it = 0; it < NCT * 2; it++) {
        const int idx = it * 64 + lane;
        const int co = idx >> 3, c = idx & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(El + co * 128 + ((c ^ (co & 7)) << 4));
        if (co < CO) *reinterpret_cast<uint4*>(yp + co * 64 + c * 8) = add_bf16x8(v, av[it]);
      }
    } else {
#pragma unroll
      for (int it = 0; it < NCT * 2; it++) {
        const int idx = it * 64 + lane;
        const int co = idx >> 3, c = idx & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(El + co * 128 + ((c ^ (co & 7)) << 4));
        if (co < CO) *reinterpret_cast<uint4*>(yp + co * 64 + c * 8) = v;
      }
    }
  }
  IG_STAMP(5);
#ifdef IG8D_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  IG_STAMP(6);
#endif
}
#ifdef IG8D_STAMP
extern "C" int lamp_debug_ig8d_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(ig8d_stamps), sizeof(ig8d_stamps)) == hipSuccess ? 0 : 1; }
extern "C" int lamp_debug_ig8d_rt_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(ig8d_rt_stamps), sizeof(ig8d_rt_stamps)) == hipSuccess ? 0 : 1; }
#endif

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
// NARROW: some 16-channel tiles hold only padding and are not written (see tile_active)
// CIT = 32-channel slices of Cin one workgroup owns.  3x3: 1 (144 accumulator registers for the nine taps of one slice).  1x1: up to 4 -
// the whole Cin - so that dY is read ONCE instead of once per slice (128 -> 100: 139 -> 60 MB per launch, 33 -> ~15 us).
// PAIR (round 5, 3x3 with one Cin slice per workgroup): the output gradient dy2 [N][CO2][64] of a sibling 1x1 convolution of the same x is staged
// behind the X copies and its weight gradient (the centre-tap product) accumulates in four more tiles: partial2[split][COP2][CIP].
// CI16 (round 5): a partial-sum tile of 16 input-channel columns (Cin <= 16).  The default wave layout - two halves of the output channels x two
// halves of the 32 columns - leaves the waves of the second column half multiplying padding; here the four waves own 32 output channels each:
// half the MFMAs per wave, all of them real.
template <int KS, bool NARROW, int CIT, bool PAIR = false, bool CI16 = false>
__global__ __launch_bounds__(256) void ig_wgrad8v2_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ partial,
                                                          int N, int CO, int CI, int CIP, int images_per_split, int COP, int ntile,
                                                          const bf16_t* __restrict__ dy2, float* __restrict__ partial2, int CO2, int COP2) {
  static_assert(!PAIR || (KS == 3 && CIT == 1), "the pair rides on the 3x3 kernel with one slice of Cin");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RS = KS * KS;
  constexpr int PAD = (KS - 1) / 2;
  // XCD-aware mapping: the ntile workgroups that read the SAME images' dY (one per Cin slice) get block ids that are equal modulo 8 -
  // workgroups are dealt round-robin over the 8 XCDs - so they share one XCD's L2: dY comes from HBM once and from L2 ntile - 1
  // times (3x3, 128 channels: 167 -> 67 MB of HBM reads per launch; the kernel was bound by those reads, not by the matrix cores)
  const int nsplit = gridDim.x / ntile;
  int tile, split;
  {
    const int b = blockIdx.x;
    if ((nsplit & 7) == 0) { const int xcd = b & 7, slot = b >> 3; tile = slot % ntile; split = xcd + 8 * (slot / ntile); }
    else { tile = b % ntile; split = b / ntile; }
  }
  const int ci0 = tile * WG_CI * CIT;
  constexpr int STAGE1 = IG_WTILE + (CIT > 1 ? CIT * WG_XCOPY : 3 * WG_XCOPY) + 512;  // dY tile + X copies
  constexpr int STAGE = STAGE1 + (PAIR ? IG_WTILE : 0);                               // (+ the sibling's dY tile); host: 2 * this
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  constexpr int NI = CI16 ? 2 : 4;                           // output-channel tiles of 16 per wave
  const int wr = CI16 ? wid : wid >> 1, wc = CI16 ? 0 : wid & 1;
  const int co_w = wr * (NI * 16);                            // this wave's first output channel
  const int nbeg = split * images_per_split, nend = min(nbeg + images_per_split, N);
  // zero both stages once: the padding rows of the shifted copies are never written again
  for (int o = tid * 16; o < 2 * STAGE; o += 256 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
  __syncthreads();

  const int xci = tid >> 3, xh = tid & 7;          // this thread's (channel within a slice, image row) of the X tile
  auto load_x = [&](int n, int ct) -> uint4 {
    const int c = ci0 + ct * WG_CI + xci;
    if (c < CI) return *reinterpret_cast<const uint4*>(x + ((int64_t)n * CI + c) * 64 + xh * 8);
    return make_uint4(0, 0, 0, 0);
  };
  auto store_x = [&](char* stage, uint4 v, int ct) {
    char* xb = stage + IG_WTILE + (ct * WG_CI + xci) * WG_XCH + (xh + 1) * 16;
    if (KS == 1) { *reinterpret_cast<uint4*>(xb) = v; return; }
    // (round 5: only the unshifted copy is stored; the fragments of the filter columns s = 0 / 2, out[w] = in[w -/+ 1], are cut from it in
    // registers after the read - 4 + 3 instead of 4 + 9 fragment reads per k-step, one LDS write per packet instead of three)
    *reinterpret_cast<uint4*>(xb + WG_XCOPY) = v;
  };

  f4v acc[CIT][RS][NI];
#pragma unroll
  for (int ct = 0; ct < CIT; ct++)
#pragma unroll
    for (int t = 0; t < RS; t++)
#pragma unroll
      for (int i = 0; i < NI; i++) acc[ct][t][i] = f4v{0.f, 0.f, 0.f, 0.f};
  // Narrow layers (at most 64 output or 16 input channels): output-channel tiles at or beyond COP = round16(CO) and the second
  // 16-column half when CIP = 16 hold only padding - they are not written, and the partial sums are [COP][CIP] per tap instead of
  // [128][32] (16 -> 16: 2.4 MB of partials per launch instead of 37.7 MB).  They ARE still multiplied: uniform branches inside the
  // MFMA chain made the kernel 25 - 50 % slower (measured), and skipping the tiles did not make the narrow layers faster either
  const bool col_active = ci0 + wc * 16 < CIP;
  bool tile_active[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) tile_active[i] = col_active && (co_w + i * 16 < COP);

  uint4 ra[4], rx[CIT], rb[PAIR ? 4 : 1];
  f4v acc2[PAIR ? NI : 1];
  if constexpr (PAIR) {
#pragma unroll
    for (int i = 0; i < NI; i++) acc2[i] = f4v{0.f, 0.f, 0.f, 0.f};
  }
  if (nbeg < nend) {
    ig_stage_load_rows(ra, dy + (int64_t)nbeg * CO * 64, 64, 0, CO, 64, tid);
    if constexpr (PAIR) ig_stage_load_rows(rb, dy2 + (int64_t)nbeg * CO2 * 64, 64, 0, CO2, 64, tid);
#pragma unroll
    for (int ct = 0; ct < CIT; ct++) rx[ct] = load_x(nbeg, ct);
    ig_stage_store_rows(ra, smem, tid);
    if constexpr (PAIR) ig_stage_store_rows(rb, smem + STAGE1, tid);
#pragma unroll
    for (int ct = 0; ct < CIT; ct++) store_x(smem, rx[ct], ct);
  }
  __syncthreads();
  for (int n = nbeg; n < nend; n++) {
    const int cur = (n - nbeg) & 1;
    if (n + 1 < nend) {
      ig_stage_load_rows(ra, dy + (int64_t)(n + 1) * CO * 64, 64, 0, CO, 64, tid);
      if constexpr (PAIR) ig_stage_load_rows(rb, dy2 + (int64_t)(n + 1) * CO2 * 64, 64, 0, CO2, 64, tid);
#pragma unroll
      for (int ct = 0; ct < CIT; ct++) rx[ct] = load_x(n + 1, ct);
    }
    const char* st = smem + cur * STAGE;
    const char* xl = st + IG_WTILE + (wc * 16 + (lane & 15)) * WG_XCH;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf8v fa[NI];
#pragma unroll
      for (int i = 0; i < NI; i++) fa[i] = ig_frag_rows(st, co_w + i * 16, ks, lane);
      const int h = 4 * ks + (lane >> 4);
#pragma unroll
      for (int ct = 0; ct < CIT; ct++) {
        if constexpr (KS == 3) {
          typedef unsigned int u4v_ __attribute__((ext_vector_type(4)));
#pragma unroll
          for (int r = 0; r < KS; r++) {
            const u4v_ xc = *reinterpret_cast<const u4v_*>(xl + ct * WG_CI * WG_XCH + WG_XCOPY + (h + r + (1 - PAD)) * 16);
#pragma unroll
            for (int s_ = 0; s_ < KS; s_++) {
              const u4v_ sh = s_ == 0 ? u4v_{xc[0] << 16, (xc[1] << 16) | (xc[0] >> 16), (xc[2] << 16) | (xc[1] >> 16), (xc[3] << 16) | (xc[2] >> 16)}
                            : s_ == 1 ? xc
                                      : u4v_{(xc[0] >> 16) | (xc[1] << 16), (xc[1] >> 16) | (xc[2] << 16), (xc[2] >> 16) | (xc[3] << 16), xc[3] >> 16};
              const bf8v fb = __builtin_bit_cast(bf8v, sh);
#pragma unroll
              for (int i = 0; i < NI; i++) acc[ct][r * KS + s_][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[ct][r * KS + s_][i], 0, 0, 0);
              if constexpr (PAIR) {
                if (r == PAD && s_ == PAD) {                // the centre tap: the sibling's product on the same X fragment
#pragma unroll
                  for (int i = 0; i < NI; i++)
                    acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ig_frag_rows(st + STAGE1, co_w + i * 16, ks, lane), fb, acc2[i], 0, 0, 0);
                }
              }
            }
          }
        } else {
#pragma unroll
          for (int t = 0; t < RS; t++) {
            const int r = t / KS, s = t % KS;
            s8v v = *reinterpret_cast<const s8v*>(xl + ct * WG_CI * WG_XCH + s * WG_XCOPY + (h + r + (1 - PAD)) * 16);
            const bf8v fb = __builtin_bit_cast(bf8v, v);
#pragma unroll
            for (int i = 0; i < NI; i++) acc[ct][t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[ct][t][i], 0, 0, 0);
          }
        }
      }
    }
    if (n + 1 < nend) {
      char* nx = smem + (cur ^ 1) * STAGE;
      ig_stage_store_rows(ra, nx, tid);
      if constexpr (PAIR) ig_stage_store_rows(rb, nx + STAGE1, tid);
#pragma unroll
      for (int ct = 0; ct < CIT; ct++) store_x(nx, rx[ct], ct);
    }
    __syncthreads();
  }
  if constexpr (PAIR) {
    float* out2 = partial2 + (int64_t)split * COP2 * CIP;
#pragma unroll
    for (int i = 0; i < NI; i++)
      if (col_active && co_w + i * 16 < COP2) {
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const int co = co_w + i * 16 + (lane >> 4) * 4 + rr, ci = ci0 + wc * 16 + (lane & 15);
          if (co < CO2 && ci < CI) out2[co * CIP + ci] = acc2[i][rr];
        }
      }
  }
  // partial[(split * RS + t)][COP][CIP]
#pragma unroll
  for (int t = 0; t < RS; t++) {
    float* out = partial + (int64_t)(split * RS + t) * COP * CIP;
#pragma unroll
    for (int ct = 0; ct < CIT; ct++)
#pragma unroll
      for (int i = 0; i < NI; i++)
        if (!NARROW || tile_active[i]) {
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int co = co_w + i * 16 + (lane >> 4) * 4 + rr, ci = ci0 + ct * WG_CI + wc * 16 + (lane & 15);
            if (co < CO && ci < CI) out[co * CIP + ci] = acc[ct][t][i][rr];   // the reduction skips the padding too
          }
        }
  }
}
// ---- wgrad, 3x3, wide layers: eight waves per workgroup -------------------------------------------------------------------------
// ig_wgrad8v2_kernel<3> runs ONE wave per SIMD (144 accumulator + ~100 other registers per lane, and the compiler fills the rest of
// the file): rocprofv3 shows the matrix pipe 40 % busy, the waves 34 % of their time in s_waitcnt and the rest at the per-image
// barrier, with nobody to fill the gaps; two workgroups per CU cannot co-reside (registers).  Here the same 128 co x 32 ci x 9 taps
// tile is owned by EIGHT waves (32 co x 16 ci each: 72 accumulator registers, launch bound 512 => two waves per SIMD) that overlap
// each other's stalls.  The X fragments are read by four instead of two waves (LDS reads per image 104 -> 176 KB, still under the
// MFMA time); image ranges, partial sums and their reduction are exactly those of the v2 kernel.
// (A 128 co x 64 ci tile per workgroup - dY read half as often - was 10 us faster on the class but doubles the partial sums for
// a fixed number of workgroups: +15 us on the step through the reduction; removed.)
// SHIFT_DY (round 3): the three filter ROWS are taken on the dY side.  A filter-row shift is a shift of the pixel index k by one image row =
// one 16-byte chunk = one lane group of the fragment, and a product summed over k does not care which operand carries the shift:
// sum_k dY[k] X[k + 8 (r - 1)] = sum_k dY[k - 8 (r - 1)] X[k].  So a k-step reads 3 row-shifted dY fragments per channel tile (rows that fall
// outside the image are zeroed in registers) and 3 column-shifted X fragments, 6 + 3 = 9 fragment reads for its 18 MFMAs instead of 2 + 9 = 11:
// the kernel is bound by its LDS fragment reads (176 KB per image and CU against 1152 matrix cycles).
// SHIFT_DY = 2 (round 5): ... and the two column-shifted X fragments are cut from the centre one in registers (the shifts store_x used to write as
// two more LDS copies: out[w] = in[w -/+ 1] with a zero at the border, four shift / or pairs each): 6 + 1 = 7 fragment reads for the 18 MFMAs of a
// k-step instead of 9 - the LDS array falls from the matrix pipe's 1152 cycles per image to 896 - and one LDS write per X packet instead of three.
// PAIR (round 5, with SHIFT_DY = 2): the output gradient dy2 [N][CO2][64] of a sibling 1x1 convolution of the same x is staged beside dY (the LDS the
// two shifted X copies no longer need holds it: 2 x 16 KiB + 5.5 KiB per image), and its weight gradient [CO2][CI] - the centre-tap product
// dy2 . x - accumulates in 8 more registers per lane: 2 more fragment reads and 2 more MFMAs per k-step (18 -> 20) instead of a launch of its own
// that reads x again.  partial2[split][128][CIP].
// SHIFT_DY = 2, later in round 5 (EXPERIMENTS 51 - 57; the cycle counts above assumed 8 LDS cycles per ds_read_b128 - it is 4, and the LDS array
// is ~40 % busy here): a wave's tile is 16 co x 32 ci (3 dY + 2 X fragment reads per k-step; PAIR: + 1), X is the MFMA's A operand (a lane
// holds four consecutive ci of one co: 16-byte partial-sum stores), the fragments of k-step j + 1 are requested before the MFMAs of k-step j
// (rd / mul below), no image request sits in a branch (clamped rows / channels / images, the missing image of an odd range zeroed when it is
// stored), dY rows outside the image are read from a zero slot of the stage, the centre column's MFMAs come first with the shifts of the other
// two columns spread between MFMAs, and the multiplying wave runs at s_setprio 2; the dY tiles arrive by LDS-DMA (dma_tile below: swizzle on the
// source address, inline asm, one vmcnt wait in front of the pair's barrier), x through registers (one packet per thread and pair; the batch-norm
// fold transforms it on the way).  -DLAMP_WG8H_STAMPS: s_memtime per phase (scripts/wg8h_stamps.py); -DLAMP_WG8H_DMA=0: dY through registers.
#ifdef LAMP_WG8H_STAMPS
__device__ unsigned int g_wg8h_stamps[1024 * 8 * 8];
#endif
#ifndef LAMP_WG8H_DMA
#define LAMP_WG8H_DMA 1          // SHIFT_DY = 2: the dY tiles arrive by LDS-DMA (0: through registers and ds_write, the A/B form)
#endif
// A SECOND PROBLEM in the same launch (round 6, igemm_wgrad_group; wgs0 = 0: none): workgroups [0, wgs0) belong to the kernel's own arguments,
// [wgs0, gridDim.x) to this one - another layer's weight gradient of the same batch.  Two layers share the CUs, so each workgroup walks twice as
// many images and each layer leaves HALF the partial sums (they are the kernel's store phase and the reduction's whole input: EXPERIMENTS 73).
struct WgSecondProblem { const bf16_t* dy; const bf16_t* x; float* partial; int CO, CI, CIP, images_per_split, ntile; const float4* affine; int wgs0; };
template <int SHIFT_DY, bool PAIR = false, bool DMA = (LAMP_WG8H_DMA != 0)>
__global__ __launch_bounds__(512) void ig_wgrad8h_kernel(const bf16_t* __restrict__ dy_, const bf16_t* __restrict__ x_, float* __restrict__ partial_,
                                                         int N, int CO_, int CI_, int CIP_, int images_per_split_, int ntile_, const float4* __restrict__ affine_,
                                                         int stream_out, const bf16_t* __restrict__ dy2, float* __restrict__ partial2, int CO2,
                                                         const WgSecondProblem sp) {
  static_assert(!PAIR || SHIFT_DY == 2, "the pair uses the LDS of the shifted X copies");
  const bool second_problem = !PAIR && sp.wgs0 > 0 && (int)blockIdx.x >= sp.wgs0;          // (uniform per workgroup)
  const bf16_t* __restrict__ dy = second_problem ? sp.dy : dy_;
  const bf16_t* __restrict__ x = second_problem ? sp.x : x_;
  float* __restrict__ partial = second_problem ? sp.partial : partial_;
  const int CO = second_problem ? sp.CO : CO_, CI = second_problem ? sp.CI : CI_, CIP = second_problem ? sp.CIP : CIP_;
  const int images_per_split = second_problem ? sp.images_per_split : images_per_split_, ntile = second_problem ? sp.ntile : ntile_;
  const float4* __restrict__ affine = second_problem ? sp.affine : affine_;
  const int wg_index = second_problem ? (int)blockIdx.x - sp.wgs0 : (int)blockIdx.x;
  const int wg_count = second_problem ? (int)gridDim.x - sp.wgs0 : ((!PAIR && sp.wgs0 > 0) ? sp.wgs0 : (int)gridDim.x);
  // stage layout: [dY tile | X copies ...] or, PAIR, [dY tile | dY2 tile | X]; XC = offset of the unshifted X copy
  constexpr int XC = PAIR ? 2 * IG_WTILE : IG_WTILE + WG_XCOPY;
  constexpr int STG = PAIR ? 2 * IG_WTILE + WG_XCOPY + 512 : WG_STAGE;
  constexpr int ZSLOT = STG - 512;                          // 16 zero bytes per image stage (in the padding; SHIFT_DY = 2 reads them for dY rows outside the image)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = 3, RS = 9, PAD = 1;
  const int nsplit = wg_count / ntile;
  int tile, split;
  {                                                           // XCD-aware: the tiles of one image range share an L2 (see ig_wgrad8v2_kernel)
    const int b = wg_index;                                   // (host: wgs0 a multiple of 8 - the XCD of a workgroup is blockIdx.x % 8 for both problems)
    if ((nsplit & 7) == 0) { const int xcd = b & 7, slot = b >> 3; tile = slot % ntile; split = xcd + 8 * (slot / ntile); }
    else { tile = b % ntile; split = b / ntile; }
  }
  const int ci0 = tile * WG_CI;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wid >> 1, wc = wid & 1;                      // 32-channel block of co, 16-channel half of ci
  // SHIFT_DY = 2: the wave's tile is 16 co x 32 ci instead (wave = co block, both ci halves): 3 dY fragments and 2 X fragments serve the
  // 18 MFMAs of a k-step - 5 LDS reads instead of 6 + 1 (the LDS pipe: 640 + 288 cycles per image instead of 896 + 288, next to 1152 MFMA cycles)
  constexpr bool W16 = SHIFT_DY == 2;
  const int nbeg = split * images_per_split, nend = min(nbeg + images_per_split, N);
  // (the shifted X copies kept zero rows / columns that are never written again; with SHIFT_DY = 2 every byte that is read - the dY tiles, rows
  // 1 .. 8 of the unshifted X copy - is written for every image, so nothing has to be cleared)
  if (SHIFT_DY != 2) {
    for (int o = tid * 16; o < 4 * STG; o += 512 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);   // 2 stages x 2 images
    __syncthreads();
  } else if (tid < 4) {
    *reinterpret_cast<uint4*>(smem + tid * STG + ZSLOT) = make_uint4(0, 0, 0, 0);   // (the barrier behind the first pair's stores publishes it)
  }

  const bool xthread = tid < 256;
  const int xci = (tid & 255) >> 3, xh = tid & 7;             // (channel, image row) of the X tile: threads 0..255
  // affine: x is a convolution's raw output and the forward multiplied relu(bn(x)) (ig_conv8d_kernel): the same values are rebuilt here
  const float4 aff = (affine && xthread && ci0 + xci < CI) ? affine[ci0 + xci] : make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_x = [&](int n) -> uint4 {
    if (xthread && ci0 + xci < CI) {
      return *reinterpret_cast<const uint4*>(x + ((int64_t)n * CI + ci0 + xci) * 64 + xh * 8);
    }
    return make_uint4(0, 0, 0, 0);
  };
  // W16: the X tiles of a PAIR of images are 512 packets - one per thread (image tid >> 8), requested without a branch: channels beyond CI read
  // channel CI - 1 (accumulator rows that are never stored), an image beyond nend reads image nend - 1 (its dY is stored as zeros)
  const int xim = tid >> 8, xcc = min(ci0 + xci, CI - 1);
  const float4 aff2 = affine ? affine[xcc] : make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_x2 = [&](int n) -> uint4 {
    return *reinterpret_cast<const uint4*>(x + ((int64_t)min(n + xim, nend - 1) * CI + xcc) * 64 + xh * 8);
  };
  auto store_x2 = [&](char* stage, uint4 v) {
    if (affine) v = ig_bn_relu_x8(v, aff2.x, aff2.y, aff2.z);
    *reinterpret_cast<uint4*>(stage + xim * STG + XC + xci * WG_XCH + (xh + 1) * 16) = v;
  };
  auto store_x = [&](char* stage, uint4 v) {                  // copy s holds out[w] = in[w + s - 1]
    if (!xthread) return;
    // the table is applied HERE, a pair after the load was requested (at the load it would wait for the data and undo the prefetch).
    // Images beyond nend arrive as zeros and leave as relu(bn(0)): their dY is zero, so they add nothing
    if (affine) v = ig_bn_relu_x8(v, aff.x, aff.y, aff.z);
    if (SHIFT_DY == 2) { *reinterpret_cast<uint4*>(stage + XC + xci * WG_XCH + (xh + 1) * 16) = v; return; }   // the shifted fragments are cut from this one after the read
    char* xb = stage + IG_WTILE + xci * WG_XCH + (xh + 1) * 16;
    *reinterpret_cast<uint4*>(xb + WG_XCOPY) = v;
    *reinterpret_cast<uint4*>(xb) = make_uint4(v.x << 16, (v.y << 16) | (v.x >> 16), (v.z << 16) | (v.y >> 16), (v.w << 16) | (v.z >> 16));
    *reinterpret_cast<uint4*>(xb + 2 * WG_XCOPY) = make_uint4((v.x >> 16) | (v.y << 16), (v.y >> 16) | (v.z << 16), (v.z >> 16) | (v.w << 16), v.w >> 16);
  };
  // dY tile [128 co][64 px]: 1024 16-byte packets, two per thread
  auto load_dy = [&](uint4 (&r)[2], int n) {
    const bf16_t* base = dy + (int64_t)n * CO * 64;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int c = tid + i * 512, gr = c >> 3;
      // (W16: rows beyond CO load row CO - 1 again - they only reach accumulator columns that are never stored - so the request needs no branch)
      if constexpr (W16) r[i] = nt_load16(reinterpret_cast<const uint4*>(base + min(gr, CO - 1) * 64 + ((c & 7) << 3)));
      else r[i] = gr < CO ? nt_load16(reinterpret_cast<const uint4*>(base + gr * 64 + ((c & 7) << 3))) : make_uint4(0, 0, 0, 0);   // dY's last reader
    }
  };
  auto store_dy = [&](const uint4 (&r)[2], char* stage) {
#pragma unroll
    for (int i = 0; i < 2; i++) { const int c = tid + i * 512; *reinterpret_cast<uint4*>(stage + ig_kc_off(c >> 3, c & 7)) = r[i]; }
  };
  auto load_dy2 = [&](uint4 (&r)[2], int n) {
    const bf16_t* base = dy2 + (int64_t)n * CO2 * 64;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int c = tid + i * 512, gr = c >> 3;
      r[i] = nt_load16(reinterpret_cast<const uint4*>(base + min(gr, CO2 - 1) * 64 + ((c & 7) << 3)));   // (PAIR implies W16: as above)
    }
  };

  f4v acc[RS][2];
#pragma unroll
  for (int t = 0; t < RS; t++)
#pragma unroll
    for (int i = 0; i < 2; i++) acc[t][i] = f4v{0.f, 0.f, 0.f, 0.f};
  f4v acc2[2] = {f4v{0.f, 0.f, 0.f, 0.f}, f4v{0.f, 0.f, 0.f, 0.f}};      // PAIR: the sibling's [32 co][16 ci] block of this wave

  // Image PAIRS: an LDS stage holds two images, and per pair the loop is
  //     LDS-store the NEXT pair (its registers were requested one pair ago)  ->  request the pair after that  ->  multiply THIS pair  ->  barrier
  // so the LDS stores and the global-load latency run under the MFMAs of two images and there is one barrier per two images.
  // In-kernel s_memtime stamps of the one-image-per-barrier loop (a temporary diagnostic build) showed 2519 cycles per image:
  // 1060 multiplying (the matrix pipe's share), 546 storing, 482 at the barrier, 430 issuing loads - every wave in the same phase
  // at the same time, so the matrix pipe idled 58 % of the loop.  (The barrier is a raw s_barrier behind lgkmcnt(0): __syncthreads()
  // would drain vmcnt and with it the prefetch.)
  // (SHIFT_DY = 0 / 1 only - the A/B forms kept behind LAMP_WGRAD_SHIFT_DY; SHIFT_DY = 2 multiplies through rd() / mul() below)
  auto compute = [&](const char* st) {
    const char* xl = st + IG_WTILE + (wc * 16 + (lane & 15)) * WG_XCH;   // (xl + WG_XCOPY = the unshifted copy)
    if (SHIFT_DY) {
      const bf8v zero8 = __builtin_bit_cast(bf8v, s8v{0, 0, 0, 0, 0, 0, 0, 0});
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const int h = 4 * ks + (lane >> 4);                   // image row of this lane group's eight k
        bf8v fa[KS][2];
#pragma unroll
        for (int r = 0; r < KS; r++) {
          const int hr = h - (r - PAD);                       // the dY row that meets X row h under filter row r
          const bool inside = hr >= 0 && hr < 8;
          const int hc = inside ? hr : h;
#pragma unroll
          for (int i = 0; i < 2; i++) {
            const s8v v = *reinterpret_cast<const s8v*>(st + ig_kc_off(wq * 32 + i * 16 + (lane & 15), hc));
            fa[r][i] = inside ? __builtin_bit_cast(bf8v, v) : zero8;
          }
        }
#pragma unroll
        for (int s_ = 0; s_ < KS; s_++) {
          const s8v v = *reinterpret_cast<const s8v*>(xl + s_ * WG_XCOPY + (h + 1) * 16);   // X row h itself (rows sit one slot down: the zero row above the image)
          const bf8v fb = __builtin_bit_cast(bf8v, v);
#pragma unroll
          for (int r = 0; r < KS; r++)
#pragma unroll
            for (int i = 0; i < 2; i++) acc[r * KS + s_][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[r][i], fb, acc[r * KS + s_][i], 0, 0, 0);
        }
      }
      return;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf8v fa[2];
#pragma unroll
      for (int i = 0; i < 2; i++) fa[i] = ig_frag_rows(st, wq * 32 + i * 16, ks, lane);
      const int h = 4 * ks + (lane >> 4);
#pragma unroll
      for (int t = 0; t < RS; t++) {
        const int r = t / KS, s = t % KS;
        s8v v = *reinterpret_cast<const s8v*>(xl + s * WG_XCOPY + (h + r + (1 - PAD)) * 16);
        const bf8v fb = __builtin_bit_cast(bf8v, v);
#pragma unroll
        for (int i = 0; i < 2; i++) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[t][i], 0, 0, 0);
      }
    }
  };
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); };
  // images at or beyond nend load as zeros (an odd image count: the missing partner contributes nothing)
  uint4 rb[PAIR ? 2 : 1][2];                                 // PAIR: the sibling's dY packets of the pair in flight
  // LDS-DMA of a [128 co][64 px] dY tile (global_load_lds_dwordx4: no staging registers, no ds_write): the destination is lane-linear, so the
  // tile's XOR swizzle sits on the SOURCE address - position p of the image holds chunk (p & 7) ^ (row & 7) of row p >> 3, as ig_kc_off() reads it.
  // Rows beyond `rows` fetch the last real row (accumulator columns that are never stored).
  typedef __attribute__((address_space(3))) char lds_char_t;
  auto dma_tile = [&](const bf16_t* src, int rows, char* tile) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int piece = wid * 2 + i, pos = piece * 64 + lane, row = pos >> 3, chunk = (pos & 7) ^ (row & 7);
      // As inline asm: through the builtin hipcc knows of a pending LDS write and puts vmcnt(0) in front of the next ds_read of ANY stage (it
      // cannot tell the stages apart) - the DMA would be waited for where it is issued.  Its arrival is counted by hand in front of the
      // pair's barrier; hipcc's own count for the x request stays right (that request is the only load it sees, and the youngest).
      const bf16_t* gsrc = src + min(row, rows - 1) * 64 + chunk * 8;
      const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_char_t*)(tile + piece * 1024));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    }
  };
  auto zero_tile = [&](char* tile) {
#pragma unroll
    for (int i = 0; i < 2; i++) *reinterpret_cast<uint4*>(tile + (tid + i * 512) * 16) = make_uint4(0, 0, 0, 0);
  };
  auto dma_pair = [&](char* stage, int n) {                  // the missing partner of an odd image count: a tile of zeros (uniform branch)
#pragma unroll
    for (int im = 0; im < 2; im++) {
      if (n + im < nend) {
        dma_tile(dy + (int64_t)(n + im) * CO * 64, CO, stage + im * STG);
        if constexpr (PAIR) dma_tile(dy2 + (int64_t)(n + im) * CO2 * 64, CO2, stage + im * STG + IG_WTILE);
      } else {
        zero_tile(stage + im * STG);
        if constexpr (PAIR) zero_tile(stage + im * STG + IG_WTILE);
      }
    }
  };
  auto load_pair = [&](uint4 (&ra_)[2][2], uint4 (&rx_)[2], int n) {
    if constexpr (W16) {
#pragma unroll
      for (int im = 0; im < 2; im++) {
        const int nn = min(n + im, nend - 1);
        load_dy(ra_[im], nn);
        if constexpr (PAIR) load_dy2(rb[im], nn);
      }
      rx_[0] = load_x2(n);
      return;
    }
#pragma unroll
    for (int im = 0; im < 2; im++) {
      if (n + im < nend) { load_dy(ra_[im], n + im); rx_[im] = load_x(n + im); if constexpr (PAIR) load_dy2(rb[im], n + im); }
      else {
        ra_[im][0] = ra_[im][1] = make_uint4(0, 0, 0, 0); rx_[im] = make_uint4(0, 0, 0, 0);
        if constexpr (PAIR) rb[im][0] = rb[im][1] = make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_pair = [&](const uint4 (&ra_)[2][2], const uint4 (&rx_)[2], char* stage, int n) {
    if constexpr (W16) {
#pragma unroll
      for (int im = 0; im < 2; im++) {
        const bool live = n + im < nend;                    // (uniform) the missing partner of an odd image count: zeros
        // (component selects: a ?: between two uint4 objects selects ADDRESSES and sends the register arrays to scratch memory)
        auto keep = [live](const uint4& v) { return make_uint4(live ? v.x : 0u, live ? v.y : 0u, live ? v.z : 0u, live ? v.w : 0u); };
        const uint4 t[2] = {keep(ra_[im][0]), keep(ra_[im][1])};
        store_dy(t, stage + im * STG);
        if constexpr (PAIR) { const uint4 t2[2] = {keep(rb[im][0]), keep(rb[im][1])}; store_dy(t2, stage + im * STG + IG_WTILE); }
      }
      store_x2(stage, rx_[0]);
      return;
    }
#pragma unroll
    for (int im = 0; im < 2; im++) {
      store_dy(ra_[im], stage + im * STG); store_x(stage + im * STG, rx_[im]);
      if constexpr (PAIR) store_dy(rb[im], stage + im * STG + IG_WTILE);
    }
  };
  uint4 ra[2][2], rx[2];
  if constexpr (W16 && DMA) {
    if (nbeg < nend) {
      dma_pair(smem, nbeg);
      store_x2(smem, load_x2(nbeg));
      if (nbeg + 2 < nend) rx[0] = load_x2(nbeg + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the DMA's arrival is counted by vmcnt (the x request behind it is waited for too, once)
  } else if (nbeg < nend) {
    load_pair(ra, rx, nbeg);
    store_pair(ra, rx, smem, nbeg);
    if (nbeg + 2 < nend) load_pair(ra, rx, nbeg + 2);
  }
  __syncthreads();
  int cur = 0;
  // (round 5) the second wave of every SIMD (waves 4 - 7) multiplies the pair's first image BEFORE it stores / requests the next pairs: while one
  // wave of a SIMD is busy with LDS stores and load issue the other one has MFMAs to issue (LAMP_WGRAD_STAGGER=0 via `stream_out` bit 1: off)
  const bool late = (wid >> 2) != 0 && !(stream_out & 2) && images_per_split >= 8;     // (with one or two pairs per workgroup the order only delays: B = 256 +2 us)
  const bool prio = (stream_out & 4) != 0;
#ifdef LAMP_WG8H_STAMPS
  unsigned int stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = 0;
  const unsigned int stamp_begin = (unsigned int)__builtin_amdgcn_s_memtime();
#define WG8H_STAMP(k) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned int t_ = (unsigned int)__builtin_amdgcn_s_memtime(); \
    if (k) stamp_sum[k] += t_ - stamp_last; stamp_last = t_; } while (0)
#else
#define WG8H_STAMP(k) do { } while (0)
#endif
  // W16: the pair's four k-steps as a pipeline - the fragments of k-step j + 1 are requested before the MFMAs of k-step j are issued (two sets of
  // 20 / 24 registers), so one LDS latency per pair is exposed instead of four.  In-kernel clocks of the unpipelined loop: a wave alone on its SIMD's
  // matrix pipe needed 1110 ticks for the 36 MFMAs of an image (~440 ticks of issue) - two exposed fragment reads per image.
  struct Frag { bf8v fa[KS]; unsigned int xc[2][4]; bf8v d2; };
  auto rd = [&](Frag& f, const char* st, int ks) {
    const int h = 4 * ks + (lane >> 4);
#pragma unroll
    for (int r = 0; r < KS; r++) {
      const int hr = h - (r - PAD);
      const bool inside = hr >= 0 && hr < 8;
      // a dY row outside the image: the lane reads the stage's zero slot instead (an address chosen once, outside the loop - a select per
      // register here would take the vector issue slots the MFMAs leave: 2 per MFMA for both waves of the SIMD together)
      f.fa[r] = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(st + (inside ? ig_kc_off(wid * 16 + (lane & 15), hr) : ZSLOT)));
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const uint4 v = *reinterpret_cast<const uint4*>(st + XC + (i * 16 + (lane & 15)) * WG_XCH + (h + 1) * 16);
      f.xc[i][0] = v.x; f.xc[i][1] = v.y; f.xc[i][2] = v.z; f.xc[i][3] = v.w;
    }
    if constexpr (PAIR) f.d2 = __builtin_bit_cast(bf8v, *reinterpret_cast<const s8v*>(st + IG_WTILE + ig_kc_off(wid * 16 + (lane & 15), h)));
  };
  auto mul = [&](const Frag& f) {
    typedef unsigned int u4v_ __attribute__((ext_vector_type(4)));
    // the centre column's six MFMAs need no shifted fragment: they are issued first, and the shifts / permutes of the other two columns are
    // spread between MFMAs (two vector instructions fit in the issue slots an MFMA leaves; a clump of eight in front of them drains the pipe)
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const bf8v fc = __builtin_bit_cast(bf8v, u4v_{f.xc[i][0], f.xc[i][1], f.xc[i][2], f.xc[i][3]});
      if constexpr (PAIR) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc, f.d2, acc2[i], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < KS; r++) acc[r * KS + 1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc, f.fa[r], acc[r * KS + 1][i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const u4v_ c = u4v_{f.xc[i][0], f.xc[i][1], f.xc[i][2], f.xc[i][3]};
#pragma unroll
      for (int s_ = 0; s_ < KS; s_ += 2) {
        const u4v_ sh = s_ == 0 ? u4v_{c[0] << 16, (c[1] << 16) | (c[0] >> 16), (c[2] << 16) | (c[1] >> 16), (c[3] << 16) | (c[2] >> 16)}
                                : u4v_{(c[0] >> 16) | (c[1] << 16), (c[1] >> 16) | (c[2] << 16), (c[2] >> 16) | (c[3] << 16), c[3] >> 16};
        const bf8v fb = __builtin_bit_cast(bf8v, sh);
#pragma unroll
        for (int r = 0; r < KS; r++) acc[r * KS + s_][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, f.fa[r], acc[r * KS + s_][i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
    __builtin_amdgcn_sched_group_barrier(0x008, PAIR ? 11 : 9, 0);
  };
#define WG8H_SB() __builtin_amdgcn_sched_barrier(0)
  for (int n = nbeg; n < nend; n += 2, cur ^= 1) {
    char* st = smem + cur * (2 * STG);
    WG8H_STAMP(0);
    if constexpr (W16) {
      Frag f0, f1;
      if (late) {
        if (prio) __builtin_amdgcn_s_setprio(2);
        rd(f0, st, 0); rd(f1, st, 1); WG8H_SB(); mul(f0); WG8H_SB(); rd(f0, st + STG, 0); WG8H_SB(); mul(f1); WG8H_SB();
        if (prio) __builtin_amdgcn_s_setprio(0);
      }
      WG8H_STAMP(1);
      if constexpr (DMA) {
        // x of pair n + 2 from its registers (requested one pair ago), the request of x for pair n + 4, then pair n + 2's dY tiles by DMA - in
        // this order: hipcc puts a vmcnt(0) of its own in front of the x request (it counts nothing in flight there), which behind the DMAs
        // would wait for them.  Everything requested here is waited for once, in front of the pair's barrier, a pair of images later.
        char* nx = smem + (cur ^ 1) * (2 * STG);                // nobody reads that stage since the last barrier
        if (n + 2 < nend) store_x2(nx, rx[0]);
        WG8H_STAMP(2);
        if (n + 4 < nend) rx[0] = load_x2(n + 4);
        if (n + 2 < nend) dma_pair(nx, n + 2);
      } else {
        if (n + 2 < nend) store_pair(ra, rx, smem + (cur ^ 1) * (2 * STG), n + 2);      // nobody reads that stage since the last barrier
        WG8H_STAMP(2);
        if (n + 4 < nend) load_pair(ra, rx, n + 4);
      }
      WG8H_STAMP(3);
      if (prio) __builtin_amdgcn_s_setprio(2);
      if (!late) { rd(f0, st, 0); rd(f1, st, 1); WG8H_SB(); mul(f0); WG8H_SB(); rd(f0, st + STG, 0); WG8H_SB(); mul(f1); WG8H_SB(); }
      WG8H_STAMP(4);
      rd(f1, st + STG, 1); WG8H_SB(); mul(f0); WG8H_SB(); mul(f1);
      if (prio) __builtin_amdgcn_s_setprio(0);
      WG8H_STAMP(5);
      if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      WG8H_STAMP(6);
      continue;
    }
    if (late) { if (prio) __builtin_amdgcn_s_setprio(2); compute(st); if (prio) __builtin_amdgcn_s_setprio(0); }
    WG8H_STAMP(1);
    if (n + 2 < nend) store_pair(ra, rx, smem + (cur ^ 1) * (2 * STG), n + 2);      // nobody reads that stage since the last barrier
    WG8H_STAMP(2);
    if (n + 4 < nend) load_pair(ra, rx, n + 4);
    WG8H_STAMP(3);
    if (prio) __builtin_amdgcn_s_setprio(2);
    if (!late) compute(st);
    WG8H_STAMP(4);
    compute(st + STG);
    if (prio) __builtin_amdgcn_s_setprio(0);
    WG8H_STAMP(5);
    lds_barrier();
    WG8H_STAMP(6);
  }
#ifdef LAMP_WG8H_STAMPS
  if (lane == 0) {
    unsigned int* o = g_wg8h_stamps + (blockIdx.x * 8 + wid) * 8;
    for (int k = 1; k < 7; k++) o[k] = stamp_sum[k];
    o[0] = stamp_last - stamp_begin;                       // the loop
    o[7] = (unsigned int)__builtin_amdgcn_s_memtime() - stamp_begin;
  }
#endif
  if constexpr (W16) {
    // acc[t][i] = [16 ci of half i (rows: 4 consecutive per lane)][16 co (lane & 15)]; columns ci >= CI of a row are padding the reduction never reads
    const int co = wid * 16 + (lane & 15);
    if constexpr (PAIR) {
      float* out2 = partial2 + (int64_t)split * IG_M * CIP;
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int ci = ci0 + i * 16 + (lane >> 4) * 4;
        if (co < CO2 && ci < CI) __builtin_nontemporal_store(acc2[i], reinterpret_cast<f4v*>(&out2[co * CIP + ci]));
      }
    }
#pragma unroll
    for (int t = 0; t < RS; t++) {
      float* out = partial + (int64_t)(split * RS + t) * IG_M * CIP;
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int ci = ci0 + i * 16 + (lane >> 4) * 4;
        if (co < CO && ci < CI) __builtin_nontemporal_store(acc[t][i], reinterpret_cast<f4v*>(&out[co * CIP + ci]));
      }
    }
    return;
  }
  if constexpr (PAIR) {
    // the sibling's block: partial2[split][128][CIP]
    float* out2 = partial2 + (int64_t)split * IG_M * CIP;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int co = wq * 32 + i * 16 + (lane >> 4) * 4 + rr, ci = ci0 + wc * 16 + (lane & 15);
        if (co < CO2 && ci < CI) __builtin_nontemporal_store(acc2[i][rr], &out2[co * CIP + ci]);
      }
  }
  // partial[(split * RS + t)][128][CIP]
#pragma unroll
  for (int t = 0; t < RS; t++) {
    float* out = partial + (int64_t)(split * RS + t) * IG_M * CIP;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int co = wq * 32 + i * 16 + (lane >> 4) * 4 + rr, ci = ci0 + wc * 16 + (lane & 15);
        // padding rows / columns: never read by the reduction (100 of 128: 39 % of the tile).  Streaming stores: 113 MB of partial sums per
        // step that nobody reads before the end of backprop must not push the activations out of L2 / Infinity Cache
        if (co < CO && ci < CI) __builtin_nontemporal_store(acc[t][i][rr], &out[co * CIP + ci]);
      }
  }
}
// (the reduction of the v2 partial sums lives in wgrad_reduce.h: it runs batched with the other layers' reductions)

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
static int pad_k(int64_t c) { return c <= 64 ? 64 : 128; }

// Both packed layouts of a weight tensor (fprop and dgrad) are produced by ONE launch and cached per (weight storage, view,
// stream) while the storage's version is unchanged: every kernel that writes a tensor obtains a mutable pointer through
// Tensor::data()/ptr<T>(), which bumps the version (core/tensor.h), so "same version" proves "same contents".  In a training
// step the weights change once (the optimiser), so each convolution packs once per step instead of once per fprop and once
// per dgrad; with frozen weights (evaluation, gradient accumulation) nothing is repacked at all.  Storages that wrap caller
// memory (lamp_tensor_from_blob) are never cached.  LAMP_PACK_CACHE=0 disables the cache.
namespace {
struct PackKey {
  uint64_t uid; int64_t offset; int KS, Cout, Cin; hipStream_t st;
  bool operator<(const PackKey& o) const { return std::tie(uid, offset, KS, Cout, Cin, st) < std::tie(o.uid, o.offset, o.KS, o.Cout, o.Cin, o.st); }
};
struct PackVal { uint64_t version; Tensor* packed; uint64_t tick; bool pinned = false; };   // pinned: a captured HIP graph reads this image's address
std::mutex g_pack_mu;
std::map<PackKey, PackVal> g_pack_cache;
uint64_t g_pack_tick = 0;
}  // namespace

static int pad_k(int64_t c);
// returns a +1 handle on the buffer [fprop image | dgrad image]; *dgrad_offset = element offset of the second image
// (tail_offset, optional: element offsets of the fprop / dgrad K-tail images, -1 where the filter has none - ig_tail_ok)
static Tensor* packed_weights(const Tensor* w, const ConvGeom& g, int KS, hipStream_t st, int64_t* dgrad_offset, int64_t* tail_offset = nullptr) {
  const int RS = KS * KS;
  const int KPf = pad_k(g.Cin), KPd = pad_k(g.Cout);
  const int64_t nf = (int64_t)RS * IG_M * KPf, nd = (int64_t)RS * IG_M * KPd;
  const int64_t ntf = ig_tail_ok((int)g.Cin, KS) ? IG_TAIL_ELEMS : 0, ntd = ig_tail_ok((int)g.Cout, KS) ? IG_TAIL_ELEMS : 0;
  *dgrad_offset = nf;
  if (tail_offset) { tail_offset[0] = ntf ? nf + nd : -1; tail_offset[1] = ntd ? nf + nd + ntf : -1; }
  static const bool cache_on = [] { const char* e = getenv("LAMP_PACK_CACHE"); return !(e && e[0] == '0'); }();
  const bool cacheable = cache_on && w->st->owned && !w->st->scratch;
  const PackKey key{w->st->uid, w->offset, KS, (int)g.Cout, (int)g.Cin, st};
  const uint64_t ver = w->st->version.load(std::memory_order_relaxed);
  if (cacheable) {
    std::lock_guard<std::mutex> lk(g_pack_mu);
    auto it = g_pack_cache.find(key);
    if (it != g_pack_cache.end() && it->second.version == ver) {
      it->second.tick = ++g_pack_tick;
      if (allocator_capturing()) it->second.pinned = true;   // the graph being captured records this address: never evict the entry
      return retain(it->second.packed);
    }
  }
  int64_t ps[1] = {nf + nd + ntf + ntd};
  Hold wp(new_tensor(ps, 1, kBF16, w->device()));
  hipLaunchKernelGGL(ig_pack_weights_kernel, dim3(grid_for(ps[0], 256)), dim3(256), 0, st, w->ptr<bf16_t>(), wp->ptr<bf16_t>(), (int)g.Cout,
                     (int)g.Cin, KS, KPf, KPd);
  LAMP_LAUNCH_CHECK();
  if (cacheable) {
    std::lock_guard<std::mutex> lk(g_pack_mu);
    auto it = g_pack_cache.find(key);
    if (it != g_pack_cache.end()) { release(it->second.packed); g_pack_cache.erase(it); }
    if (g_pack_cache.size() >= 256) {           // evict the least recently used entry that no captured graph reads
      auto victim = g_pack_cache.end();
      for (auto i = g_pack_cache.begin(); i != g_pack_cache.end(); ++i)
        if (!i->second.pinned && (victim == g_pack_cache.end() || i->second.tick < victim->second.tick)) victim = i;
      if (victim != g_pack_cache.end()) { release(victim->second.packed); g_pack_cache.erase(victim); }
    }
    g_pack_cache[key] = PackVal{ver, retain(wp.get()), ++g_pack_tick, allocator_capturing()};
  }
  return wp.take();
}

// Called by the optimisers right after they have written the parameters: every parameter whose packed images are cached (i.e. that an
// implicit-GEMM convolution used before, on this stream) is packed again now, all of them in one launch, and the cache entries are
// moved to the new storage version - the next step's convolutions find them fresh.  Values are exactly those a lazy pack at first use
// would produce (same kernel body, same weights).  LAMP_PACK_AFTER_STEP=0 restores the lazy packs.
void narrow_repack_cached(lamp_tensor* const* params, int n, hipStream_t st, NcvPackMany* fill, int* fill_cnt);   // conv_narrow.hip
void narrow_pack_launch(const NcvPackMany& a, int cnt, hipStream_t st);
// (round 5: ... and the narrow convolutions' fragment images ride in the last of these launches - the step had two pack launches of 4 - 6 us)
void igemm_repack_cached(lamp_tensor* const* params, int n, hipStream_t st) {
  static const bool on = [] { const char* e = getenv("LAMP_PACK_AFTER_STEP"); return !(e && e[0] == '0'); }();
  if (!on) return;
  NcvPackMany nb;
  int ncnt = 0;
  narrow_repack_cached(params, n, st, &nb, &ncnt);
  struct NarrowLeft { const NcvPackMany& b; int& c; hipStream_t s; ~NarrowLeft() { if (c > 0) narrow_pack_launch(b, c, s); } } narrow_left{nb, ncnt, st};
  PackMany a;
  int cnt = 0, maxtotal = 0;
  std::vector<std::pair<PackKey, uint64_t>> done;       // (entry, storage version its image now corresponds to)
  std::lock_guard<std::mutex> lk(g_pack_mu);
  auto flush = [&] {                                    // one launch per IG_PACK_MAX images (the loop used to STOP there, and a
    if (cnt == 0) return;                               // replayed graph that had captured a cache hit kept reading the stale image)
    hipLaunchKernelGGL(ig_pack_weights_many_kernel, dim3((unsigned)std::min(512, (maxtotal + 255) / 256), (unsigned)cnt), dim3(256), 0, st, a);
    LAMP_LAUNCH_CHECK();
    cnt = 0; maxtotal = 0;
  };
  for (int i = 0; i < n; i++) {
    if (cnt == IG_PACK_MAX) flush();
    const Tensor* w = params[i];
    if (!w || !w->is_device() || w->dtype != kBF16 || w->ndim != 4 || !w->st->owned || !w->is_contiguous()) continue;
    for (auto& kv : g_pack_cache) {
      if (kv.first.uid != w->st->uid || kv.first.offset != w->offset || kv.first.st != st) continue;
      if (kv.first.Cout != (int)w->sizes[0] || kv.first.Cin != (int)w->sizes[1] || kv.first.KS != (int)w->sizes[2]) continue;
      const int KS = kv.first.KS, RS = KS * KS, KPf = pad_k(kv.first.Cin), KPd = pad_k(kv.first.Cout);
      const int total = RS * IG_M * (KPf + KPd) + (ig_tail_ok(kv.first.Cin, KS) ? IG_TAIL_ELEMS : 0) + (ig_tail_ok(kv.first.Cout, KS) ? IG_TAIL_ELEMS : 0);
      if (kv.second.packed->numel() != total) continue;
      a.w[cnt] = w->ptr<bf16_t>(); a.Cout[cnt] = kv.first.Cout; a.Cin[cnt] = kv.first.Cin; a.KS[cnt] = KS; a.KPf[cnt] = KPf; a.KPd[cnt] = KPd;
      // IN PLACE: the entry belongs to this stream, so every convolution that read the old image is ordered before this launch - and a
      // HIP graph captured earlier keeps reading the same address (bench.py replays forward + backprop around the eager optimiser)
      a.wp[cnt] = static_cast<bf16_t*>(kv.second.packed->raw());
      done.push_back({kv.first, w->st->version.load(std::memory_order_relaxed)});
      maxtotal = std::max(maxtotal, total);
      cnt++;
      break;
    }
  }
  static_assert(NCV_NKMAX * 64 <= 4 * 256, "a narrow image is four blocks of the pack kernel");
  if (cnt > 0 && ncnt > 0 && 4 * ncnt <= std::min(512, (maxtotal + 255) / 256)) {      // the last batch takes the narrow images along
    hipLaunchKernelGGL(ig_ncv_pack_many_kernel, dim3((unsigned)std::min(512, (maxtotal + 255) / 256), (unsigned)(cnt + 1)), dim3(256), 0, st, a, nb, cnt, ncnt);
    LAMP_LAUNCH_CHECK();
    cnt = 0; ncnt = 0;
  }
  flush();
  for (auto& d : done) {
    auto it = g_pack_cache.find(d.first);
    if (it != g_pack_cache.end()) { it->second.version = d.second; it->second.tick = ++g_pack_tick; }
  }
}

// addend (dgrad, optional): out = round(round(conv) + addend) where the kernel chosen can do it in its epilogue; *addend_fused says whether it did
// affine (fprop, optional): f32 [Cin][4] = (mean, invstd * weight, bias, -) of the batch norm + relu that stands between the producer of
// `in` and this convolution; applied while staging where the kernel chosen can do it - *affine_used says whether it did (else the caller
// materialises relu(bn(in)) and calls again without it)
// sibling (fprop of a 3x3, optional): a 1x1 convolution of the same input with the same number of output channels, computed by the same launch
// where the eight-image kernel runs - *sibling_fused says whether it did (else the caller runs it as its own convolution)
struct SiblingConv { const Tensor* w; const Tensor* bias; Tensor* out; const ConvGeom* g; };
// second (dgrad of a 3x3, optional): the output gradient and the filter of a sibling 1x1 convolution of the same input; its input gradient is
// accumulated by the same launch (ig_conv8d_kernel<3, ., 2>) - *second_fused says whether it was (igemm_conv_dgrad_pair checks the
// conditions first, so that nothing is launched otherwise)
struct SecondGradConv { const Tensor* dy; const Tensor* w; const ConvGeom* g; };
static void run_conv8(const Tensor* in, const Tensor* w, const Tensor* bias, Tensor* out, const ConvGeom& g, bool dgrad, hipStream_t st,
                      const Tensor* addend = nullptr, bool* addend_fused = nullptr, const Tensor* affine = nullptr, bool* affine_used = nullptr,
                      const SiblingConv* sibling = nullptr, bool* sibling_fused = nullptr, const SecondGradConv* second = nullptr,
                      bool* second_fused = nullptr) {
  if (addend_fused) *addend_fused = false;
  if (affine_used) *affine_used = false;
  if (sibling_fused) *sibling_fused = false;
  if (second_fused) *second_fused = false;
  const int KS = g.kh, RS = KS * KS;
  const int CI = (int)(dgrad ? g.Cout : g.Cin), CO = (int)(dgrad ? g.Cin : g.Cout);
  const int KP = pad_k(CI);
  int64_t dgrad_off = 0;
  int64_t tail_off[2] = {-1, -1};
  Hold wpk(packed_weights(w, g, KS, st, &dgrad_off, tail_off));
  const bf16_t* wpp = static_cast<const Tensor*>(wpk.get())->ptr<bf16_t>() + (dgrad ? dgrad_off : 0);
  // the K-tail image of this direction (1 .. 8 channels beyond the last whole chunk of a 3x3's K side): LAMP_IG_KTAIL=0 multiplies the padded chunk
  const bool ktail_on = [] { const char* e = getenv("LAMP_IG_KTAIL"); return !(e && e[0] == '0'); }();      // (read per call: the A/B test flips it)
  const bf16_t* wtailp = (ktail_on && tail_off[dgrad ? 1 : 0] >= 0) ? static_cast<const Tensor*>(wpk.get())->ptr<bf16_t>() + tail_off[dgrad ? 1 : 0] : (const bf16_t*)nullptr;
  {
    const char* variant = getenv("LAMP_IG_VARIANT");
    if (!(variant && variant[0] == 'a')) {   // default: two co-resident workgroups per CU (A/B on one device: 7 % faster per launch)
      const int blocksb = (int)((g.N + 1) / 2);
      // fprop: per-image batch-norm statistics of the output from the epilogue (LAMP_CONV_BN_STATS=0 turns the hand-off off)
      static const bool bn_stats = [] { const char* e = getenv("LAMP_CONV_BN_STATS"); return !(e && e[0] == '0'); }();
      Hold statt;
      float* statp = nullptr;
      if (bn_stats && !dgrad && g.N >= 2) {
        int64_t ps[1] = {(int64_t)g.N * CO * 3};
        statt = Hold(new_tensor(ps, 1, kF32, in->device()));
        statp = statt->ptr<float>();
      }
      struct Publish { Hold& t; const Tensor* y; int P; ~Publish() { if (t.get()) conv_stats_publish(y, t.get(), P); } } publish{statt, out, (int)g.N};
      // (a sibling, when given, always runs in this launch - igemm_conv_fwd_pair checks the eight-image kernel's conditions first: its work
      // is declared with the launch; its input is the one already counted)
      const double sib_fl = sibling ? conv_flops(*sibling->g) : second ? conv_flops(*second->g) : 0.0;
      const double sib_by = sibling ? conv_bytes(*sibling->g, 2) - (double)g.N * g.Cin * 64 * 2
                                    : second ? conv_bytes(*second->g, 2) - (double)g.N * g.Cin * 64 * 2 : 0.0;   // (one output for both)
      KernelTimer kt("conv_igemm_fprop_dgrad", conv_flops(g) + sib_fl, conv_bytes(g, 2) + sib_by, st);
      const bf16_t* bpb = bias ? bias->ptr<bf16_t>() : (const bf16_t*)nullptr;
      // more than 64 output channels and enough images to give every CU a workgroup of eight: one workgroup per CU, wave = image x all
      // channels (LAMP_IG_VARIANT=d forces it for any batch, =b keeps the two-image kernel)
      const bool force_d = variant && variant[0] == 'd';
      // ... and for narrow outputs too (NCT = 1 / 4 channel tiles per wave): the 16-channel layers of the ResNet (128 -> 16 dgrad,
      // 16 -> 16) spent 16 - 30 us in the 64-row kernel multiplying padding; here they are bound by their image reads (LAMP_IG_SMALL_D=0: off)
      static const bool small_d = [] { const char* e = getenv("LAMP_IG_SMALL_D"); return !(e && e[0] == '0'); }();
      if ((CO > 64 || small_d) && !(variant && variant[0] == 'b') && (force_d || g.N >= 4 * (int64_t)num_cus())) {
        const int blocksd = (int)((g.N + 7) / 8);
        // 3 weight slots + the 32-channel image chunks (+ 9 spare pixels behind them for the taps of a 3x3 kernel) - and at least the
        // 8 x 16 KiB the epilogue stages the output through
        // (+ 8 KiB behind the epilogue's staging area: the eight images' statistics of every channel, merged per workgroup)
        const size_t ldsd = std::max<size_t>((size_t)3 * (128 * 32 * 2) + (size_t)((CI + 31) / 32) * 8 * 4096 + (KS == 3 ? 9 * 64 : 0), (size_t)8 * 16384 + 8 * 128 * 8);
        const int per_wg = (statp && g.N % 8 == 0) ? 1 : 0;
        if (per_wg) publish.P = blocksd;
#define IG_LAUNCH_D(KS_, NCT_, SIB_)                                                                                                        \
  do {                                                                                                                                      \
    allow_big_lds((const void*)ig_conv8d_kernel<KS_, NCT_, SIB_>);                                                                         \
    hipLaunchKernelGGL((ig_conv8d_kernel<KS_, NCT_, SIB_>), dim3(blocksd), dim3(512), ldsd, st, in->ptr<bf16_t>(), wpp, bpb,               \
                       out->ptr<bf16_t>(), (int)g.N, CI, KP, CO, statp, per_wg, addp, affp, sibk);                                         \
  } while (0)
        const bf16_t* addp = addend ? addend->ptr<bf16_t>() : (const bf16_t*)nullptr;
        const float4* affp = affine ? reinterpret_cast<const float4*>(affine->ptr<float>()) : (const float4*)nullptr;
        if (affine_used) *affine_used = affine != nullptr;
        if (addend_fused) *addend_fused = addend != nullptr;
        // the sibling 1x1 of a residual block's first 3x3 (same input, same output channels): second product of this launch
        static const bool sib_on = [] { const char* e = getenv("LAMP_CONV_SIBLING"); return !(e && e[0] == '0'); }();
        IgSibling sibk{nullptr, nullptr, nullptr, nullptr, nullptr};
        Hold sib_wpk, sib_statt;
        const bool sib = sib_on && sibling && !dgrad && KS == 3 && !addend && sibling->g->kh == 1 && sibling->g->Cout == g.Cout &&
                         sibling->g->Cin == g.Cin && sibling->g->N == g.N;
        struct PublishSib { Hold& t; const Tensor* y; int P; ~PublishSib() { if (t.get()) conv_stats_publish(y, t.get(), P); } }
            publish_sib{sib_statt, sib ? sibling->out : nullptr, per_wg ? blocksd : (int)g.N};
        if (sib) {
          int64_t off1 = 0;
          sib_wpk = Hold(packed_weights(sibling->w, *sibling->g, 1, st, &off1));
          sibk.wp = static_cast<const Tensor*>(sib_wpk.get())->ptr<bf16_t>();
          sibk.bias = sibling->bias ? sibling->bias->ptr<bf16_t>() : (const bf16_t*)nullptr;
          sibk.y = sibling->out->ptr<bf16_t>();
          if (statp) {
            int64_t ps[1] = {(int64_t)g.N * CO * 3};
            sib_statt = Hold(new_tensor(ps, 1, kF32, in->device()));
            sibk.stats = sib_statt->ptr<float>();
          }
          if (sibling_fused) *sibling_fused = true;
        }
        // dgrad: the input gradient of the sibling 1x1 accumulated by this launch (its output gradient is a second source of images)
        const bool dg2 = second && dgrad && KS == 3 && second->g->kh == 1 && second->g->Cout == g.Cout && second->g->Cin == g.Cin && second->g->N == g.N;
        if (dg2) {
          int64_t off1 = 0;
          sib_wpk = Hold(packed_weights(second->w, *second->g, 1, st, &off1));
          sibk.wp = static_cast<const Tensor*>(sib_wpk.get())->ptr<bf16_t>() + off1;
          sibk.x2 = second->dy->ptr<bf16_t>();
          if (second_fused) *second_fused = true;
        }
        if (KS == 3 && CO > 16 && !sib) sibk.wtail = wtailp;        // (not in the whole-tap stages of the one-tile form, not beside a forward sibling)
        if (dg2) { if (CO <= 16) IG_LAUNCH_D(3, 1, 2); else if (CO <= 64) IG_LAUNCH_D(3, 4, 2); else if (CO <= 112) IG_LAUNCH_D(3, 7, 2); else IG_LAUNCH_D(3, 8, 2); }
        else if (sib) { if (CO <= 16) IG_LAUNCH_D(3, 1, true); else if (CO <= 64) IG_LAUNCH_D(3, 4, true); else if (CO <= 112) IG_LAUNCH_D(3, 7, true); else IG_LAUNCH_D(3, 8, true); }
        else if (KS == 3) { if (CO <= 16) IG_LAUNCH_D(3, 1, false); else if (CO <= 64) IG_LAUNCH_D(3, 4, false); else if (CO <= 112) IG_LAUNCH_D(3, 7, false); else IG_LAUNCH_D(3, 8, false); }
        else { if (CO <= 16) IG_LAUNCH_D(1, 1, false); else if (CO <= 64) IG_LAUNCH_D(1, 4, false); else if (CO <= 112) IG_LAUNCH_D(1, 7, false); else IG_LAUNCH_D(1, 8, false); }
#undef IG_LAUNCH_D
        LAMP_LAUNCH_CHECK();
        return;
      }
      // the two-image kernels add the second contribution of a residual block's input gradient in their stores too (B <= 512: two add launches less)
      const bf16_t* addbc = (addend && dgrad) ? addend->ptr<bf16_t>() : (const bf16_t*)nullptr;
      if (addend_fused) *addend_fused = addbc != nullptr;
      // ... and take the sibling 1x1 of a block's first 3x3 (fprop, the 128-row kernel) or its output gradient as a second source (dgrad) too
      IgSibling sibb{nullptr, nullptr, nullptr, nullptr, nullptr};
      Hold sibb_wpk, sibb_statt;
      const bool same_shape2 = second && second->g->kh == 1 && second->g->Cout == g.Cout && second->g->Cin == g.Cin && second->g->N == g.N;
      const bool dg2b = second && dgrad && KS == 3 && same_shape2;
      const bool kernel_c = CO <= 64 && !(variant && variant[0] == 'b');
      static const bool sib_on_b = [] { const char* e = getenv("LAMP_CONV_SIBLING"); return !(e && e[0] == '0'); }();
      const bool sibf = sib_on_b && sibling && !dgrad && KS == 3 && !addend && !kernel_c && sibling->g->kh == 1 && sibling->g->Cout == g.Cout &&
                        sibling->g->Cin == g.Cin && sibling->g->N == g.N;
      struct PublishSibB { Hold& t; const Tensor* y; int P; ~PublishSibB() { if (t.get()) conv_stats_publish(y, t.get(), P); } }
          publish_sibb{sibb_statt, sibf ? sibling->out : nullptr, (int)g.N};
      if (dg2b) {
        int64_t off1 = 0;
        sibb_wpk = Hold(packed_weights(second->w, *second->g, 1, st, &off1));
        sibb.wp = static_cast<const Tensor*>(sibb_wpk.get())->ptr<bf16_t>() + off1;
        sibb.x2 = second->dy->ptr<bf16_t>();
        if (second_fused) *second_fused = true;
      } else if (sibf) {
        int64_t off1 = 0;
        sibb_wpk = Hold(packed_weights(sibling->w, *sibling->g, 1, st, &off1));
        sibb.wp = static_cast<const Tensor*>(sibb_wpk.get())->ptr<bf16_t>();
        sibb.bias = sibling->bias ? sibling->bias->ptr<bf16_t>() : (const bf16_t*)nullptr;
        sibb.y = sibling->out->ptr<bf16_t>();
        if (statp) {
          int64_t ps[1] = {(int64_t)g.N * CO * 3};
          sibb_statt = Hold(new_tensor(ps, 1, kF32, in->device()));
          sibb.stats = sibb_statt->ptr<float>();
        }
        if (sibling_fused) *sibling_fused = true;
      }
      if (kernel_c) {   // 64-row weight stages, four-slot ring (LAMP_IG_VARIANT=b: the 128-row kernel)
        const size_t ldsc = (size_t)2 * 64 * KP * 2 + KP * 2 + 4 * (64 * 64 * 2);
        if (dg2b) {
          allow_big_lds((const void*)ig_conv8c_kernel<3, 2>);
          hipLaunchKernelGGL((ig_conv8c_kernel<3, 2>), dim3(blocksb), dim3(256), ldsc, st, in->ptr<bf16_t>(), wpp, bpb, out->ptr<bf16_t>(), (int)g.N, CI, KP, CO, statp, addbc, sibb);
        } else if (KS == 3) {
          allow_big_lds((const void*)ig_conv8c_kernel<3>);
          hipLaunchKernelGGL((ig_conv8c_kernel<3>), dim3(blocksb), dim3(256), ldsc, st, in->ptr<bf16_t>(), wpp, bpb, out->ptr<bf16_t>(), (int)g.N, CI, KP, CO, statp, addbc, sibb);
        } else {
          allow_big_lds((const void*)ig_conv8c_kernel<1>);
          hipLaunchKernelGGL((ig_conv8c_kernel<1>), dim3(blocksb), dim3(256), ldsc, st, in->ptr<bf16_t>(), wpp, bpb, out->ptr<bf16_t>(), (int)g.N, CI, KP, CO, statp, addbc, sibb);
        }
        LAMP_LAUNCH_CHECK();
        return;
      }
      // at most one workgroup per CU (small batches): eight waves per image pair (LAMP_IG_W8=0: always four); at most one IMAGE per CU: one
      // image per workgroup (LAMP_IG_ONE_IMAGE=0: pairs)
      static const bool w8_on = [] { const char* e = getenv("LAMP_IG_W8"); return !(e && e[0] == '0'); }();
      static const bool one_on = [] { const char* e = getenv("LAMP_IG_ONE_IMAGE"); return !(e && e[0] == '0'); }();
      const bool w8 = w8_on && blocksb <= num_cus();
      const bool one = one_on && w8 && g.N <= num_cus();
      const size_t ldsb = (size_t)(one ? 1 : 2) * 64 * KP * 2 + KP * 2 + (one ? 4 : 2) * IG_WTILE;
#define IG_LAUNCH_B(KS_, NWV_, SIBM_, NW_)                                                                                                      \
  do {                                                                                                                                          \
    allow_big_lds((const void*)ig_conv8b_kernel<KS_, NWV_, SIBM_, NW_>);                                                                       \
    hipLaunchKernelGGL((ig_conv8b_kernel<KS_, NWV_, SIBM_, NW_>), dim3(NW_ == 1 ? (int)g.N : blocksb), dim3(NWV_ * 64), ldsb, st,               \
                       in->ptr<bf16_t>(), wpp, bpb, out->ptr<bf16_t>(), (int)g.N, CI, KP, CO, statp, addbc, sibb);                               \
  } while (0)
      if (dg2b) { if (one) IG_LAUNCH_B(3, 8, 2, 1); else if (w8) IG_LAUNCH_B(3, 8, 2, 2); else IG_LAUNCH_B(3, 4, 2, 2); }
      else if (sibf) { if (one) IG_LAUNCH_B(3, 8, 1, 1); else if (w8) IG_LAUNCH_B(3, 8, 1, 2); else IG_LAUNCH_B(3, 4, 1, 2); }
      else if (KS == 3) { if (one) IG_LAUNCH_B(3, 8, 0, 1); else if (w8) IG_LAUNCH_B(3, 8, 0, 2); else IG_LAUNCH_B(3, 4, 0, 2); }
      else { if (one) IG_LAUNCH_B(1, 8, 0, 1); else if (w8) IG_LAUNCH_B(1, 8, 0, 2); else IG_LAUNCH_B(1, 4, 0, 2); }
#undef IG_LAUNCH_B
      LAMP_LAUNCH_CHECK();
      return;
    }
  }
  const int NW = g.N >= 1024 ? 4 : 2;        // images per workgroup (keep >= 256 workgroups before widening)
  const size_t lds = (size_t)NW * 100 * KP * 2 + 3 * IG_WTILE;
  const int blocks = (int)((g.N + NW - 1) / NW);
  // fprop and dgrad are the SAME kernel (ig_conv8_kernel), so they share one timer class
  KernelTimer kt("conv_igemm_fprop_dgrad", conv_flops(g), conv_bytes(g, 2), st);
  const bf16_t* bp = bias ? bias->ptr<bf16_t>() : (const bf16_t*)nullptr;
#define IG_LAUNCH(KS_, NW_)                                                                                                        \
  do {                                                                                                                             \
    static bool attr = false;                                                                                                      \
    if (!attr) {                                                                                                                   \
      allow_big_lds((const void*)ig_conv8_kernel<KS_, NW_>); \
      attr = true;                                                                                                                 \
    }                                                                                                                              \
    hipLaunchKernelGGL((ig_conv8_kernel<KS_, NW_>), dim3(blocks), dim3(NW_ * 128), lds, st, in->ptr<bf16_t>(), wpp, bp, \
                       out->ptr<bf16_t>(), (int)g.N, CI, KP, CO);                                                                  \
  } while (0)
  if (KS == 3) { if (NW == 4) IG_LAUNCH(3, 4); else IG_LAUNCH(3, 2); }
  else { if (NW == 4) IG_LAUNCH(1, 4); else IG_LAUNCH(1, 2); }
#undef IG_LAUNCH
  LAMP_LAUNCH_CHECK();
}

// dx = dgrad3x3(dy, w) + dgrad1x1(dy1, w1) [+ addend] in one launch of the eight-image kernel (the two products share the accumulators: one
// rounding); false (nothing launched) when that kernel does not take the geometry
bool igemm_conv_dgrad_pair(const Tensor* dy, const Tensor* w, const ConvGeom& g, const Tensor* dy1, const Tensor* w1, const ConvGeom& g1, Tensor* dx,
                           hipStream_t st, const Tensor* addend, bool* addend_fused) {
  if (addend_fused) *addend_fused = false;
  static const bool on = [] { const char* e = getenv("LAMP_CONV_DGRAD_PAIR"); return !(e && e[0] == '0'); }();
  if (!on || !ig_qualifies(g, dy->dtype) || !ig_qualifies(g1, dy1->dtype) || g.kh != 3 || g1.kh != 1) return false;
  if (g.Cout != g1.Cout || g.Cin != g1.Cin || g.N != g1.N) return false;
  // every kernel of run_conv8's default variant takes a second source (eight images per workgroup at large batches, two below)
  const char* variant = getenv("LAMP_IG_VARIANT");
  if (variant && variant[0] == 'a') return false;
  const SecondGradConv sg{dy1, w1, &g1};
  bool fused = false;
  run_conv8(dy, w, nullptr, dx, g, true, st, addend, addend_fused, nullptr, nullptr, nullptr, nullptr, &sg, &fused);
  LAMP_CHECK(fused, "internal: the eight-image kernel did not take the second gradient");
  return true;
}

bool igemm_conv_fwd(const Tensor* x, const Tensor* w, const Tensor* bias, Tensor* y, const ConvGeom& g, hipStream_t st) {
  if (!ig_qualifies(g, x->dtype)) return false;
  run_conv8(x, w, bias, y, g, false, st);
  return true;
}
static bool ig_fwd_folds_affine(const ConvGeom& g, int dtype);
// y = conv3x3(x, w, bias) and y1 = conv1x1(x, w1, bias1) - two convolutions of ONE input with the same number of output channels (the two
// branches of lamp's residual block, cnn.scala:16-20) - in one launch of the eight-image kernel: false (nothing launched) when that kernel
// does not take the geometry; the values of both outputs and of the statistics hand-offs are those of the two separate launches
bool igemm_conv_fwd_pair(const Tensor* x, const Tensor* w, const Tensor* bias, Tensor* y, const ConvGeom& g, const Tensor* w1, const Tensor* bias1,
                         Tensor* y1, const ConvGeom& g1, hipStream_t st) {
  if (!ig_qualifies(g, x->dtype) || !ig_qualifies(g1, x->dtype) || g.kh != 3 || g1.kh != 1) return false;
  if (g.Cout != g1.Cout || g.Cin != g1.Cin || g.N != g1.N) return false;
  // the eight-image kernel (its conditions), or - small batches - the two-image kernel with 128-row weight stages (more than 64 output channels)
  const char* variant = getenv("LAMP_IG_VARIANT");
  if (variant && variant[0] == 'a') return false;
  if (!ig_fwd_folds_affine(g, x->dtype) && !(g.Cout > 64 || (variant && variant[0] == 'b'))) return false;
  static const bool sib_on = [] { const char* e = getenv("LAMP_CONV_SIBLING"); return !(e && e[0] == '0'); }();
  if (!sib_on) return false;
  const SiblingConv sc{w1, bias1, y1, &g1};
  bool fused = false;
  run_conv8(x, w, bias, y, g, false, st, nullptr, nullptr, nullptr, nullptr, &sc, &fused);
  LAMP_CHECK(fused, "internal: the eight-image kernel did not take the sibling convolution");
  return true;
}
// y = conv(relu(bn(x))) with the batch norm given as its per-channel table: true only if a kernel that applies it while staging ran
// the eight-image kernel's conditions (run_conv8)
static bool ig_fwd_folds_affine(const ConvGeom& g, int dtype) {
  if (!ig_qualifies(g, dtype)) return false;
  const char* variant = getenv("LAMP_IG_VARIANT");
  if (variant && (variant[0] == 'a' || variant[0] == 'b')) return false;
  static const bool small_d = [] { const char* e = getenv("LAMP_IG_SMALL_D"); return !(e && e[0] == '0'); }();
  const bool force_d = variant && variant[0] == 'd';
  return (g.Cout > 64 || small_d) && (force_d || g.N >= 4 * (int64_t)num_cus());
}
// the eight-wave weight-gradient kernel's conditions (igemm_conv_wgrad)
static bool ig_wgrad_folds_affine(const ConvGeom& g, int dtype) {
  static const bool wide_on = [] { const char* e = getenv("LAMP_WGRAD_WIDE"); return !(e && e[0] == '0'); }();
  return ig_qualifies(g, dtype) && wide_on && g.kh == 3 && g.Cin > WG_CI && g.Cout > 64;
}
// both directions apply the batch-norm table while staging: only then does folding the batch norm into this convolution save a pass
bool igemm_conv_folds_affine(const ConvGeom& g, int dtype) { return ig_fwd_folds_affine(g, dtype) && ig_wgrad_folds_affine(g, dtype); }
bool igemm_conv_fwd_affine(const Tensor* x, const Tensor* affine, const Tensor* w, const Tensor* bias, Tensor* y, const ConvGeom& g, hipStream_t st) {
  if (!ig_fwd_folds_affine(g, x->dtype)) return false;           // checked up front: nothing must be launched when the conditions do not hold
  bool used = false;
  run_conv8(x, w, bias, y, g, false, st, nullptr, nullptr, affine, &used);
  LAMP_CHECK(used, "internal: the eight-image kernel did not run");
  return true;
}
bool igemm_conv_dgrad(const Tensor* dy, const Tensor* w, Tensor* dx, const ConvGeom& g, hipStream_t st, const Tensor* addend, bool* addend_fused) {
  if (addend_fused) *addend_fused = false;
  if (!ig_qualifies(g, dy->dtype)) return false;
  run_conv8(dy, w, nullptr, dx, g, true, st, addend, addend_fused);
  return true;
}
// affine (optional): x is the raw output of the producing convolution and the forward multiplied relu(bn(x)) - only the eight-wave kernel
// rebuilds it while staging: false (nothing launched) when the geometry takes another kernel
// images per workgroup at least: fewer images per workgroup = more, shorter workgroups and more partial sums.  8 at large batches; 2 at
// N <= 512, where the launch is a latency chain and the partial sums are small (B = 256: 0.5606 -> 0.5538 ms per step, round 5).
// LAMP_WGRAD_MIN_IPS overrides both.
static int wgrad_min_ips(int64_t N) {
  static const int v = [] { const char* e = getenv("LAMP_WGRAD_MIN_IPS"); return e ? std::max(2, atoi(e)) : 0; }();
  return v ? v : (N <= 512 ? 2 : 8);
}
// ---- the eight-wave weight-gradient kernel: one launch for one layer (optionally with its sibling 1x1: `second`) or for TWO layers (round 6) ----
struct SecondWgradConv;
struct Wg8hProblem { const Tensor* dy; const Tensor* x; Tensor* dw; ConvGeom g; const Tensor* affine; };
static void wg8h_launch(const Wg8hProblem& a, const Wg8hProblem* b, hipStream_t st, const SecondWgradConv* second);
// Two layers in one launch.  A layer's launch leaves (256 workgroups) x (its accumulators) of partial sums - 33 MB written by the kernel and read
// again by the batched reduction, whatever the layer's size; with two layers' workgroups side by side each layer is walked by half the workgroups,
// twice as many images each, and leaves half of that.  Nothing in backprop waits for a weight gradient (only the optimiser reads it), so a layer
// that qualifies is PARKED here - its tensors retained, its gradient's storage marked pending like a deferred reduction's - until the next one
// arrives (res4's second convolution waits for res3's, three kernels later) or anything flushes: the end of backprop, a read of the gradient
// (Tensor::raw -> resolve_deferred), a stream / device synchronisation, the end of a graph capture.  LAMP_WGRAD_GROUP=0: every layer at once.
namespace {
struct ParkedWgrad { Tensor* dy; Tensor* x; Tensor* dw; ConvGeom g; Tensor* affine; hipStream_t st; int device; uint64_t vdy, vx; std::thread::id owner; };
std::mutex g_wgpark_mu;
std::vector<ParkedWgrad> g_wgpark;
void wgpark_release(ParkedWgrad& p) { release(p.dy); release(p.x); release(p.dw); if (p.affine) release(p.affine); }
void wgpark_check(const ParkedWgrad& p) {
  // (x is a forward activation: nothing writes it during backprop, but the batch-norm backward that runs meanwhile takes its pointer through the
  // mutable accessor, which counts as a write - only the gradient's version is held to its value at parking)
  LAMP_CHECK(p.dy->st->version.load(std::memory_order_relaxed) == p.vdy, "internal: the output gradient of a parked weight gradient was written before its launch");
}
}  // namespace
// the parked layers, each alone (called with nothing of this file's locks held).  An entry belongs to the thread that parked it, like a deferred
// reduction: flush_deferred() launches the CALLER's (all = false) - a replica thread of the single-process data-parallel step must not launch
// another's half-finished pass early -, a read of a pending gradient (resolve_deferred) launches everybody's: the launch goes to the entry's stream
void igemm_wgrad_flush_parked(bool all) {
  std::vector<ParkedWgrad> v;
  {
    std::lock_guard<std::mutex> lk(g_wgpark_mu);
    const std::thread::id me = std::this_thread::get_id();
    size_t keep = 0;
    for (size_t i = 0; i < g_wgpark.size(); i++) {
      if (all || g_wgpark[i].owner == me) v.push_back(g_wgpark[i]);
      else g_wgpark[keep++] = g_wgpark[i];
    }
    g_wgpark.resize(keep);
  }
  for (auto& p : v) {
    struct Rel { ParkedWgrad& p; ~Rel() { wgpark_release(p); } } rel{p};
    wgpark_check(p);
    const int prev = current_device();
    if (prev != p.device) set_device(p.device);
    struct Back { int prev, dev; ~Back() { if (prev != dev) set_device(prev); } } back{prev, p.device};
    Wg8hProblem a{p.dy, p.x, p.dw, p.g, p.affine};
    wg8h_launch(a, nullptr, p.st, nullptr);
  }
}
static bool wg8h_group_defer(const Tensor* dy, const Tensor* x, Tensor* dw, const ConvGeom& g, hipStream_t st, const Tensor* affine) {
  static const bool on = [] { const char* e = getenv("LAMP_WGRAD_GROUP"); return !(e && e[0] == '0'); }();
  static const int shift_dy = [] { const char* e = getenv("LAMP_WGRAD_SHIFT_DY"); return e ? atoi(e) : 2; }();
  // large batches only (a workgroup still walks >= 16 images with half the workgroups), gradients the library owns (the pending flag is honoured
  // by every reader that goes through Tensor::raw), deferred reductions on
  if (!on || shift_dy < 2 || !wgrad_reduce_deferred() || !dw->st->owned || g.N < 8 * (int64_t)num_cus()) return false;
  ParkedWgrad mate{};
  bool have = false;
  {
    std::lock_guard<std::mutex> lk(g_wgpark_mu);
    const std::thread::id me = std::this_thread::get_id();
    for (size_t i = 0; i < g_wgpark.size(); i++)
      if (g_wgpark[i].owner == me && g_wgpark[i].st == st && g_wgpark[i].device == dw->device() && g_wgpark[i].g.N == g.N && g_wgpark[i].dw->st != dw->st) {
        mate = g_wgpark[i]; g_wgpark.erase(g_wgpark.begin() + i); have = true; break;
      }
    if (!have) {
      ParkedWgrad p{retain(const_cast<Tensor*>(dy)), retain(const_cast<Tensor*>(x)), retain(dw), g, affine ? retain(const_cast<Tensor*>(affine)) : nullptr, st,
                    dw->device(), dy->st->version.load(std::memory_order_relaxed), x->st->version.load(std::memory_order_relaxed), me};
      g_wgpark.push_back(p);
      dw->st->version.fetch_add(1, std::memory_order_relaxed);  // a writer like any other
      dw->st->pending.store(1, std::memory_order_release);
      return true;
    }
  }
  struct Rel { ParkedWgrad& p; ~Rel() { wgpark_release(p); } } rel{mate};
  wgpark_check(mate);
  Wg8hProblem a{mate.dy, mate.x, mate.dw, mate.g, mate.affine}, b{dy, x, dw, g, affine};
  wg8h_launch(a, &b, st, nullptr);
  return true;
}
// second (optional, the eight-wave kernel only): the output gradient of a sibling 1x1 convolution of the same x and the tensor that receives ITS
// weight gradient - both from the one launch (igemm_conv_wgrad_pair checks the conditions first)
struct SecondWgradConv { const Tensor* dy; Tensor* dw; const ConvGeom* g; };
static void wg8h_launch(const Wg8hProblem& pa, const Wg8hProblem* pb, hipStream_t st, const SecondWgradConv* second) {
  const bool pair = second != nullptr;
  LAMP_CHECK(!(pair && pb), "internal: the sibling pair and the two-layer group do not combine");
  const int RS = 9;
  const int cus = num_cus();
  struct Plan { int ntile, CIP, ips, nsplit, wgs; Hold partial; };
  auto plan = [&](const Wg8hProblem& p, int wg_budget) {
    Plan q;
    q.ntile = (int)((p.g.Cin + WG_CI - 1) / WG_CI);
    q.CIP = q.ntile * WG_CI;
    const int target = std::max(1, wg_budget / q.ntile);
    q.ips = (int)std::max<int64_t>(1, (p.g.N + target - 1) / target);
    if (q.ips < wgrad_min_ips(p.g.N) && p.g.N >= wgrad_min_ips(p.g.N)) q.ips = wgrad_min_ips(p.g.N);
    q.nsplit = (int)((p.g.N + q.ips - 1) / q.ips);
    q.wgs = q.ntile * q.nsplit;
    int64_t ps[1] = {(int64_t)q.nsplit * RS * IG_M * q.CIP};
    q.partial = Hold(new_tensor(ps, 1, kF32, p.x->device()));
    return q;
  };
  // two layers: half the CUs' workgroups each, the first problem's count a multiple of 8 (the kernel's XCD mapping)
  Plan A = plan(pa, pb ? cus / 2 : cus), B;
  if (pb) {
    B = plan(*pb, cus / 2);
    if (A.wgs % 8 != 0) {                                     // (a ragged first problem would shift the second one's XCDs: not grouped)
      Wg8hProblem a1 = pa, b1 = *pb;
      wg8h_launch(a1, nullptr, st, nullptr);
      wg8h_launch(b1, nullptr, st, nullptr);
      return;
    }
  }
  static const int shift_dy = [] { const char* e = getenv("LAMP_WGRAD_SHIFT_DY"); return e ? atoi(e) : 2; }();   // 0: off, 1: dY rows, 2: + X columns in registers
  LAMP_CHECK(!(pair || pb) || shift_dy >= 2, "internal: the weight-gradient pair / group needs the register-shifted X fragments");
  Hold partial2;
  if (pair) { int64_t ps2[1] = {(int64_t)A.nsplit * IG_M * A.CIP}; partial2 = Hold(new_tensor(ps2, 1, kF32, pa.x->device())); }
  const size_t lds = pair ? 4 * (size_t)(2 * IG_WTILE + WG_XCOPY + 512) : 4 * (size_t)WG_STAGE;
  {
    const double sec_fl = pair ? conv_flops(*second->g) : pb ? conv_flops(pb->g) : 0.0;
    const double sec_by = pair ? conv_bytes(*second->g, 2) - (double)pa.g.N * pa.g.Cin * 64 * 2 : pb ? conv_bytes(pb->g, 2) : 0.0;   // (pair: x is the one already counted)
    KernelTimer kt("conv_wgrad_igemm", conv_flops(pa.g) + sec_fl, conv_bytes(pa.g, 2) + sec_by, st);
    const bf16_t* dy2p = pair ? second->dy->ptr<bf16_t>() : (const bf16_t*)nullptr;
    float* p2p = pair ? partial2->ptr<float>() : (float*)nullptr;
    const int co2 = pair ? (int)second->g->Cout : 0;
    const float4* affp = pa.affine ? reinterpret_cast<const float4*>(pa.affine->ptr<float>()) : (const float4*)nullptr;
    WgSecondProblem sp{};
    if (pb) {
      sp.dy = pb->dy->ptr<bf16_t>(); sp.x = pb->x->ptr<bf16_t>(); sp.partial = B.partial->ptr<float>();
      sp.CO = (int)pb->g.Cout; sp.CI = (int)pb->g.Cin; sp.CIP = B.CIP; sp.images_per_split = B.ips; sp.ntile = B.ntile;
      sp.affine = pb->affine ? reinterpret_cast<const float4*>(pb->affine->ptr<float>()) : (const float4*)nullptr;
      sp.wgs0 = A.wgs;
    }
    static const bool wg_stagger = [] { const char* e = getenv("LAMP_WGRAD_STAGGER"); return !(e && e[0] == '0'); }();
    static const bool wg_prio = [] { const char* e = getenv("LAMP_WGRAD_PRIO"); return !(e && e[0] == '0'); }();
    // LAMP_WG8H_DMA=0 (run time): the dY tiles through registers and ds_write instead of LDS-DMA - the same arithmetic in the same order, kept
    // as the fallback and the bitwise A/B of the hand-counted vmcnt waits (tests/test_ops_gpu.py)
    static const bool wg_dma = [] { const char* e = getenv("LAMP_WG8H_DMA"); return e ? e[0] != '0' : (LAMP_WG8H_DMA != 0); }();
    const int grid = A.wgs + (pb ? B.wgs : 0);
#define IG_LAUNCH_WG8H(M_, P_)                                                                                                              \
  do {                                                                                                                                      \
    if (M_ == 2 && !wg_dma) { IG_LAUNCH_WG8H_(M_, P_, false); } else { IG_LAUNCH_WG8H_(M_, P_, true); }                                       \
  } while (0)
#define IG_LAUNCH_WG8H_(M_, P_, D_)                                                                                                         \
  do {                                                                                                                                      \
    allow_big_lds((const void*)ig_wgrad8h_kernel<M_, P_, D_>);                                                                             \
    hipLaunchKernelGGL((ig_wgrad8h_kernel<M_, P_, D_>), dim3(grid), dim3(512), lds, st, pa.dy->ptr<bf16_t>(), pa.x->ptr<bf16_t>(),          \
                       A.partial->ptr<float>(), (int)pa.g.N, (int)pa.g.Cout, (int)pa.g.Cin, A.CIP, A.ips, A.ntile, affp,                    \
                       (wgrad_reduce_deferred() ? 1 : 0) | (wg_stagger ? 0 : 2) | (wg_prio ? 4 : 0), dy2p, p2p, co2, sp);                   \
  } while (0)
    if (pair) IG_LAUNCH_WG8H(2, true);
    else if (shift_dy >= 2) IG_LAUNCH_WG8H(2, false); else if (shift_dy == 1) IG_LAUNCH_WG8H(1, false); else IG_LAUNCH_WG8H(0, false);
#undef IG_LAUNCH_WG8H
#undef IG_LAUNCH_WG8H_
    LAMP_LAUNCH_CHECK();
  }
  auto enqueue = [&](const Wg8hProblem& p, Plan& q) {
    const int64_t cols = (int64_t)RS * IG_M * q.CIP / 4;
    WgradReduceArgs ra{};
    ra.kind = 0; ra.CO = (int)p.g.Cout; ra.CI = (int)p.g.Cin; ra.CIP = q.CIP; ra.COP = IG_M; ra.RS = RS; ra.nsplit = q.nsplit; ra.blocks = (int)((cols + 31) / 32);
    wgrad_reduce_enqueue(ra, q.partial.get(), p.dw, st);
  };
  enqueue(pa, A);
  if (pb) enqueue(*pb, B);
  if (pair) {
    const int64_t cols2 = (int64_t)IG_M * A.CIP / 4;
    WgradReduceArgs rb{};
    rb.kind = 0; rb.CO = (int)second->g->Cout; rb.CI = (int)pa.g.Cin; rb.CIP = A.CIP; rb.COP = IG_M; rb.RS = 1; rb.nsplit = A.nsplit; rb.blocks = (int)((cols2 + 31) / 32);
    wgrad_reduce_enqueue(rb, partial2.get(), second->dw, st);
  }
}
static bool igemm_conv_wgrad_impl(const Tensor* dy, const Tensor* x, Tensor* dw, const ConvGeom& g, hipStream_t st, const Tensor* affine,
                                  const SecondWgradConv* second) {
  if (!ig_qualifies(g, x->dtype)) return false;
  const int KS = g.kh, RS = KS * KS;
  static const bool wide_on = [] { const char* e = getenv("LAMP_WGRAD_WIDE"); return !(e && e[0] == '0'); }();
  const float4* affp = affine ? reinterpret_cast<const float4*>(affine->ptr<float>()) : (const float4*)nullptr;
  if (affine && !ig_wgrad_folds_affine(g, x->dtype)) return false;
  if (wide_on && KS == 3 && g.Cin > WG_CI && g.Cout > 64) {
    // eight-wave kernel: the v2 decomposition (32-channel slice of Cin, image range) with two waves per SIMD
    if (!second && wg8h_group_defer(dy, x, dw, g, st, affine)) return true;      // launched with the next such layer's, or by the flush (below)
    Wg8hProblem a{dy, x, dw, g, affine};
    wg8h_launch(a, nullptr, st, second);
    return true;
  }
  {
    // v2: workgroup = (32-channel slice of Cin, image range), all taps in registers
    // 1x1 with more than one slice of Cin: one workgroup owns all of them (CIT = 4), dY is read once
    const bool all_ci = KS == 1 && g.Cin > WG_CI;
    const int ntile = all_ci ? 1 : (int)((g.Cin + WG_CI - 1) / WG_CI);
    const int CIP = g.Cin <= 16 ? 16 : (all_ci ? 4 : ntile) * WG_CI;   // columns of a partial-sum tile (see tile_active in the kernel)
    const bool narrow = g.Cout <= 64 || g.Cin <= 16;               // the NARROW instantiation (branches in the MFMA chain) only where it pays
    const int COP = narrow ? (int)((g.Cout + 15) / 16) * 16 : IG_M;  // rows
    static const int wgs_per_cu = [] { const char* e = getenv("LAMP_WGRAD_WGS_PER_CU"); return e ? std::max(1, atoi(e)) : 1; }();
    // one workgroup per CU: half the partial-sum traffic of two, same speed
    static const int narrow_per_cu = [] { const char* e = getenv("LAMP_WGRAD_NARROW_PER_CU"); return e ? std::max(1, atoi(e)) : 1; }();   // measured: 2 and 3 are 1 % slower on the step
    int target = std::max(1, (num_cus() * (narrow ? narrow_per_cu : wgs_per_cu)) / ntile);
    int ips = (int)std::max<int64_t>(1, (g.N + target - 1) / target);
    if (ips < wgrad_min_ips(g.N) && g.N >= wgrad_min_ips(g.N)) ips = wgrad_min_ips(g.N);
    const int nsplit = (int)((g.N + ips - 1) / ips);
    int64_t ps[1] = {(int64_t)nsplit * RS * COP * CIP};
    Hold partial(new_tensor(ps, 1, kF32, x->device()));
    const bool pair = second != nullptr;
    if (pair && (KS != 3 || all_ci)) return false;              // (igemm_conv_wgrad_pair only asks for a 3x3)
    const int COP2 = pair ? (int)((second->g->Cout + 15) / 16) * 16 : 0;
    Hold partial2;
    if (pair) { int64_t ps2[1] = {(int64_t)nsplit * COP2 * CIP}; partial2 = Hold(new_tensor(ps2, 1, kF32, x->device())); }
    const size_t lds = all_ci ? 2 * (size_t)(IG_WTILE + 4 * WG_XCOPY + 512) : 2 * (size_t)(WG_STAGE + (pair ? IG_WTILE : 0));
    {
      const double sec_fl = pair ? conv_flops(*second->g) : 0.0;
      const double sec_by = pair ? conv_bytes(*second->g, 2) - (double)g.N * g.Cin * 64 * 2 : 0.0;
      KernelTimer kt("conv_wgrad_igemm", conv_flops(g) + sec_fl, conv_bytes(g, 2) + sec_by, st);
      static const bool ci16_on = [] { const char* e = getenv("LAMP_WGRAD_CI16"); return !(e && e[0] == '0'); }();
      const bool ci16 = ci16_on && KS == 3 && CIP == 16;           // (then `narrow` holds: Cin <= 16)
      const void* kfn = ci16    ? (pair ? (const void*)ig_wgrad8v2_kernel<3, true, 1, true, true> : (const void*)ig_wgrad8v2_kernel<3, true, 1, false, true>)
                      : pair    ? (narrow ? (const void*)ig_wgrad8v2_kernel<3, true, 1, true> : (const void*)ig_wgrad8v2_kernel<3, false, 1, true>)
                      : KS == 3 ? (narrow ? (const void*)ig_wgrad8v2_kernel<3, true, 1> : (const void*)ig_wgrad8v2_kernel<3, false, 1>)
                      : all_ci  ? (narrow ? (const void*)ig_wgrad8v2_kernel<1, true, 4> : (const void*)ig_wgrad8v2_kernel<1, false, 4>)
                                : (narrow ? (const void*)ig_wgrad8v2_kernel<1, true, 1> : (const void*)ig_wgrad8v2_kernel<1, false, 1>);
      allow_big_lds(kfn);
      const bf16_t* dyp = dy->ptr<bf16_t>(); const bf16_t* xp = x->ptr<bf16_t>(); float* pp = partial->ptr<float>();
      const bf16_t* dy2p = pair ? second->dy->ptr<bf16_t>() : (const bf16_t*)nullptr;
      float* p2p = pair ? partial2->ptr<float>() : (float*)nullptr;
      int a_N = (int)g.N, a_CO = (int)g.Cout, a_CI = (int)g.Cin, a_CIP = CIP, a_ips = ips, a_COP = COP, a_ntile = ntile;
      int a_CO2 = pair ? (int)second->g->Cout : 0, a_COP2 = COP2;
      void* args[] = {(void*)&dyp, (void*)&xp, (void*)&pp, (void*)&a_N, (void*)&a_CO, (void*)&a_CI, (void*)&a_CIP, (void*)&a_ips, (void*)&a_COP, (void*)&a_ntile,
                      (void*)&dy2p, (void*)&p2p, (void*)&a_CO2, (void*)&a_COP2};
      HIP_CHECK(hipLaunchKernel(kfn, dim3(ntile * nsplit), dim3(256), args, lds, st));
      LAMP_LAUNCH_CHECK();
    }
    const int64_t cols = (int64_t)RS * COP * CIP / 4;
    WgradReduceArgs ra{};
    ra.kind = 0; ra.CO = (int)g.Cout; ra.CI = (int)g.Cin; ra.CIP = CIP; ra.COP = COP; ra.RS = RS; ra.nsplit = nsplit; ra.blocks = (int)((cols + 31) / 32);
    wgrad_reduce_enqueue(ra, partial.get(), dw, st);
    if (pair) {
      const int64_t cols2 = (int64_t)COP2 * CIP / 4;
      WgradReduceArgs rb{};
      rb.kind = 0; rb.CO = (int)second->g->Cout; rb.CI = (int)g.Cin; rb.CIP = CIP; rb.COP = COP2; rb.RS = 1; rb.nsplit = nsplit; rb.blocks = (int)((cols2 + 31) / 32);
      wgrad_reduce_enqueue(rb, partial2.get(), second->dw, st);
    }
    return true;
  }
}
bool igemm_conv_wgrad(const Tensor* dy, const Tensor* x, Tensor* dw, const ConvGeom& g, hipStream_t st, const Tensor* affine) {
  return igemm_conv_wgrad_impl(dy, x, dw, g, st, affine, nullptr);
}
// dw = wgrad3x3(dy, x) and dw1 = wgrad1x1(dy1, x) from one launch of the eight-wave kernel (x staged once; the sibling's output gradient sits in
// the LDS the shifted X copies no longer need): the values of the two separate launches up to the order the image ranges are summed in for dw1;
// false = nothing launched
bool igemm_conv_wgrad_pair(const Tensor* dy, const Tensor* dy1, const Tensor* x, Tensor* dw, Tensor* dw1, const ConvGeom& g, const ConvGeom& g1, hipStream_t st) {
  static const bool on = [] { const char* e = getenv("LAMP_CONV_WGRAD_PAIR"); return !(e && e[0] == '0'); }();
  static const bool wide_on = [] { const char* e = getenv("LAMP_WGRAD_WIDE"); return !(e && e[0] == '0'); }();
  static const int shift_dy = [] { const char* e = getenv("LAMP_WGRAD_SHIFT_DY"); return e ? atoi(e) : 2; }();
  if (!on || !wide_on || shift_dy < 2) return false;
  if (!ig_qualifies(g, x->dtype) || !ig_qualifies(g1, x->dtype) || g.kh != 3 || g1.kh != 1) return false;
  if (g.Cin != g1.Cin || g.N != g1.N) return false;
  // (more than 32 input and 64 output channels: the eight-wave kernel; else the four-wave kernel, one Cin slice per workgroup)
  const SecondWgradConv sw{dy1, dw1, &g1};
  return igemm_conv_wgrad_impl(dy, x, dw, g, st, nullptr, &sw);
}

}  // namespace lamp

#ifdef LAMP_WG8H_STAMPS
extern "C" int lamp_debug_wg8h_stamps(unsigned int* out, int n) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lamp::g_wg8h_stamps), (size_t)n * 4);
}
#endif
