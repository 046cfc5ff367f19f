// Matrix-core convolution for the NARROW bf16 layers of the CIFAR ResNet (3..16 channels on 32x32 / 16x16 / 8x8
// maps: stem 3->6 5x5, res1 6->6 (stride 2), res2 6->16 (stride 2) and their 1x1 projections; reference
// workload example-cifar100/.../cnn.scala:89-131, op ops.scala:1547-1651).
//
// These layers are bound by activation traffic (a few FLOPs per byte), but a scalar-FMA direct convolution
// still needs ~100 VALU instructions per output pixel and ends up 3-10x above the HBM time.  Here the
// multiply-accumulates go to v_mfma_f32_16x16x32_bf16 with a formulation that needs NO im2col gather:
//
//   fprop / dgrad:   D[pixel][co] = sum_{(c, r)} sum_{j = 0..7} X[c][ho*sh + r][wo*sw + j] * Wk[(c, r)][j][co]
//     the K dimension is (channel, filter row) x an 8-wide window of the image row; Wk is the filter row
//     zero-padded from kw to 8 taps.  The A fragment of a lane (8 consecutive k of one pixel) is then 16
//     CONTIGUOUS bytes of the LDS image, one ds_read_b128 at 2-byte alignment (gfx950 has unaligned DS
//     access), the B fragments (packed weights) live in registers for the whole kernel.  The padding costs
//     MFMA work only, of which there is plenty to spare.  dgrad = fprop over the stride-dilated dy image
//     (zeros inserted while staging) with the filter mirrored and Cin/Cout swapped.
//   wgrad:           D[co][(ci, r, s)] = sum_{pixels} dY[co][p] * X[ci][ho*sh - ph + r][wo*sw - pw + s]
//     K runs over output pixels (8 consecutive pixels of a row per lane): A = 16 aligned bytes of dY,
//     B = 16 contiguous bytes of the shifted image row (stride 2: of its even / odd column plane).
//
// One workgroup stages one image (or a few 8x8 ones) in LDS with coalesced 16-byte loads and walks a strided
// range of images; every kernel reads each activation once and writes each result once.
// Accumulation is fp32 in a fixed order: results are bitwise reproducible.
#include "device_utils.h"
#include "conv_geom.h"
#include "wgrad_reduce.h"
#include <map>
#include <type_traits>
#include <mutex>
#include <tuple>
#include <vector>
#include "conv_narrow_pack.h"

namespace lamp {

typedef nv_s8 __attribute__((aligned(2))) nv_s8_u;      // 16-byte LDS read at 2-byte alignment
typedef float nv_f4 __attribute__((ext_vector_type(4)));

struct NcvGeom {
  int N, C, CO;          // images, loop (K-side) channels, output channels
  int H, W;              // staged tensor [C][H][W]
  int Hs, Ws;            // LDS image [C][Hs][Ws]
  int top, left, dil;    // staged pixel (a, b) lands at (top + a*dil, left + b*dil)
  int Ho, Wo;            // output map
  int sh, sw, wx;        // window of output (ho, wo), filter row r: LDS row ho*sh + r, columns wo*sw + wx + [0, 8)
  int kh;
  int kh_inv;            // 65536 / kh + 1: pair / kh == (pair * kh_inv) >> 16 for pair < 4096 (no integer division in the prologue)
  int pf;                // the image fits the register prefetch (W % 8 == 0 and at most NCV_PF 16-byte packets per thread)
  // ncv_fwd2_kernel, dgrad of a PAIR (round 5): channels [0, C1) of the staged image come from the first tensor and take part with all kh
  // filter rows, channels [C1, C) come from a second tensor (the gradient of a sibling 1x1 convolution of the same input) and take part with
  // the centre row only: K pairs (c, r) = C1 * kh + (C - C1).  C1 = C: one source.
  int C1;
};

constexpr int NCV_LEFT = 8;   // staged images start at column 8: 16-byte aligned rows for the vector copy

// (the weight-fragment layout, ncv_weight_frag and NcvPackMany live in conv_narrow_pack.h: conv_igemm.hip packs these images in its own launch)
__global__ __launch_bounds__(256) void ncv_pack_kernel(NcvPackMany a) { ncv_pack_body(a, blockIdx.y, blockIdx.x * 256 + threadIdx.x); }

// Image staging in two halves, so that the NEXT image's global loads are in flight while the current one is multiplied: the image
// [C][H][W] is read as consecutive 16-byte packets (W % 8 == 0), packet tid + k * nthreads into register k of the thread, and
// written to the LDS image [C][Hs][Ws] at (top + a*dil, left + b*dil) once the previous image's reads are done.  A thread handles
// the same packets of every image, so their LDS destinations are computed once (ncv_stage_plan): -1 = no packet.
constexpr int NCV_PF = 4;
struct NcvPre { uint4 v[NCV_PF]; };
struct NcvPlan { int dst[NCV_PF]; int src[NCV_PF]; };      // src: element offset in the image's source tensor, + (1 << 30) for the second one
__device__ __forceinline__ NcvPlan ncv_stage_plan(const NcvGeom& q, int tid, int nthreads) {
  NcvPlan pl;
  const int rc = q.W >> 3, total = q.C * q.H * rc, first = q.C1 * q.H * rc;
#pragma unroll
  for (int k = 0; k < NCV_PF; k++) {
    const int i = tid + k * nthreads;
    const int b = i % rc, a = (i / rc) % q.H, c = i / (rc * q.H);
    pl.dst[k] = (q.pf && i < total) ? (c * q.Hs + q.top + a * q.dil) * q.Ws + q.left + b * 8 * q.dil : -1;
    pl.src[k] = i < first ? i * 8 : ((i - first) * 8) | (1 << 30);
  }
  return pl;
}
__device__ __forceinline__ void ncv_stage_load(NcvPre& r, const NcvPlan& pl, const bf16_t* __restrict__ sp, const bf16_t* __restrict__ sp2) {
#pragma unroll
  for (int k = 0; k < NCV_PF; k++)
    if (pl.dst[k] >= 0) r.v[k] = *reinterpret_cast<const uint4*>(((pl.src[k] >> 30) ? sp2 : sp) + (pl.src[k] & ((1 << 30) - 1)));
}
__device__ __forceinline__ void ncv_stage_store(unsigned short* xs, const NcvPre& r, const NcvPlan& pl, int dil) {
#pragma unroll
  for (int k = 0; k < NCV_PF; k++) {
    if (pl.dst[k] < 0) continue;
    if (dil == 1) *reinterpret_cast<uint4*>(xs + pl.dst[k]) = r.v[k];
    else {
      unsigned short* d = xs + pl.dst[k];
      const unsigned int u[4] = {r.v[k].x, r.v[k].y, r.v[k].z, r.v[k].w};
#pragma unroll
      for (int j = 0; j < 4; j++) { d[(2 * j) * dil] = (unsigned short)(u[j] & 0xffffu); d[(2 * j + 1) * dil] = (unsigned short)(u[j] >> 16); }
    }
  }
}

__device__ __forceinline__ void ncv_stage(unsigned short* xs, const bf16_t* __restrict__ sp, const NcvGeom& q, int tid, int nthreads) {
  if (q.dil == 1) {
    const int rc = q.W >> 3, total = q.C * q.H * rc;
    for (int i = tid; i < total; i += nthreads) {
      const int b = i % rc, a = (i / rc) % q.H, c = i / (rc * q.H);
      const uint4 v = *reinterpret_cast<const uint4*>(sp + (c * q.H + a) * q.W + b * 8);
      *reinterpret_cast<uint4*>(xs + (c * q.Hs + q.top + a) * q.Ws + q.left + b * 8) = v;
    }
  } else {
    const int total = q.C * q.H * q.W;
    for (int i = tid; i < total; i += nthreads) {
      const int b = i % q.W, a = (i / q.W) % q.H, c = i / (q.W * q.H);
      xs[(c * q.Hs + q.top + a * q.dil) * q.Ws + q.left + b * q.dil] = sp[i].bits;
    }
  }
}

template <int NK>
__global__ __launch_bounds__(256) void ncv_fwd_kernel(const bf16_t* __restrict__ src, const nv_bf8* __restrict__ wpk, const bf16_t* __restrict__ bias,
                                                      bf16_t* __restrict__ dst, NcvGeom q) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* xs = reinterpret_cast<unsigned short*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int co = lane & 15;
  nv_bf8 wfr[NK];
  int koff[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ks++) {
    wfr[ks] = wpk[ks * 64 + lane];
    int pair = ks * 4 + (lane >> 4);
    if (pair >= q.C * q.kh) pair = 0;                    // padded k: weights are zero, any valid address will do
    const int c = (pair * q.kh_inv) >> 16, r = pair - c * q.kh;
    koff[ks] = (c * q.Hs + r) * q.Ws * 2;
  }
  const float bv = (bias && co < q.CO) ? (float)bias[co] : 0.f;
  const int HoWo = q.Ho * q.Wo, ntiles = (HoWo + 15) >> 4;
  const int img_elems = q.C * q.Hs * q.Ws;
  for (int o = tid * 8; o < img_elems; o += 256 * 8) *reinterpret_cast<uint4*>(xs + o) = make_uint4(0, 0, 0, 0);   // img_elems % 8 == 0
  bool first = true;
  for (int n = blockIdx.x; n < q.N; n += gridDim.x) {
    __syncthreads();                                     // zero fill / previous image's reads are done
    ncv_stage(xs, src + (int64_t)n * q.C * q.H * q.W, q, tid, 256);
    __syncthreads();
    first = false;
    bf16_t* yp = dst + (int64_t)n * q.CO * HoWo;
    for (int tile = wid; tile < ntiles; tile += 4) {
      const int p = min(tile * 16 + (lane & 15), HoWo - 1);
      const int ho = p / q.Wo, wo = p - ho * q.Wo;
      const char* base = smem + ((ho * q.sh) * q.Ws + wo * q.sw + q.wx) * 2;
      nv_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NK; ks++) {
        const nv_s8 a = *reinterpret_cast<const nv_s8_u*>(base + koff[ks]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(nv_bf8, a), wfr[ks], acc, 0, 0, 0);
      }
      // D: column = output channel (lane & 15), rows = 4 consecutive pixels
      const int p4 = tile * 16 + (lane >> 4) * 4;
      if (co < q.CO && p4 < HoWo) {
        const bf16_t o0(acc[0] + bv), o1(acc[1] + bv), o2(acc[2] + bv), o3(acc[3] + bv);
        uint2 pk;
        pk.x = (unsigned)o0.bits | ((unsigned)o1.bits << 16);
        pk.y = (unsigned)o2.bits | ((unsigned)o3.bits << 16);
        *reinterpret_cast<uint2*>(yp + co * HoWo + p4) = pk;
      }
    }
  }
  (void)first;
}

// Aligned-window variant.  An unaligned ds_read_b128 runs at ~1/11 of the aligned rate on gfx950
// (scripts/microbench/lds_unaligned.hip), and the windows of neighbouring pixels start 2 (stride 1) or 4 bytes
// apart.  So one MFMA tile takes 16 pixels whose windows have the SAME phase inside a 16-byte segment: columns
// cg*P + d with a fixed d (P = 8 / stride phases), from TR = 16 / (Wo / P) image rows.  The P tiles d = 0..P-1 of
// such a "super-tile" (TR full output rows) read the SAME two or three aligned segments per lane, which are loaded
// once and funnel-shifted in registers by a compile-time amount; a lane ends up holding P consecutive output
// pixels per accumulator register, stored as one 16- or 8-byte write.
//   PH0 = (window origin wx) & 7.
// -DNCV_STAMP (diagnostic builds only; scripts/ncv_stamp_probe.py): s_memtime stamps of thread 0 of the first 1024 workgroups
#if defined(NCV_SKIP) && (NCV_SKIP & 2)
#define NCV_STORE_COND && pk[0] == 0x12345678u
#else
#define NCV_STORE_COND
#endif
#ifdef NCV_STAMP
__device__ unsigned long long ncv_stamps[1024 * 8];
#define NCV_STAMP_AT(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) ncv_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define NCV_STAMP_ONCE(k) do { if (n == (int)blockIdx.x) NCV_STAMP_AT(k); } while (0)
#else
#define NCV_STAMP_AT(k) do { } while (0)
#define NCV_STAMP_ONCE(k) do { } while (0)
#endif
//   NS = shifts per MFMA (see NcvW): NS = 2 halves the P window phases an MFMA has to be issued for.
//   ADD: dst = round(round(conv) + add) - a separate instantiation (dgrad only: SW = 1) because the addend's registers cost the plain
//   kernels a wave of occupancy
template <int I0, int I1, class F> __device__ __forceinline__ void ncv_static_for(F&& f) {
  if constexpr (I0 < I1) { f(std::integral_constant<int, I0>{}); ncv_static_for<I0 + 1, I1>(f); }
}
struct NcvWf { float n, mean, m2; };
__device__ __forceinline__ NcvWf ncv_wf_merge(const NcvWf& a, const NcvWf& b) {   // Chan's merge; either side may be empty
  NcvWf r;
  r.n = a.n + b.n;
  if (r.n == 0.f) { r.mean = 0.f; r.m2 = 0.f; return r; }
  const float d = b.mean - a.mean, f = b.n / r.n;
  r.mean = a.mean + d * f;
  r.m2 = a.m2 + b.m2 + d * d * a.n * f;
  return r;
}
//   STATS (fprop): the batch-norm statistics of the output leave with it, as Welford triples (count, mean, M2) over the ROUNDED values as
//   they are stored: stats[channel][part][3], part = image (stats_per_wg == 0) or workgroup (every workgroup walks the same number of images) -
//   the layout bn_merge_channel takes from the implicit-GEMM kernels (equal counts), so the batch norm behind a narrow convolution launches
//   no statistics pass (5 launches of the ResNet step).  The sums are taken by the matrix cores, which idle here: a lane's packed output
//   row IS a B fragment (8 bf16 of column j = lane & 15), so  ones . Y  accumulates sum y per column and  Y^T . Y  accumulates sum y^2 on its
//   diagonal - exact products, f32 accumulation, no VALU work per value (as shifted VALU sums the statistics cost the B = 2048 step what the
//   five statistics launches had cost).  Unshifted sums over at most a few thousand values: M2 loses ~1e-6 (mean / std)^2 relative.
//   The epilogue must not cost the kernel a wave of occupancy: hence the second launch bound.  What it still costs (scripts/ncv_stats_probe.py,
//   stem / 6 -> 6 / 6 -> 16 layer at N = 2048: 11.3 / 5.5 / 4.8 us without, 14.2 / 7.4 / 6.3 us with): 1.3 - 2.2 us for the end of the workgroup
//   (barrier, the waves' sums through LDS, one more store round trip behind the last output store) and, on the stem, 1.5 us of matrix-core time.
//   PAR (round 6; input gradients of stride-2 convolutions, kh = 3, NK even): the staged image is zero-dilated, so an output row meets non-zero
//   image rows under ONE parity of the filter row r only.  The K pairs are packed by that parity (NcvW::nke = NK / 2) and a super-tile takes TR
//   rows of one parity (rows blk * 2 TR + 2 tr + par): it runs the NK / 2 k-steps of its class instead of all NK - the other half multiplied
//   structural zeros (EXPERIMENTS 66: the two such launches of the step spent 12 and 9 us in their k-loops).  Same products, same order per
//   output element as far as non-zero terms go; the skipped terms were exact zeros.
template <int NK, int SW, int PH0, int NS, bool ADD, bool STATS, bool PAR = false>
__global__ __launch_bounds__(256, (STATS && NK <= 5) ? 3 : 1) void ncv_fwd2_kernel(const bf16_t* __restrict__ src, const nv_bf8* __restrict__ wpk, const bf16_t* __restrict__ bias,
                                                       bf16_t* dst, NcvGeom q, const bf16_t* add, bf16_t* dst2, const bf16_t* __restrict__ bias2, int co_a,
                                                       float* __restrict__ stats, float* __restrict__ stats2, int stats_per_wg,
                                                       const bf16_t* __restrict__ src2) {
  // src2 (dgrad of a pair): the second source of the staged image, channels [q.C1, q.C)
  // co_a: output columns [0, co_a) belong to dst, [co_a, q.CO) to dst2 (the sibling 1x1 of NcvW; co_a = q.CO and dst2 = nullptr otherwise)
  __shared__ float sst[STATS ? 2 : 1][4][16][3];           // [image parity][wave][MFMA column]: the waves' triples of one image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* xs = reinterpret_cast<unsigned short*>(smem);
  NCV_STAMP_AT(0);
  constexpr int P = 8 / SW;
  constexpr int ND = P / NS;                               // accumulators (MFMAs per k-step) of a super-tile
  constexpr int NSEG = (PH0 + (ND - 1) * NS * SW + 7) < 16 ? 2 : 3;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nthreads = blockDim.x, nwaves = nthreads >> 6;
  const int co = NS == 2 ? (lane & 7) : (lane & 15);
  const int ncg = q.Wo / P, TR = 16 / ncg;
  // The first image's packets are requested before anything else: a workgroup lives for two or three images, and in-kernel stamps
  // (scripts/ncv_stamp_probe.py) showed 3800 - 6300 of its 11 - 20 k cycles between "weights in registers" and "first image in LDS" -
  // the HBM round trip of that first load, started only after the weights had arrived and the LDS was zeroed.
  const int64_t img_in = (int64_t)q.C1 * q.H * q.W, img_in2 = (int64_t)(q.C - q.C1) * q.H * q.W;
  NcvPre pre;
  const NcvPlan plan = ncv_stage_plan(q, tid, nthreads);
  if (q.pf && (int)blockIdx.x < q.N) ncv_stage_load(pre, plan, src + blockIdx.x * img_in, src2 + blockIdx.x * img_in2);
  nv_bf8 wfr[NK];
  int koff[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ks++) {
    wfr[ks] = wpk[ks * 64 + lane];
    int pair = ks * 4 + (lane >> 4);
    const int pairs1 = q.C1 * q.kh;
    if (pair >= pairs1 + (q.C - q.C1)) pair = 0;
    int c = (pair * q.kh_inv) >> 16, r = pair - c * q.kh;
    if (pair >= pairs1) { c = q.C1 + (pair - pairs1); r = q.kh >> 1; }   // second source: centre row only
    if constexpr (PAR) {                                  // the packed image's parity order (NcvW::nke = NK / 2, kh == 3)
      const int pr = ks * 4 + (lane >> 4);
      if (ks < NK / 2) { c = pr >> 1; r = 2 * (pr & 1); if (c >= q.C1) { c = 0; r = 0; } }
      else {
        const int p2 = pr - (NK / 2) * 4;
        if (p2 < q.C1) { c = p2; r = 1; }
        else if (p2 < q.C) { c = p2; r = 1; }            // second source: channels [C1, C) of the staged image, centre row
        else { c = 0; r = 0; }                            // (padded pairs: zero weights, any valid address)
      }
    }
    koff[ks] = (c * q.Hs + r) * q.Ws * 2;
  }
  const float bv = co < co_a ? (bias ? (float)bias[co] : 0.f) : ((bias2 && co < q.CO) ? (float)bias2[co - co_a] : 0.f);
#ifdef NCV_STAMP
  __builtin_amdgcn_s_waitcnt(0);
  NCV_STAMP_AT(1);
#endif
  const int HoWo = q.Ho * q.Wo, nsuper = q.Ho / TR;
  const int img_elems = q.C * q.Hs * q.Ws;
  for (int o = tid * 8; o < img_elems; o += nthreads * 8) *reinterpret_cast<uint4*>(xs + o) = make_uint4(0, 0, 0, 0);
  // A-side lane -> (row within the super-tile, column group)
  const int a_tr = (lane & 15) / ncg, a_cg = (lane & 15) - a_tr * ncg;
  const int a_off = ((PAR ? 2 : 1) * a_tr * q.sh * q.Ws + a_cg * 8 + (q.wx - PH0)) * 2;      // PAR: a tile's rows are two apart
  // the waves' sums of one part (left in sst[par] before the barrier the caller has just passed) -> stats[channel][part] = (count, mean, M2),
  // added in wave order; NS = 2: channel co sits in columns co and co + 8
  auto stats_flush = [&](int part, int nparts, int par) {
    if (wid == 0 && lane < (NS == 2 ? 8 : 16) && lane < q.CO) {
      float cnt = 0.f, sy = 0.f, sq = 0.f;
      for (int h = 0; h < NS; h++)
        for (int k = 0; k < nwaves; k++) { cnt += sst[par][k][lane + 8 * h][0]; sy += sst[par][k][lane + 8 * h][1]; sq += sst[par][k][lane + 8 * h][2]; }
      const float mean = sy / cnt;                         // (cnt > 0: every image has at least one super-tile)
      float* o = lane < co_a ? stats + ((int64_t)lane * nparts + part) * 3 : stats2 + ((int64_t)(lane - co_a) * nparts + part) * 3;
      o[0] = cnt; o[1] = mean; o[2] = fmaxf(sq - sy * mean, 0.f);
    }
  };
  // sum y (every row of st1 alike) and sum y^2 (diagonal of st2: row j of column j = lane & 15 is register j & 3 of lane group j >> 2) of this
  // wave's `tiles` super-tiles -> sst[par][wave][column] = (count, sum y, sum y^2)
  nv_f4 st1 = nv_f4{0.f, 0.f, 0.f, 0.f}, st2 = nv_f4{0.f, 0.f, 0.f, 0.f};
  int tiles = 0;
  auto stats_leave = [&](int par) {
    const int j = lane & 15, r = j & 3;
    const float sq = r == 0 ? st2[0] : r == 1 ? st2[1] : r == 2 ? st2[2] : st2[3];
    if ((lane >> 4) == (j >> 2)) { sst[par][wid][j][0] = (float)(tiles * (16 * P / NS)); sst[par][wid][j][1] = st1[0]; sst[par][wid][j][2] = sq; }
    st1 = nv_f4{0.f, 0.f, 0.f, 0.f}; st2 = nv_f4{0.f, 0.f, 0.f, 0.f}; tiles = 0;
  };
  const unsigned ones2 = 0x3f803f80u;                      // two bf16 ones
  typedef unsigned int nv_u4 __attribute__((ext_vector_type(4)));
  const nv_bf8 ones = __builtin_bit_cast(nv_bf8, nv_u4{ones2, ones2, ones2, ones2});
  auto stats_take = [&](unsigned a, unsigned b, unsigned c, unsigned d) {
    const nv_bf8 y = __builtin_bit_cast(nv_bf8, nv_u4{a, b, c, d});
    st1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, y, st1, 0, 0, 0);
    st2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, y, st2, 0, 0, 0);
  };
  int it = 0;
  for (int n = blockIdx.x; n < q.N; n += gridDim.x, it++) {
    __syncthreads();                                     // zero fill / the previous image's reads are done
    NCV_STAMP_ONCE(2);
    if (STATS && !stats_per_wg && it > 0) stats_flush(n - (int)gridDim.x, q.N, (it - 1) & 1);
#if defined(NCV_SKIP) && (NCV_SKIP & 4)
    if (n < 0)
#endif
    {
    if (q.pf) ncv_stage_store(xs, pre, plan, q.dil);
    else ncv_stage(xs, src + n * img_in, q, tid, nthreads);
    }
    __syncthreads();
    NCV_STAMP_ONCE(3);
#if defined(NCV_SKIP) && (NCV_SKIP & 4)
    if (n < 0)
#endif
    if (q.pf && n + (int)gridDim.x < q.N)                 // in flight during the MFMAs
      ncv_stage_load(pre, plan, src + (n + (int)gridDim.x) * img_in, src2 + (n + (int)gridDim.x) * img_in2);
    // this lane's output plane: channel co of dst, or channel co - co_a of the sibling's tensor
    bf16_t* yc = co < co_a ? dst + ((int64_t)n * co_a + co) * HoWo : dst2 + ((int64_t)n * (q.CO - co_a) + (co - co_a)) * HoWo;
    for (int st = wid; st < nsuper; st += nwaves) {
      // rows of the super-tile: h0 + RS * tr (PAR: super-tiles 2 b and 2 b + 1 share the 2 TR rows of block b, one parity each)
      constexpr int RS = PAR ? 2 : 1;
      const int h0 = PAR ? (st >> 1) * (2 * TR) + (st & 1) : st * TR;
      const char* base = smem + a_off + h0 * q.sh * q.Ws * 2;
      nv_f4 acc[ND];
#pragma unroll
      for (int d = 0; d < ND; d++) acc[d] = nv_f4{0.f, 0.f, 0.f, 0.f};
#ifdef NCV_SKIP                      // diagnostic builds only (scripts/build_variant.sh): 1 = no k-loop, 2 = no output stores, 4 = no image staging
      if (NCV_SKIP & 1) { acc[0][0] = (float)lane; }
      else
#endif
      {
      auto kstep = [&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;
        unsigned int sg[NSEG * 4 + 1];
#pragma unroll
        for (int e = 0; e < NSEG; e++) {
          const uint4 v = *reinterpret_cast<const uint4*>(base + koff[ks] + e * 16);
          sg[e * 4 + 0] = v.x; sg[e * 4 + 1] = v.y; sg[e * 4 + 2] = v.z; sg[e * 4 + 3] = v.w;
        }
        sg[NSEG * 4] = 0;
#pragma unroll
        for (int d = 0; d < ND; d++) {
          const int o = PH0 + d * NS * SW;                 // compile-time after unrolling: window origin of phase d * NS
          const int dq = o >> 1;
          unsigned int f[4];
#pragma unroll
          for (int j = 0; j < 4; j++) f[j] = (o & 1) ? __builtin_amdgcn_alignbit(sg[dq + j + 1], sg[dq + j], 16) : sg[dq + j];
          typedef unsigned int u4v __attribute__((ext_vector_type(4)));
          const u4v fv = {f[0], f[1], f[2], f[3]};
          acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(nv_bf8, fv), wfr[ks], acc[d], 0, 0, 0);
        }
      };
      if constexpr (PAR) {
        // rows h0 + 2 tr of the dilated image hold values iff (row + r - top) is even: the class of k-steps whose r has the parity of (h0 + top)
        if (((h0 + q.top) & 1) == 0) ncv_static_for<0, NK / 2>(kstep);          // r in {0, 2}
        else ncv_static_for<NK / 2, NK>(kstep);                                  // r = 1 (and the second source's centre row)
      } else ncv_static_for<0, NK>(kstep);
      }
      // add != nullptr: dst = round(round(conv) + add) (see ig_conv8d_kernel); the rows this lane stores.  Requested after the MFMAs:
      // held across them they cost 12 - 20 registers and a wave of occupancy
      constexpr int NROW = NS == 1 ? 4 : 2;
      const int row0 = (lane >> 4) * 4 + (NS == 1 ? 0 : ((lane >> 3) & 1) * 2);
      uint4 addv[NROW];
      if (ADD && co < q.CO) {
        const bf16_t* ap = add + (int64_t)n * q.CO * HoWo + co * HoWo;
#pragma unroll
        for (int k = 0; k < NROW; k++) {
          const int i = row0 + k, tr = i / ncg, cg = i - tr * ncg;
          const bf16_t* a = ap + (h0 + RS * tr) * q.Wo + cg * P;
          if (P == 8) addv[k] = *reinterpret_cast<const uint4*>(a);
          else { const uint2 t = *reinterpret_cast<const uint2*>(a); addv[k] = make_uint4(t.x, t.y, 0, 0); }
        }
      }
      if (STATS) tiles++;
      if (NS == 1) {
        unsigned int prev[P / 2];                          // P = 4: two rows make one fragment
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const int i = (lane >> 4) * 4 + rr;
          const int tr = i / ncg, cg = i - tr * ncg;
          bf16_t* o = yc + (h0 + RS * tr) * q.Wo + cg * P;
          unsigned int pk[P / 2];
#pragma unroll
          for (int d = 0; d < P; d += 2) {
            const bf16_t lo(acc[d % ND][rr] + bv), hi(acc[(d + 1) % ND][rr] + bv);
            pk[d >> 1] = (unsigned)lo.bits | ((unsigned)hi.bits << 16);
          }
          if (ADD) {
#pragma unroll
            for (int j = 0; j < P / 2; j++) pk[j] = add_bf16x2(pk[j], (&addv[rr % NROW].x)[j]);
          }
          if (STATS) {                                     // (every lane: the columns beyond CO carry zeros)
            if (P == 8) stats_take(pk[0], pk[1], pk[2 % (P / 2)], pk[3 % (P / 2)]);
            else if (rr & 1) stats_take(prev[0], prev[1], pk[0], pk[1]);
            else { prev[0] = pk[0]; prev[1] = pk[1]; }
          }
          if (co < q.CO NCV_STORE_COND) {
            if (P == 8) *reinterpret_cast<uint4*>(o) = make_uint4(pk[0], pk[1], pk[2 % (P / 2)], pk[3 % (P / 2)]);
            else *reinterpret_cast<uint2*>(o) = make_uint2(pk[0], pk[1]);
          }
        }
      } else {
        // lane (s, co, g) holds phases 2d + s of rows 4g .. 4g+3; its partner lane ^ 8 holds the other parity.  The s = 0 lane
        // completes rows 4g, 4g+1 and the s = 1 lane rows 4g+2, 4g+3: each sends the two rows it does not store.
        const int sft = (lane >> 3) & 1;
        unsigned int keep[2][ND], got[2][ND];
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
          for (int d = 0; d < ND; d++) {
            const bf16_t mine(acc[d][sft * 2 + h] + bv), theirs(acc[d][(1 - sft) * 2 + h] + bv);    // rows this lane stores / sends
            keep[h][d] = mine.bits;
            got[h][d] = (unsigned)__shfl_xor((int)theirs.bits, 8, 64);
          }
        unsigned int prev[ND];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int i = (lane >> 4) * 4 + sft * 2 + h;
          const int tr = i / ncg, cg = i - tr * ncg;
          bf16_t* o = yc + (h0 + RS * tr) * q.Wo + cg * P;
          unsigned int pk[ND];
#pragma unroll
          for (int d = 0; d < ND; d++) {
            const unsigned int even = sft ? got[h][d] : keep[h][d], odd = sft ? keep[h][d] : got[h][d];
            pk[d] = even | (odd << 16);
          }
          if (ADD) {
#pragma unroll
            for (int d = 0; d < ND; d++) pk[d] = add_bf16x2(pk[d], (&addv[h % NROW].x)[d]);
          }
          if (STATS) {
            if (P == 8) stats_take(pk[0], pk[1], pk[2 % ND], pk[3 % ND]);
            else if (h) stats_take(prev[0], prev[1], pk[0], pk[1]);
            else { prev[0] = pk[0]; prev[1] = pk[1]; }
          }
          if (co < q.CO NCV_STORE_COND) {
            if (P == 8) *reinterpret_cast<uint4*>(o) = make_uint4(pk[0], pk[1], pk[2 % ND], pk[3 % ND]);
            else *reinterpret_cast<uint2*>(o) = make_uint2(pk[0], pk[1]);
          }
        }
      }
    }
    if (STATS && !stats_per_wg) stats_leave(it & 1);
    NCV_STAMP_ONCE(4);
  }
  if (STATS && it > 0) {
    if (stats_per_wg) stats_leave(0);
    __syncthreads();
    if (stats_per_wg) stats_flush((int)blockIdx.x, (int)gridDim.x, 0);
    else stats_flush((int)blockIdx.x + (it - 1) * (int)gridDim.x, q.N, (it - 1) & 1);
  }
#ifdef NCV_STAMP
  NCV_STAMP_AT(5);
  __builtin_amdgcn_s_waitcnt(0);
  NCV_STAMP_AT(6);
#endif
}
#ifdef NCV_STAMP
extern "C" int lamp_debug_ncv_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(ncv_stamps), sizeof(ncv_stamps)) == hipSuccess ? 0 : 1; }
#endif

// ---- wgrad ---------------------------------------------------------------------------------------------
struct NcvWGeom {
  int N, Cin, Cout, H, W, Ho, Wo;
  int kh, kw, ph, pw;
  int Hs, Ws;            // SW == 1: LDS x image [Cin][Hs][Ws]; SW == 2: [Cin][Hs][2 parities][Ws]
  int ncol;              // Cin * kh * kw
  int IG;                // images staged per round
  // ncv_wgrad2_kernel, weight gradients of a PAIR (round 5): rows [Cout, Cout + Cout2) of the staged gradient image come from a second tensor,
  // the output gradient of a sibling 1x1 convolution of the same x (same stride and output map); of their products only the centre tap is
  // its weight gradient [Cout2][Cin] (the other taps are computed and dropped: the MFMA's 16 rows were 6 of 16 used).  0: none.
  int Cout2;
};
constexpr int NCV_OFF2 = 4;   // stride 2: column w lives at plane (w & 1), position (w >> 1) + NCV_OFF2

template <int NT, int SW>
__global__ __launch_bounds__(256) void ncv_wgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ partial, NcvWGeom q,
                                                        int images_per_block) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int HoWo = q.Ho * q.Wo;
  const int ximg = q.Cin * q.Hs * q.Ws * (SW == 2 ? 2 : 1);        // elements per staged image
  unsigned short* xs = reinterpret_cast<unsigned short*>(smem);      // [IG][ximg]
  unsigned short* ds = xs + q.IG * ximg;                             // [IG][16][HoWo]  (rows >= Cout stay zero)
  float* red = reinterpret_cast<float*>(ds + q.IG * 16 * HoWo);      // [4 waves][16][NT * 16]
  {
    const int tot = q.IG * (ximg + 16 * HoWo);                       // multiple of 8 (host guarantees)
    for (int o = tid * 8; o < tot; o += 256 * 8) *reinterpret_cast<uint4*>(xs + o) = make_uint4(0, 0, 0, 0);
  }
  // per-lane column (ci, r, s) of each n-tile -> byte offset of its shifted row inside a staged image
  int coff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) {
    int cidx = nt * 16 + (lane & 15);
    if (cidx >= q.ncol) cidx = 0;                                     // padded columns: computed, never written
    const int s = cidx % q.kw, r = (cidx / q.kw) % q.kh, ci = cidx / (q.kw * q.kh);
    if (SW == 1) coff[nt] = ((ci * q.Hs + r) * q.Ws + s - q.pw + NCV_LEFT) * 2;
    else {
      const int t = s - q.pw, par = t & 1, half = (t - par) >> 1;
      coff[nt] = (((ci * q.Hs + r) * 2 + par) * q.Ws + half + NCV_OFF2) * 2;
    }
  }
  nv_f4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) acc[nt] = nv_f4{0.f, 0.f, 0.f, 0.f};

  const int64_t n0 = (int64_t)blockIdx.x * images_per_block, n1 = min<int64_t>(n0 + images_per_block, q.N);
  const int chunks_per_img = HoWo >> 5;
  for (int64_t nb = n0; nb < n1; nb += q.IG) {
    const int ig = (int)min<int64_t>(q.IG, n1 - nb);
    __syncthreads();
    for (int im = 0; im < ig; im++) {
      const bf16_t* xp = x + (nb + im) * q.Cin * q.H * q.W;
      unsigned short* xi = xs + im * ximg;
      const int rc = q.W >> 3, total = q.Cin * q.H * rc;
      for (int i = tid; i < total; i += 256) {
        const int b = i % rc, a = (i / rc) % q.H, c = i / (rc * q.H);
        const uint4 v = *reinterpret_cast<const uint4*>(xp + (c * q.H + a) * q.W + b * 8);
        if (SW == 1) {
          *reinterpret_cast<uint4*>(xi + (c * q.Hs + q.ph + a) * q.Ws + NCV_LEFT + b * 8) = v;
        } else {
          // de-interleave even / odd columns
          uint2 ev, od;
          ev.x = (v.x & 0xffffu) | (v.y << 16); ev.y = (v.z & 0xffffu) | (v.w << 16);
          od.x = (v.x >> 16) | (v.y & 0xffff0000u); od.y = (v.z >> 16) | (v.w & 0xffff0000u);
          unsigned short* row = xi + ((c * q.Hs + q.ph + a) * 2) * q.Ws + NCV_OFF2 + b * 4;
          *reinterpret_cast<uint2*>(row) = ev;
          *reinterpret_cast<uint2*>(row + q.Ws) = od;
        }
      }
      const bf16_t* dp = dy + (nb + im) * q.Cout * HoWo;
      unsigned short* di = ds + im * 16 * HoWo;
      const int dtot = q.Cout * HoWo >> 3;
      for (int i = tid; i < dtot; i += 256) *reinterpret_cast<uint4*>(di + i * 8) = *reinterpret_cast<const uint4*>(dp + i * 8);
    }
    __syncthreads();
    const int nchunks = ig * chunks_per_img;
    for (int ch = wid; ch < nchunks; ch += 4) {
      const int pg = ch * 32 + (lane >> 4) * 8;                      // this lane's 8 consecutive output pixels
      const int im = pg / HoWo, pp = pg - im * HoWo;
      const int ho = pp / q.Wo, wo0 = pp - ho * q.Wo;
      const nv_s8 a = *reinterpret_cast<const nv_s8*>(ds + (im * 16 + (lane & 15)) * HoWo + pp);
      const char* xb = reinterpret_cast<const char*>(xs + im * ximg) +
                       (SW == 1 ? (ho * q.Ws + wo0) * 2 : ((ho * 2) * 2 * q.Ws + wo0) * 2);
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
        const nv_s8 b = *reinterpret_cast<const nv_s8_u*>(xb + coff[nt]);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(nv_bf8, a), __builtin_bit_cast(nv_bf8, b), acc[nt], 0, 0, 0);
      }
    }
  }
  // D: column = (ci, r, s) index, rows = 4 output channels; combine the 4 waves in a fixed order
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int rr = 0; rr < 4; rr++) red[(wid * 16 + (lane >> 4) * 4 + rr) * (NT * 16) + nt * 16 + (lane & 15)] = acc[nt][rr];
  __syncthreads();
  const int O = q.Cout * q.ncol;
  for (int i = tid; i < O; i += 256) {
    const int co = i / q.ncol, cidx = i - co * q.ncol;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; w++) a += red[(w * 16 + co) * (NT * 16) + cidx];
    partial[(int64_t)blockIdx.x * O + i] = a;
  }
}

// Aligned-read wgrad.  The MFMA columns of one tile are 16 (ci, r) pairs at ONE filter column s, so the window shift s - pw is
// the same for the whole wave: per (32-pixel chunk, pair tile) a lane reads two or three ALIGNED 16-byte segments of its image
// row and cuts the KW shifted fragments out of them with compile-time funnel shifts (an unaligned ds_read_b128 costs 11 aligned
// ones, see ncv_fwd2_kernel).  Stride 2: even / odd column planes, filter column s reads plane (s - pw) & 1 at offset
// (s - pw) >> 1 in {-1, 0} (KW <= 3).  acc[pair tile][s] holds dW[co][(ci, r)][s].
constexpr int NCV_OFFA = 8;   // column origin of the staged rows (both strides): 16-byte aligned segments
constexpr int NCV_WPF = 6;    // 16-byte packets per thread the image-group prefetch holds
template <int KW, int PW, int SW, int NPT>
__global__ __launch_bounds__(256) void ncv_wgrad2_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ partial, NcvWGeom q,
                                                         int images_per_block, const bf16_t* __restrict__ dy2, float* __restrict__ partial2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int HoWo = q.Ho * q.Wo;
  const int ximg = q.Cin * q.Hs * q.Ws * (SW == 2 ? 2 : 1);
  unsigned short* xs = reinterpret_cast<unsigned short*>(smem);
  unsigned short* ds = xs + q.IG * ximg;
  constexpr int NCOLT = NPT * KW * 16;                               // accumulator columns per output channel
  float* red = reinterpret_cast<float*>(ds + q.IG * 16 * HoWo);      // [4 waves][16][NCOLT]
  {
    const int tot = q.IG * (ximg + 16 * HoWo);
    for (int o = tid * 8; o < tot; o += 256 * 8) *reinterpret_cast<uint4*>(xs + o) = make_uint4(0, 0, 0, 0);
  }
  // per-lane row offset (bytes) of pair tile pt: pair = pt*16 + (lane & 15) -> (ci, r)
  const int npairs = q.Cin * q.kh;
  int poff[NPT];
#pragma unroll
  for (int pt = 0; pt < NPT; pt++) {
    int pair = pt * 16 + (lane & 15);
    if (pair >= npairs) pair = 0;                                    // padded columns: computed, never written
    const int ci = pair / q.kh, r = pair - ci * q.kh;
    poff[pt] = (SW == 1 ? (ci * q.Hs + r) * q.Ws : (ci * q.Hs + r) * 2 * q.Ws) * 2;
  }
  nv_f4 acc[NPT][KW];
#pragma unroll
  for (int pt = 0; pt < NPT; pt++)
#pragma unroll
    for (int s = 0; s < KW; s++) acc[pt][s] = nv_f4{0.f, 0.f, 0.f, 0.f};

  const int64_t n0 = (int64_t)blockIdx.x * images_per_block, n1 = min<int64_t>(n0 + images_per_block, q.N);
  const int chunks_per_img = HoWo >> 5;
  // Staging in two halves (as in ncv_fwd2_kernel): the x and dy packets of the NEXT image group are loaded into registers while this
  // group is multiplied, and written to LDS after the barrier.  A group is [image][x packets | dy packets] of 16 bytes.
  const int xpk = (q.Cin * q.H * q.W) >> 3, dpk = (q.Cout * HoWo) >> 3, dpk2 = (q.Cout2 * HoWo) >> 3, per = xpk + dpk + dpk2;
  const int rc = q.W >> 3;
  // a thread handles the same packets of every group: source offsets (elements, from the group's first image) and LDS destinations
  // (elements from xs) are computed once; meta = image index within the group, or -1 (no packet)
  const bool pf = q.IG * per <= NCV_WPF * 256;
  uint4 pre[NCV_WPF];
  int p_src[NCV_WPF], p_dst[NCV_WPF], p_im[NCV_WPF];
#pragma unroll
  for (int k = 0; k < NCV_WPF; k++) {
    const int idx = tid + k * 256;
    const int im = idx / per, j = idx - im * per;
    p_im[k] = (pf && im < q.IG) ? im : -1;
    if (j < xpk) {
      const int b = j % rc, a = (j / rc) % q.H, c = j / (rc * q.H);
      p_src[k] = im * q.Cin * q.H * q.W + j * 8;
      p_dst[k] = im * ximg + (SW == 1 ? (c * q.Hs + q.ph + a) * q.Ws + NCV_OFFA + b * 8 : ((c * q.Hs + q.ph + a) * 2) * q.Ws + NCV_OFFA + b * 4);
    } else if (j < xpk + dpk) {
      p_src[k] = -(im * q.Cout * HoWo + (j - xpk) * 8) - 1;          // negative: a dy packet
      p_dst[k] = q.IG * ximg + im * 16 * HoWo + (j - xpk) * 8;
    } else {
      p_src[k] = -(im * q.Cout2 * HoWo + (j - xpk - dpk) * 8) - 1 - (1 << 30);   // below -2^30: a packet of the second gradient tensor
      p_dst[k] = q.IG * ximg + im * 16 * HoWo + q.Cout * HoWo + (j - xpk - dpk) * 8;
    }
  }
  auto pre_load = [&](int64_t nb, int ig) {
    const bf16_t* xg = x + nb * q.Cin * q.H * q.W;
    const bf16_t* dg = dy + nb * q.Cout * HoWo;
    const bf16_t* dg2 = dy2 + nb * q.Cout2 * HoWo;
#pragma unroll
    for (int k = 0; k < NCV_WPF; k++)
      if (p_im[k] >= 0 && p_im[k] < ig) {
        const bf16_t* sp = p_src[k] >= 0 ? xg + p_src[k] : p_src[k] >= -(1 << 30) ? dg + (-p_src[k] - 1) : dg2 + (-p_src[k] - 1 - (1 << 30));
        pre[k] = *reinterpret_cast<const uint4*>(sp);
      }
  };
  auto stage_x = [&](unsigned short* dst, const uint4 v) {
    if (SW == 1) {
      *reinterpret_cast<uint4*>(dst) = v;
    } else {
      uint2 ev, od;
      ev.x = (v.x & 0xffffu) | (v.y << 16); ev.y = (v.z & 0xffffu) | (v.w << 16);
      od.x = (v.x >> 16) | (v.y & 0xffff0000u); od.y = (v.z >> 16) | (v.w & 0xffff0000u);
      *reinterpret_cast<uint2*>(dst) = ev;
      *reinterpret_cast<uint2*>(dst + q.Ws) = od;
    }
  };
  auto pre_store = [&](int ig) {
#pragma unroll
    for (int k = 0; k < NCV_WPF; k++) {
      if (p_im[k] < 0 || p_im[k] >= ig) continue;
      const uint4 v = pre[k];
      if (SW == 1 || p_src[k] < 0) *reinterpret_cast<uint4*>(xs + p_dst[k]) = v;      // stride 1 rows and dy packets: one 16-byte write
      else stage_x(xs + p_dst[k], v);
    }
  };
  auto x_dst = [&](unsigned short* xi, int j) {
    const int b = j % rc, a = (j / rc) % q.H, c = j / (rc * q.H);
    return xi + (SW == 1 ? (c * q.Hs + q.ph + a) * q.Ws + NCV_OFFA + b * 8 : ((c * q.Hs + q.ph + a) * 2) * q.Ws + NCV_OFFA + b * 4);
  };

  if (pf && n0 < n1) pre_load(n0, (int)min<int64_t>(q.IG, n1 - n0));
  for (int64_t nb = n0; nb < n1; nb += q.IG) {
    const int ig = (int)min<int64_t>(q.IG, n1 - nb);
    __syncthreads();
#if defined(NCV_WSKIP) && (NCV_WSKIP & 4)   // diagnostic builds only: 1 = no k-loop, 4 = no staging (requests and LDS writes)
    if (nb < 0)
#endif
    if (pf) pre_store(ig);
    else for (int im = 0; im < ig; im++) {
      const bf16_t* xp = x + (nb + im) * q.Cin * q.H * q.W;
      unsigned short* xi = xs + im * ximg;
      for (int i = tid; i < xpk; i += 256) stage_x(x_dst(xi, i), *reinterpret_cast<const uint4*>(xp + i * 8));
      const bf16_t* dp = dy + (nb + im) * q.Cout * HoWo;
      unsigned short* di = ds + im * 16 * HoWo;
      for (int i = tid; i < dpk; i += 256) *reinterpret_cast<uint4*>(di + i * 8) = *reinterpret_cast<const uint4*>(dp + i * 8);
      const bf16_t* dp2 = dy2 + (nb + im) * q.Cout2 * HoWo;
      for (int i = tid; i < dpk2; i += 256) *reinterpret_cast<uint4*>(di + q.Cout * HoWo + i * 8) = *reinterpret_cast<const uint4*>(dp2 + i * 8);
    }
    __syncthreads();
#if defined(NCV_WSKIP) && (NCV_WSKIP & 4)
    if (nb < 0)
#endif
    if (pf && nb + q.IG < n1) pre_load(nb + q.IG, (int)min<int64_t>(q.IG, n1 - nb - q.IG));      // in flight during the MFMAs
    const int nchunks = ig * chunks_per_img;
#if defined(NCV_WSKIP) && (NCV_WSKIP & 1)
    if (nb < 0)
#endif
    for (int ch = wid; ch < nchunks; ch += 4) {
      const int pg = ch * 32 + (lane >> 4) * 8;
      const int im = pg / HoWo, pp = pg - im * HoWo;
      const int ho = pp / q.Wo, wo0 = pp - ho * q.Wo;
      const nv_s8 a = *reinterpret_cast<const nv_s8*>(ds + (im * 16 + (lane & 15)) * HoWo + pp);
      const nv_bf8 af = __builtin_bit_cast(nv_bf8, a);
      // byte address of element column (wo0 + NCV_OFFA - 8) of this lane's group row: segment -1
      const char* xb = reinterpret_cast<const char*>(xs + im * ximg) +
                       (SW == 1 ? (ho * q.Ws + wo0 + NCV_OFFA - 8) * 2 : ((ho * 2) * 2 * q.Ws + wo0 + NCV_OFFA - 8) * 2);
#pragma unroll
      for (int pt = 0; pt < NPT; pt++) {
        const char* rowp = xb + poff[pt];
        if (SW == 1) {
          // 24 elements around the aligned window: [seg-1 | seg0 | seg+1]
          constexpr bool need_m = PW > 0, need_p = (KW - 1 - PW) > 0;
          unsigned int sg[13];
          const uint4 z = make_uint4(0, 0, 0, 0);
          const uint4 v0 = need_m ? *reinterpret_cast<const uint4*>(rowp) : z;
          const uint4 v1 = *reinterpret_cast<const uint4*>(rowp + 16);
          const uint4 v2 = need_p ? *reinterpret_cast<const uint4*>(rowp + 32) : z;
          sg[0] = v0.x; sg[1] = v0.y; sg[2] = v0.z; sg[3] = v0.w; sg[4] = v1.x; sg[5] = v1.y; sg[6] = v1.z; sg[7] = v1.w;
          sg[8] = v2.x; sg[9] = v2.y; sg[10] = v2.z; sg[11] = v2.w; sg[12] = 0;
#pragma unroll
          for (int s = 0; s < KW; s++) {
            const int o = 8 + s - PW;                              // compile-time element offset of the window
            const int dq = o >> 1;
            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
            u4v f;
#pragma unroll
            for (int j = 0; j < 4; j++) f[j] = (o & 1) ? __builtin_amdgcn_alignbit(sg[dq + j + 1], sg[dq + j], 16) : sg[dq + j];
            acc[pt][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(nv_bf8, f), acc[pt][s], 0, 0, 0);
          }
        } else {
          // per parity plane: [seg-1 | seg0]; filter column s -> plane (s - PW) & 1, element offset 8 + ((s - PW) >> 1)
          unsigned int sg[2][9];
#pragma unroll
          for (int par = 0; par < 2; par++) {
            const uint4 v0 = *reinterpret_cast<const uint4*>(rowp + par * q.Ws * 2);
            const uint4 v1 = *reinterpret_cast<const uint4*>(rowp + par * q.Ws * 2 + 16);
            sg[par][0] = v0.x; sg[par][1] = v0.y; sg[par][2] = v0.z; sg[par][3] = v0.w;
            sg[par][4] = v1.x; sg[par][5] = v1.y; sg[par][6] = v1.z; sg[par][7] = v1.w; sg[par][8] = 0;
          }
#pragma unroll
          for (int s = 0; s < KW; s++) {
            const int t = s - PW, par = t & 1, half = (t - par) >> 1;   // compile time
            const int o = 8 + half, dq = o >> 1;
            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
            u4v f;
#pragma unroll
            for (int j = 0; j < 4; j++) f[j] = (o & 1) ? __builtin_amdgcn_alignbit(sg[par][dq + j + 1], sg[par][dq + j], 16) : sg[par][dq + j];
            acc[pt][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(nv_bf8, f), acc[pt][s], 0, 0, 0);
          }
        }
      }
    }
  }
  // D: column = pair within the tile, rows = 4 output channels; combine the 4 waves in a fixed order
#pragma unroll
  for (int pt = 0; pt < NPT; pt++)
#pragma unroll
    for (int s = 0; s < KW; s++)
#pragma unroll
      for (int rr = 0; rr < 4; rr++) red[(wid * 16 + (lane >> 4) * 4 + rr) * NCOLT + (pt * KW + s) * 16 + (lane & 15)] = acc[pt][s][rr];
  __syncthreads();
  const int O = q.Cout * q.ncol;
  for (int i = tid; i < O; i += 256) {
    const int co = i / q.ncol, cidx = i - co * q.ncol;              // cidx = (ci*kh + r)*kw + s
    const int pair = cidx / KW, s = cidx - pair * KW;
    const int col = ((pair >> 4) * KW + s) * 16 + (pair & 15);
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; w++) a += red[(w * 16 + co) * NCOLT + col];
    partial[(int64_t)blockIdx.x * O + i] = a;
  }
  // the sibling 1x1's weight gradient [Cout2][Cin]: the centre tap of rows Cout ...
  const int O2 = q.Cout2 * q.Cin;
  for (int i = tid; i < O2; i += 256) {
    const int co2 = i / q.Cin, ci = i - co2 * q.Cin;
    const int pair = ci * q.kh + (q.kh >> 1), s = KW >> 1;
    const int col = ((pair >> 4) * KW + s) * 16 + (pair & 15);
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; w++) a += red[(w * 16 + q.Cout + co2) * NCOLT + col];
    partial2[(int64_t)blockIdx.x * O2 + i] = a;
  }
}

// (the reduction of the per-block partial sums lives in wgrad_reduce.h: it runs batched with the other layers' reductions)

// ---- host ------------------------------------------------------------------------------------------------
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

static bool ncv_common(const ConvGeom& g, int dtype) {
  if (dtype != kBF16) return false;
  if (g.groups != 1 || g.transposed || g.dh != 1 || g.dw != 1) return false;
  if (g.Cin > 16 || g.Cout > 16 || g.N < 1) return false;
  if (g.sh != g.sw || g.sh < 1 || g.sh > 2) return false;
  if (g.kw > 8 || g.ph > g.kh - 1 || g.pw > g.kw - 1 || g.pw > 8) return false;
  return true;
}

template <int NK>
static void ncv_launch(const bf16_t* src, const nv_bf8* wpk, const bf16_t* bias, bf16_t* dst, const NcvGeom& q, int blocks, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((ncv_fwd_kernel<NK>), dim3(blocks), dim3(256), lds, st, src, wpk, bias, dst, q);
}

// ---- packed weight-fragment images, cached per (storage uid, view, geometry, direction, stream) and storage version: the same scheme
// as the implicit-GEMM images (conv_igemm.hip, "packed-weight cache"); LAMP_PACK_CACHE=0 disables it.
namespace {
struct NcvPackKey {
  uint64_t uid; int64_t offset; int Cout, Cin, kh, kw, dgrad, ns, sw; hipStream_t st;
  uint64_t uid2; int64_t offset2; int Cout2;     // the sibling 1x1 filter packed into the same image (0 / 0 / 0: none)
  bool operator<(const NcvPackKey& o) const {
    return std::tie(uid, offset, Cout, Cin, kh, kw, dgrad, ns, sw, st, uid2, offset2, Cout2) <
           std::tie(o.uid, o.offset, o.Cout, o.Cin, o.kh, o.kw, o.dgrad, o.ns, o.sw, o.st, o.uid2, o.offset2, o.Cout2);
  }
};
// version2: the sibling's storage version.  No data pointers are kept: the re-pack hook resolves both filters among the parameters it is handed
struct NcvPackVal { uint64_t version; Tensor* packed; uint64_t tick; bool pinned = false; uint64_t version2 = 0; };
std::mutex g_ncv_mu;
std::map<NcvPackKey, NcvPackVal> g_ncv_cache;
uint64_t g_ncv_tick = 0;
constexpr int64_t NCV_PACK_ELEMS = (int64_t)NCV_NKMAX * 64 * 8;
}  // namespace

static void ncv_pack_launch(const NcvPackMany& a, int cnt, hipStream_t st) {
  hipLaunchKernelGGL(ncv_pack_kernel, dim3((NCV_NKMAX * 64 + 255) / 256, (unsigned)cnt), dim3(256), 0, st, a);
  LAMP_LAUNCH_CHECK();
}

// returns a +1 handle on the fragment image of `w` for this direction
static Tensor* ncv_packed_weights(const Tensor* w, const NcvW& wq, hipStream_t st, const Tensor* w2 = nullptr) {
  static const bool cache_on = [] { const char* e = getenv("LAMP_PACK_CACHE"); return !(e && e[0] == '0'); }();
  const bool cacheable = cache_on && w->st->owned && !w->st->scratch && (!w2 || (w2->st->owned && !w2->st->scratch));
  const NcvPackKey key{w->st->uid, w->offset, wq.Cout, wq.Cin, wq.kh, wq.kw, wq.dgrad, wq.ns + 16 * wq.nke, wq.sw, st,
                       w2 ? w2->st->uid : 0, w2 ? w2->offset : 0, w2 ? wq.Cout2 : 0};
  const uint64_t ver = w->st->version.load(std::memory_order_relaxed);
  const uint64_t ver2 = w2 ? w2->st->version.load(std::memory_order_relaxed) : 0;
  if (cacheable) {
    std::lock_guard<std::mutex> lk(g_ncv_mu);
    auto it = g_ncv_cache.find(key);
    if (it != g_ncv_cache.end() && it->second.version == ver && it->second.version2 == ver2) {
      it->second.tick = ++g_ncv_tick;
      if (allocator_capturing()) it->second.pinned = true;       // a graph being captured records this address: never evict the entry
      return retain(it->second.packed);
    }
  }
  int64_t ps[1] = {NCV_PACK_ELEMS};
  Hold wp(new_tensor(ps, 1, kBF16, w->device()));
  NcvPackMany a;
  a.w[0] = wq; a.dst[0] = reinterpret_cast<nv_bf8*>(wp->ptr<bf16_t>());
  ncv_pack_launch(a, 1, st);
  if (cacheable) {
    std::lock_guard<std::mutex> lk(g_ncv_mu);
    auto it = g_ncv_cache.find(key);
    if (it != g_ncv_cache.end()) { release(it->second.packed); g_ncv_cache.erase(it); }
    if (g_ncv_cache.size() >= 256) {           // evict the least recently used entry that no captured graph reads
      auto victim = g_ncv_cache.end();
      for (auto i = g_ncv_cache.begin(); i != g_ncv_cache.end(); ++i)
        if (!i->second.pinned && (victim == g_ncv_cache.end() || i->second.tick < victim->second.tick)) victim = i;
      if (victim != g_ncv_cache.end()) { release(victim->second.packed); g_ncv_cache.erase(victim); }
    }
    g_ncv_cache[key] = NcvPackVal{ver, retain(wp.get()), ++g_ncv_tick, allocator_capturing(), ver2};
  }
  return wp.take();
}

// Called by the optimisers right after they have written the parameters (next to igemm_repack_cached): every cached fragment image of
// these parameters on this stream is packed again IN PLACE (a captured HIP graph keeps the address), all of them in one launch.
// `fill` (optional): instead of launching, hand the LAST batch of at most NCV_PACK_MAX images to the caller (*fill, *fill_cnt) - the optimiser's hook
// in conv_igemm.hip packs them in the launch that packs the implicit-GEMM images; earlier full batches are still launched here
void narrow_repack_cached(lamp_tensor* const* params, int n, hipStream_t st, NcvPackMany* fill, int* fill_cnt) {
  if (fill_cnt) *fill_cnt = 0;
  static const bool on = [] { const char* e = getenv("LAMP_PACK_AFTER_STEP"); return !(e && e[0] == '0'); }();
  if (!on) return;
  std::lock_guard<std::mutex> lk(g_ncv_mu);
  if (g_ncv_cache.empty()) return;
  // the narrow filters among the parameters just written: an image is packed again ONLY from tensors the caller holds right now (ADVICE r5:
  // the entry used to keep raw data pointers of both filters of a pair and read the absent one's after its storage could be gone)
  std::vector<const Tensor*> cand;
  for (int i = 0; i < n; i++) {
    const Tensor* w = params[i];
    if (!w || !w->is_device() || w->dtype != kBF16 || w->ndim != 4 || !w->st->owned || !w->is_contiguous()) continue;
    if (w->sizes[0] > 16 || w->sizes[1] > 16) continue;
    cand.push_back(w);
  }
  if (cand.empty()) return;
  auto find = [&](uint64_t uid, int64_t offset, int co, int ci, int kh, int kw) -> const Tensor* {
    for (const Tensor* w : cand)
      if (w->st->uid == uid && w->offset == offset && (int)w->sizes[0] == co && (int)w->sizes[1] == ci && (int)w->sizes[2] == kh && (int)w->sizes[3] == kw)
        return w;
    return nullptr;
  };
  NcvPackMany a;
  int cnt = 0;
  auto flush = [&] {
    if (cnt == 0) return;
    ncv_pack_launch(a, cnt, st);
    cnt = 0;
  };
  for (auto it = g_ncv_cache.begin(); it != g_ncv_cache.end();) {
    const NcvPackKey& k = it->first;
    NcvPackVal& v = it->second;
    if (k.st != st) { ++it; continue; }
    const Tensor* w1 = find(k.uid, k.offset, k.Cout, k.Cin, k.kh, k.kw);
    const Tensor* w2 = k.Cout2 > 0 ? find(k.uid2, k.offset2, k.Cout2, k.Cin, 1, 1) : nullptr;
    if (!w1 && !w2) { ++it; continue; }                          // none of this call's parameters
    if (!w1 || (k.Cout2 > 0 && !w2)) {
      // a pair image with only ONE of its filters among the parameters (the other was a temporary, or its layer was rebuilt): it cannot be
      // brought up to date here.  Drop it - the next convolution packs a fresh image from the tensors it is given - unless a captured graph
      // reads this address; that entry stays, stale by its versions, exactly as when a parameter is written outside the optimiser.
      if (!v.pinned) { release(v.packed); it = g_ncv_cache.erase(it); }
      else ++it;
      continue;
    }
    if (cnt == NCV_PACK_MAX) flush();
    a.w[cnt] = NcvW{w1->ptr<bf16_t>(), k.Cout, k.Cin, k.kh, k.kw, k.dgrad, k.ns & 15, k.sw, w2 ? w2->ptr<bf16_t>() : nullptr, k.Cout2, k.ns >> 4};   // (key: ns + 16 nke)
    a.dst[cnt] = static_cast<nv_bf8*>(v.packed->raw());
    v.version = w1->st->version.load(std::memory_order_relaxed);
    v.version2 = w2 ? w2->st->version.load(std::memory_order_relaxed) : 0;
    v.tick = ++g_ncv_tick;
    cnt++;
    ++it;
  }
  if (fill && cnt > 0) { *fill = a; *fill_cnt = cnt; }
  else flush();
}
void narrow_pack_launch(const NcvPackMany& a, int cnt, hipStream_t st) { ncv_pack_launch(a, cnt, st); }

// sibling (fprop, optional): a 1x1 convolution of the same input, same stride and output map, whose Cout2 channels fit beside w's in the
// MFMA's 16 columns: both outputs from one launch of the aligned kernel (false, nothing launched, when that kernel does not take the pair)
struct NcvSibling { const Tensor* w; const Tensor* bias; Tensor* out; };
// second (dgrad, optional): the output gradient and the filter of a sibling 1x1 convolution of the same input (same stride and output map): both
// input gradients, summed in f32, from one launch of the aligned kernel (false, nothing launched, when that kernel does not take the pair)
struct NcvSecondGrad { const Tensor* dy; const Tensor* w; };
static bool ncv_run(const Tensor* in, const Tensor* w, const Tensor* bias, Tensor* out, const ConvGeom& g, bool dgrad, hipStream_t st,
                    const Tensor* addend = nullptr, bool* addend_fused = nullptr, const NcvSibling* sib = nullptr,
                    const NcvSecondGrad* second = nullptr) {
  if (addend_fused) *addend_fused = false;
  if (!ncv_common(g, in->dtype)) return false;
  const int cout2 = sib ? (int)sib->w->sizes[0] : second ? (int)second->w->sizes[0] : 0;
  if (sib && (dgrad || g.Cout + cout2 > 16)) return false;
  if (second && (!dgrad || sib || cout2 > 16)) return false;
  NcvGeom q;
  q.N = (int)g.N; q.kh = g.kh; q.pf = 0; q.kh_inv = 65536 / g.kh + 1;
  if (!dgrad) {
    if (g.W % 8 != 0) return false;
    q.C = (int)g.Cin; q.C1 = q.C; q.CO = (int)g.Cout + cout2; q.H = (int)g.H; q.W = (int)g.W; q.dil = 1;
    q.top = g.ph; q.left = NCV_LEFT;
    q.Ho = (int)g.Ho; q.Wo = (int)g.Wo; q.sh = g.sh; q.sw = g.sw; q.wx = NCV_LEFT - g.pw;
    q.Hs = std::max((int)g.H + 2 * g.ph, (q.Ho - 1) * g.sh + g.kh);
    q.Ws = round_up(std::max(NCV_LEFT + (int)g.W, (q.Wo - 1) * g.sw + q.wx + 8), 8);
  } else {
    // dx = stride-1 correlation of the dilated dy image with the mirrored filter, padding k-1-p
    const int pt = g.kh - 1 - g.ph, pl = g.kw - 1 - g.pw;
    q.C1 = (int)g.Cout; q.C = q.C1 + (second ? cout2 : 0); q.CO = (int)g.Cin; q.H = (int)g.Ho; q.W = (int)g.Wo; q.dil = g.sh;
    if (q.dil == 1 && q.W % 8 != 0) return false;
    q.top = pt; q.left = NCV_LEFT;
    q.Ho = (int)g.H; q.Wo = (int)g.W; q.sh = 1; q.sw = 1; q.wx = NCV_LEFT - pl;
    q.Hs = std::max(pt + ((int)g.Ho - 1) * g.sh + 1, (int)g.H + g.kh - 1);
    q.Ws = round_up(std::max(NCV_LEFT + ((int)g.Wo - 1) * g.sw + 1, (int)g.W - 1 + q.wx + 8), 8);
  }
  if ((q.Ho * q.Wo) % 4 != 0) return false;
  // aligned-window kernel: Wo a multiple of the P = 8/stride phases, 16 pixels = TR full-phase rows
  const int P = 8 / q.sw, ph0 = q.wx & 7;
  bool aligned = q.Wo % P == 0 && (ph0 == 0 || ph0 == 6 || ph0 == 7);
  int ncg = aligned ? q.Wo / P : 0;
  if (aligned && !(ncg == 2 || ncg == 4 || ncg == 8 || ncg == 16)) aligned = false;
  if (aligned && q.Ho % (16 / ncg) != 0) aligned = false;
  if (aligned) q.Ws = round_up(std::max(q.Ws, q.Wo * q.sw + (q.wx - ph0) + 16), 8);
  if ((sib || second) && !aligned) return false;
  const int pairs = q.C1 * q.kh + (q.C - q.C1);
  const int nk_real = (pairs + 3) / 4;
  static const int nk_opts[] = {1, 2, 4, 5, 6, 8, 12, 16};
  int NK = 0;
  for (int o : nk_opts) if (o >= nk_real) { NK = o; break; }
  if (!NK) return false;
  // round 6: the input gradient of a stride-2 pair runs its super-tiles by row parity over K pairs packed by the parity of the filter row
  // (ncv_fwd2_kernel<.., PAR>; LAMP_NCV_DGRAD_PARITY=0: the plain form): half the k-steps per super-tile
  int nke = 0;
  static const bool par_on = [] { const char* e = getenv("LAMP_NCV_DGRAD_PARITY"); return !(e && e[0] == '0'); }();
  if (par_on && aligned && dgrad && second && q.dil == 2 && g.kh == 3 && NK > 5) {
    const int nk2 = NK <= 6 ? 6 : NK <= 8 ? 8 : 16;
    const int need_e = (q.C1 * 2 + 3) / 4, need_o = (q.C + 3) / 4;      // pairs (c, r even) / (c, 1) and the second source's centre rows
    if (need_e <= nk2 / 2 && need_o <= nk2 / 2 && q.Ho % (2 * (16 / ncg)) == 0) nke = nk2 / 2;
  }
  // The row pitch of the LDS image decides the bank conflicts of the fragment reads: the 16 lanes of a quarter-wave read 16 bytes each at
  // (row a_tr, column group a_cg) = a_tr * RS * sh * pitch + a_cg * 16 (RS = 2 with the parity tiles), TR x ncg = 16 of them; they are 16
  // DIFFERENT 16-byte bank slots iff (RS * sh * Ws / 8) mod 16 is an odd multiple of ncg.  (Round 6: the res2 pair's image had a 64-byte pitch -
  // rows a_tr and a_tr + 4 on the same banks, and with the parity tiles' two-apart rows a_tr and a_tr + 2: the halved k-loop read twice as slowly
  // and the launch gained nothing.)  The pitch is widened by at most 15 x 16 bytes to the next such value.
  static const bool pitch_on = [] { const char* e = getenv("LAMP_NCV_PITCH"); return !(e && e[0] == '0'); }();
  if (pitch_on && aligned) {
    const int rs = (nke > 0 ? 2 : 1) * q.sh;
    for (int k = 0; k < 16; k++) {
      const int step = (rs * ((q.Ws >> 3) + k)) & 15;
      if (step % ncg == 0 && ((step / ncg) & 1) == 1 && (size_t)q.C * q.Hs * (q.Ws + 8 * k) * 2 <= 64 * 1024) { q.Ws += 8 * k; break; }
    }
  }
  const size_t lds = (size_t)q.C * q.Hs * q.Ws * 2;
  if (lds > 64 * 1024) return false;
  // two output phases per MFMA where the columns allow it (see NcvW)
  static const bool two_shift_on = [] { const char* e = getenv("LAMP_NCV_TWO_SHIFT"); return !(e && e[0] == '0'); }();
  const int NS = (two_shift_on && aligned && q.CO <= 8 && g.kw + q.sw <= 8) ? 2 : 1;
  const Tensor* wsecond = sib ? sib->w : second ? second->w : nullptr;
  const NcvW wq{w->ptr<bf16_t>(), (int)g.Cout, (int)g.Cin, g.kh, g.kw, dgrad ? 1 : 0, NS, q.sw, wsecond ? wsecond->ptr<bf16_t>() : (const bf16_t*)nullptr, cout2, nke};
  Hold wpk_h(ncv_packed_weights(w, wq, st, wsecond));
  const nv_bf8* wpk = reinterpret_cast<const nv_bf8*>(static_cast<const Tensor*>(wpk_h.get())->ptr<bf16_t>());
  static const int max_per_cu = [] { const char* e = getenv("LAMP_NCV_PER_CU"); return e ? std::max(1, atoi(e)) : 4; }();   // A/B on one device: 4 beats 8 and 2
  const int lds_per_cu = (int)std::max<size_t>(1, std::min<size_t>(max_per_cu, (size_t)(150 * 1024) / std::max<size_t>(lds, 1)));
  const bf16_t* bp = bias ? bias->ptr<bf16_t>() : (const bf16_t*)nullptr;
  if (aligned) {
    const int nsuper = q.Ho / (16 / ncg);
    const int threads = 64 * std::min(4, nsuper);
    q.pf = (q.W % 8 == 0 && (int64_t)q.C * q.H * q.W <= (int64_t)NCV_PF * threads * 8) ? 1 : 0;
    if (second && !q.pf) return false;                     // (two sources are staged through the register prefetch only)
    if (NK > 12 && !second) return false;
    // (parity tiles: the kernel's k-steps are the packed image's two halves - 2 nke, whatever the plain form would have taken: 12 k-steps of
    // pairs are 16 there)
    const int NK2 = nke > 0 ? 2 * nke : NK <= 2 ? 2 : (NK <= 4 ? 4 : (NK <= 5 ? 5 : (NK <= 6 && second ? 6 : (NK <= 8 && second ? 8 : (NK <= 12 ? 12 : 16)))));
    const void* kfn = nullptr;
    const bool with_add = addend != nullptr && q.sw == 1;
    // fprop: per-image batch-norm statistics of the output(s) from the epilogue (LAMP_CONV_BN_STATS=0 turns the hand-off off)
    static const bool bn_stats = [] {
      const char* e = getenv("LAMP_CONV_BN_STATS");
      const char* n = getenv("LAMP_NCV_BN_STATS");          // the narrow kernels' alone (A/B)
      return !(e && e[0] == '0') && !(n && n[0] == '0');
    }();
    // (a filter whose output's statistics nobody took last time - the stem of Cnn.resnet feeds res1's convolutions directly - stops paying for them)
    const bool with_stats = bn_stats && !dgrad && !addend && g.N >= 2 && (int64_t)q.Ho * q.Wo >= 64 && (sib || conv_stats_wanted(w->st->uid));
#define NCV_F2(NKv, SWv, PHv, ADDv, STv) kfn = NS == 2 ? (const void*)ncv_fwd2_kernel<NKv, SWv, PHv, 2, ADDv, STv> : (const void*)ncv_fwd2_kernel<NKv, SWv, PHv, 1, ADDv, STv>
#define NCV_F2_PH(NKv, SWv, ADDv, STv) do { if (ph0 == 0) NCV_F2(NKv, SWv, 0, ADDv, STv); else if (ph0 == 6) NCV_F2(NKv, SWv, 6, ADDv, STv); else NCV_F2(NKv, SWv, 7, ADDv, STv); } while (0)
#define NCV_F2_SW(NKv) do { if (with_stats) { if (q.sw == 1) NCV_F2_PH(NKv, 1, false, true); else NCV_F2_PH(NKv, 2, false, true); }                \
                            else if (q.sw == 1) { if (with_add) NCV_F2_PH(NKv, 1, true, false); else NCV_F2_PH(NKv, 1, false, false); }                \
                            else NCV_F2_PH(NKv, 2, false, false); } while (0)
    // (6, 8 and 16 k-steps: the pairs' input gradients only - stride-1 correlations without statistics)
#define NCV_F2P(NKv, PHv, ADDv) kfn = NS == 2 ? (const void*)ncv_fwd2_kernel<NKv, 1, PHv, 2, ADDv, false, true> : (const void*)ncv_fwd2_kernel<NKv, 1, PHv, 1, ADDv, false, true>
#define NCV_F2P_PH(NKv, ADDv) do { if (ph0 == 0) NCV_F2P(NKv, 0, ADDv); else if (ph0 == 6) NCV_F2P(NKv, 6, ADDv); else NCV_F2P(NKv, 7, ADDv); } while (0)
#define NCV_F2_PAIR(NKv) do { if (nke > 0) { if (with_add) NCV_F2P_PH(NKv, true); else NCV_F2P_PH(NKv, false); }                                   \
                              else if (with_add) NCV_F2_PH(NKv, 1, true, false); else NCV_F2_PH(NKv, 1, false, false); } while (0)
    switch (NK2) {
      case 2: NCV_F2_SW(2); break;
      case 4: NCV_F2_SW(4); break;
      case 5: NCV_F2_SW(5); break;
      case 6: NCV_F2_PAIR(6); break;
      case 8: NCV_F2_PAIR(8); break;
      case 16: NCV_F2_PAIR(16); break;
      default: NCV_F2_SW(12); break;
    }
#undef NCV_F2_PAIR
#undef NCV_F2P_PH
#undef NCV_F2P
#undef NCV_F2_SW
#undef NCV_F2_PH
#undef NCV_F2
    // persistent grid: as many workgroups as are really co-resident (registers and LDS), each walks a strided range of images
    const int per_cu = std::min(lds_per_cu, kernel_occupancy(kfn, threads, lds));
    const int blocks = (int)std::min<int64_t>(g.N, (int64_t)num_cus() * per_cu);
    // (the sibling's work is declared with the launch; its input is the one already counted)
    const double sib_fl = sib ? 2.0 * (double)g.N * cout2 * (double)g.Ho * g.Wo * (double)g.Cin : 0.0;
    const double sib_by = sib ? ((double)g.N * cout2 * g.Ho * g.Wo + (double)cout2 * g.Cin) * 2.0 : 0.0;
    const double sec_fl = second ? 2.0 * (double)g.N * cout2 * (double)g.Ho * g.Wo * (double)g.Cin : 0.0;
    const double sec_by = second ? ((double)g.N * cout2 * g.Ho * g.Wo + (double)cout2 * g.Cin) * 2.0 : 0.0;
    KernelTimer kt(dgrad ? "conv_dgrad_narrow" : "conv_fwd_narrow", conv_flops(g) + sib_fl + sec_fl, conv_bytes(g, 2) + sib_by + sec_by, st);
    const bf16_t* srcp = in->ptr<bf16_t>();
    const bf16_t* src2p = second ? second->dy->ptr<bf16_t>() : (const bf16_t*)nullptr;
    bf16_t* dstp = out->ptr<bf16_t>();
    const bf16_t* addp = with_add ? addend->ptr<bf16_t>() : (const bf16_t*)nullptr;
    if (addend_fused) *addend_fused = with_add;
    bf16_t* dst2p = sib ? sib->out->ptr<bf16_t>() : (bf16_t*)nullptr;
    const bf16_t* bias2p = (sib && sib->bias) ? sib->bias->ptr<bf16_t>() : (const bf16_t*)nullptr;
    int co_a = sib ? (int)g.Cout : q.CO;
    Hold statt, statt2;
    float* statp = nullptr;
    float* stat2p = nullptr;
    // one triple per workgroup where every workgroup walks the same number of images (equal counts), else one per image
    int stats_per_wg = g.N % blocks == 0 ? 1 : 0;
    const int parts = stats_per_wg ? blocks : (int)g.N;
    if (with_stats) {
      int64_t ps[1] = {(int64_t)parts * g.Cout * 3};
      statt = Hold(new_tensor(ps, 1, kF32, in->device()));
      statp = statt->ptr<float>();
      if (sib) {
        int64_t ps2[1] = {(int64_t)parts * cout2 * 3};
        statt2 = Hold(new_tensor(ps2, 1, kF32, in->device()));
        stat2p = statt2->ptr<float>();
      }
    }
    void* args[] = {(void*)&srcp, (void*)&wpk, (void*)&bp, (void*)&dstp, (void*)&q, (void*)&addp, (void*)&dst2p, (void*)&bias2p, (void*)&co_a,
                    (void*)&statp, (void*)&stat2p, (void*)&stats_per_wg, (void*)&src2p};
    HIP_CHECK(hipLaunchKernel(kfn, dim3(blocks), dim3(threads), args, lds, st));
    LAMP_LAUNCH_CHECK();
    if (statt.get()) conv_stats_publish(out, statt.get(), parts, sib ? 0 : w->st->uid);
    if (statt2.get()) conv_stats_publish(sib->out, statt2.get(), parts);
    return true;
  }
  if (NK > 12) return false;
  const int blocks = (int)std::min<int64_t>(g.N, (int64_t)num_cus() * lds_per_cu);
  KernelTimer kt(dgrad ? "conv_dgrad_narrow" : "conv_fwd_narrow", conv_flops(g), conv_bytes(g, 2), st);
  switch (NK) {
    case 1: ncv_launch<1>(in->ptr<bf16_t>(), wpk, bp, out->ptr<bf16_t>(), q, blocks, lds, st); break;
    case 2: ncv_launch<2>(in->ptr<bf16_t>(), wpk, bp, out->ptr<bf16_t>(), q, blocks, lds, st); break;
    case 4: ncv_launch<4>(in->ptr<bf16_t>(), wpk, bp, out->ptr<bf16_t>(), q, blocks, lds, st); break;
    case 5: ncv_launch<5>(in->ptr<bf16_t>(), wpk, bp, out->ptr<bf16_t>(), q, blocks, lds, st); break;
    case 6: case 8: ncv_launch<8>(in->ptr<bf16_t>(), wpk, bp, out->ptr<bf16_t>(), q, blocks, lds, st); break;
    default: ncv_launch<12>(in->ptr<bf16_t>(), wpk, bp, out->ptr<bf16_t>(), q, blocks, lds, st); break;
  }
  LAMP_LAUNCH_CHECK();
  return true;
}

bool narrow_conv_fwd(const Tensor* x, const Tensor* w, const Tensor* bias, Tensor* y, const ConvGeom& g, hipStream_t st) {
  return ncv_run(x, w, bias, y, g, false, st);
}
// y = conv3x3(x, w) and y1 = conv1x1(x, w1), same stride and output map (the two branches of lamp's residual block on its input,
// cnn.scala:16-20), Cout + Cout1 <= 16: one launch, the values of the two separate ones (the second filter is the centre tap of extra
// output columns); false = nothing launched
bool narrow_conv_fwd_pair(const Tensor* x, const Tensor* w, const Tensor* bias, Tensor* y, const ConvGeom& g, const Tensor* w1, const Tensor* bias1,
                          Tensor* y1, const ConvGeom& g1, hipStream_t st) {
  static const bool on = [] { const char* e = getenv("LAMP_CONV_SIBLING"); return !(e && e[0] == '0'); }();
  if (!on || x->dtype != kBF16) return false;
  if (g.kh != 3 || g.kw != 3 || g.ph != 1 || g.pw != 1 || g1.kh != 1 || g1.kw != 1 || g1.ph != 0 || g1.pw != 0) return false;
  if (g.sh != g1.sh || g.sw != g1.sw || g.Ho != g1.Ho || g.Wo != g1.Wo || g.Cin != g1.Cin || g.N != g1.N || g.groups != 1 || g1.groups != 1) return false;
  if (g.dh != 1 || g.dw != 1 || g1.dh != 1 || g1.dw != 1) return false;
  const NcvSibling sb{w1, bias1, y1};
  return ncv_run(x, w, bias, y, g, false, st, nullptr, nullptr, &sb);
}
bool narrow_conv_dgrad(const Tensor* dy, const Tensor* w, Tensor* dx, const ConvGeom& g, hipStream_t st, const Tensor* addend, bool* addend_fused) {
  return ncv_run(dy, w, nullptr, dx, g, true, st, addend, addend_fused);
}
// dx = round(dgrad3x3(dy, w) + dgrad1x1(dy1, w1) [then round(. + addend)]: the input gradient of the two branches of lamp's residual block
// (cnn.scala:16-20; autograd.scala:66-84 accumulates the two partial derivatives) from ONE launch - the second gradient tensor is staged
// as extra channels of the image, the second filter is their centre tap - summed in f32 before the one rounding (the two-launch chain
// rounds each contribution first); false = nothing launched
bool narrow_conv_dgrad_pair(const Tensor* dy, const Tensor* w, const ConvGeom& g, const Tensor* dy1, const Tensor* w1, const ConvGeom& g1, Tensor* dx,
                            hipStream_t st, const Tensor* addend, bool* addend_fused) {
  if (addend_fused) *addend_fused = false;
  static const bool on = [] { const char* e = getenv("LAMP_CONV_DGRAD_PAIR"); return !(e && e[0] == '0'); }();
  if (!on || dy->dtype != kBF16 || dy1->dtype != kBF16) return false;
  if (g.kh != 3 || g.kw != 3 || g.ph != 1 || g.pw != 1 || g1.kh != 1 || g1.kw != 1 || g1.ph != 0 || g1.pw != 0) return false;
  if (g.sh != g1.sh || g.sw != g1.sw || g.Ho != g1.Ho || g.Wo != g1.Wo || g.H != g1.H || g.W != g1.W || g.Cin != g1.Cin || g.N != g1.N) return false;
  if (g.groups != 1 || g1.groups != 1 || g.dh != 1 || g.dw != 1 || g1.dh != 1 || g1.dw != 1 || g1.Cout > 16) return false;
  const NcvSecondGrad sg{dy1, w1};
  return ncv_run(dy, w, nullptr, dx, g, true, st, addend, addend_fused, nullptr, &sg);
}

template <int NT, int SW>
static void ncv_wg_launch(const bf16_t* dy, const bf16_t* x, float* partial, const NcvWGeom& q, int ipb, int blocks, size_t lds, hipStream_t st) {
  static bool attr = false;
  allow_big_lds((const void*)ncv_wgrad_kernel<NT, SW>);
  hipLaunchKernelGGL((ncv_wgrad_kernel<NT, SW>), dim3(blocks), dim3(256), lds, st, dy, x, partial, q, ipb);
}

// second (optional): the output gradient of a sibling 1x1 convolution of the same x and ITS weight gradient: both from the one launch of the
// aligned kernel (false, nothing launched, when that kernel does not take the pair)
struct NcvSecondWgrad { const Tensor* dy; Tensor* dw; };
static bool ncv_wgrad_run(const Tensor* dy, const Tensor* x, Tensor* dw, const ConvGeom& g, hipStream_t st, const NcvSecondWgrad* second) {
  if (!ncv_common(g, x->dtype)) return false;
  if (g.W % 8 != 0 || g.Wo % 8 != 0 || (g.Ho * g.Wo) % 32 != 0) return false;
  const int cout2 = second ? (int)second->dy->sizes[1] : 0;
  if (second && (g.Cout + cout2 > 16 || g.kh != 3 || g.kw != 3 || g.ph != 1 || g.pw != 1)) return false;
  NcvWGeom q;
  q.N = (int)g.N; q.Cin = (int)g.Cin; q.Cout = (int)g.Cout; q.H = (int)g.H; q.W = (int)g.W; q.Ho = (int)g.Ho; q.Wo = (int)g.Wo;
  q.kh = g.kh; q.kw = g.kw; q.ph = g.ph; q.pw = g.pw; q.Cout2 = cout2;
  q.ncol = (int)g.Cin * g.kh * g.kw;
  const int SW = g.sh;
  q.Hs = std::max((int)g.H + 2 * g.ph, ((int)g.Ho - 1) * g.sh + g.kh);
  if (SW == 1) q.Ws = round_up(std::max(NCV_LEFT + (int)g.W, (int)g.Wo - 1 + g.kw - 1 - g.pw + NCV_LEFT + 1), 8);
  else {
    const int half_max = (g.kw - 1 - g.pw) >> 1;
    q.Ws = round_up(std::max((int)g.W / 2, (int)g.Wo + half_max) + NCV_OFF2 + 1, 8);
  }
  const int HoWo = (int)(g.Ho * g.Wo);
  q.IG = std::max(1, 256 / HoWo);                                    // >= 8 chunks of 32 pixels per round
  // aligned-read kernel for the filter shapes of the CIFAR ResNet
  const int npt = (int)((g.Cin * g.kh + 15) / 16);
  const bool aligned = g.kh == g.kw && npt <= 3 &&
                       ((g.kw == 5 && g.pw == 2 && SW == 1) || (g.kw == 3 && g.pw == 1) || (g.kw == 1 && g.pw == 0));
  if (aligned) {
    if (SW == 1) q.Ws = round_up(NCV_OFFA + std::max((int)g.W, (int)g.Wo + 8) + 8, 8);
    else q.Ws = round_up(NCV_OFFA + std::max((int)g.W / 2, (int)g.Wo) + 8, 8);
    const int ximg2 = q.Cin * q.Hs * q.Ws * (SW == 2 ? 2 : 1);
    const int ncolt = npt * g.kw * 16;
    const size_t lds2 = (size_t)q.IG * (ximg2 + 16 * HoWo) * 2 + (size_t)4 * 16 * ncolt * 4;
    if (lds2 <= 150 * 1024) {
      const int O = q.Cout * q.ncol;
      const void* kfn = nullptr;
#define NCV_WG2(KWv, PWv, SWv, NPTv) kfn = (const void*)ncv_wgrad2_kernel<KWv, PWv, SWv, NPTv>
#define NCV_WG2_NPT(KWv, PWv, SWv) do { if (npt == 1) NCV_WG2(KWv, PWv, SWv, 1); else if (npt == 2) NCV_WG2(KWv, PWv, SWv, 2); else NCV_WG2(KWv, PWv, SWv, 3); } while (0)
      if (g.kw == 5) NCV_WG2_NPT(5, 2, 1);
      else if (g.kw == 3 && SW == 1) NCV_WG2_NPT(3, 1, 1);
      else if (g.kw == 3) NCV_WG2_NPT(3, 1, 2);
      else if (SW == 1) NCV_WG2_NPT(1, 0, 1);
      else NCV_WG2_NPT(1, 0, 2);
#undef NCV_WG2_NPT
#undef NCV_WG2
      allow_big_lds(kfn);
      const int per_cu = std::min((int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(150 * 1024) / lds2)), kernel_occupancy(kfn, 256, lds2));
      const int64_t rounds = (g.N + q.IG - 1) / q.IG;
      const int nb = (int)std::min<int64_t>(rounds, (int64_t)num_cus() * per_cu);
      const int ipb = (int)((rounds + nb - 1) / nb) * q.IG;
      const int nblocks = (int)((g.N + ipb - 1) / ipb);
      int64_t ps[1] = {(int64_t)nblocks * O};
      Hold partial(new_tensor(ps, 1, kF32, dy->device()));
      const int O2 = cout2 * q.Cin;
      Hold partial2;
      if (second) { int64_t ps2[1] = {(int64_t)nblocks * O2}; partial2 = Hold(new_tensor(ps2, 1, kF32, dy->device())); }
      {
        // (the sibling's work is declared with the launch; x is the one already counted)
        const double sec_fl = second ? 2.0 * (double)g.N * cout2 * (double)g.Ho * g.Wo * (double)g.Cin : 0.0;
        const double sec_by = second ? ((double)g.N * cout2 * g.Ho * g.Wo + (double)cout2 * g.Cin) * 2.0 : 0.0;
        KernelTimer kt("conv_wgrad_narrow", conv_flops(g) + sec_fl, conv_bytes(g, 2) + sec_by, st);
        const bf16_t* dp = dy->ptr<bf16_t>(); const bf16_t* xp = x->ptr<bf16_t>(); float* pp = partial->ptr<float>();
        const bf16_t* dp2 = second ? second->dy->ptr<bf16_t>() : (const bf16_t*)nullptr;
        float* pp2 = second ? partial2->ptr<float>() : (float*)nullptr;
        int ipb_arg = ipb;
        void* args[] = {(void*)&dp, (void*)&xp, (void*)&pp, (void*)&q, (void*)&ipb_arg, (void*)&dp2, (void*)&pp2};
        HIP_CHECK(hipLaunchKernel(kfn, dim3(nblocks), dim3(256), args, lds2, st));
        LAMP_LAUNCH_CHECK();
      }
      WgradReduceArgs ra{};
      ra.kind = 1; ra.O = O; ra.nsplit = nblocks; ra.blocks = (int)(((int64_t)O * 64 + 255) / 256);
      wgrad_reduce_enqueue(ra, partial.get(), dw, st);
      if (second) {
        WgradReduceArgs rb{};
        rb.kind = 1; rb.O = O2; rb.nsplit = nblocks; rb.blocks = (int)(((int64_t)O2 * 64 + 255) / 256);
        wgrad_reduce_enqueue(rb, partial2.get(), second->dw, st);
      }
      return true;
    }
    if (second) return false;
    // does not fit: restore the geometry of the generic kernel
    if (SW == 1) q.Ws = round_up(std::max(NCV_LEFT + (int)g.W, (int)g.Wo - 1 + g.kw - 1 - g.pw + NCV_LEFT + 1), 8);
    else { const int half_max = (g.kw - 1 - g.pw) >> 1; q.Ws = round_up(std::max((int)g.W / 2, (int)g.Wo + half_max) + NCV_OFF2 + 1, 8); }
  }
  if (second) return false;
  const int nt_real = (q.ncol + 15) / 16;
  static const int nt_opts[] = {1, 2, 4, 5, 9};
  int NT = 0;
  for (int o : nt_opts) if (o >= nt_real) { NT = o; break; }
  if (!NT) return false;
  const int ximg = q.Cin * q.Hs * q.Ws * (SW == 2 ? 2 : 1);
  if ((q.IG * (ximg + 16 * HoWo)) % 8 != 0) return false;
  const size_t lds = (size_t)q.IG * (ximg + 16 * HoWo) * 2 + (size_t)4 * 16 * NT * 16 * 4;
  if (lds > 150 * 1024) return false;
  const int O = q.Cout * q.ncol;
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(150 * 1024) / lds));
  const int64_t rounds = (g.N + q.IG - 1) / q.IG;
  const int nb = (int)std::min<int64_t>(rounds, (int64_t)num_cus() * per_cu);
  const int ipb = (int)((rounds + nb - 1) / nb) * q.IG;
  const int nblocks = (int)((g.N + ipb - 1) / ipb);
  int64_t ps[1] = {(int64_t)nblocks * O};
  Hold partial(new_tensor(ps, 1, kF32, dy->device()));
  {
    KernelTimer kt("conv_wgrad_narrow", conv_flops(g), conv_bytes(g, 2), st);
    const bf16_t* dp = dy->ptr<bf16_t>(); const bf16_t* xp = x->ptr<bf16_t>(); float* pp = partial->ptr<float>();
#define NCV_WG(NTv)                                                                                          \
  do {                                                                                                       \
    if (SW == 1) ncv_wg_launch<NTv, 1>(dp, xp, pp, q, ipb, nblocks, lds, st);                                \
    else ncv_wg_launch<NTv, 2>(dp, xp, pp, q, ipb, nblocks, lds, st);                                        \
  } while (0)
    switch (NT) {
      case 1: NCV_WG(1); break;
      case 2: NCV_WG(2); break;
      case 4: NCV_WG(4); break;
      case 5: NCV_WG(5); break;
      default: NCV_WG(9); break;
    }
#undef NCV_WG
    LAMP_LAUNCH_CHECK();
  }
  WgradReduceArgs ra{};
  ra.kind = 1; ra.O = O; ra.nsplit = nblocks; ra.blocks = (int)(((int64_t)O * 64 + 255) / 256);
  wgrad_reduce_enqueue(ra, partial.get(), dw, st);
  return true;
}

bool narrow_conv_wgrad(const Tensor* dy, const Tensor* x, Tensor* dw, const ConvGeom& g, hipStream_t st) { return ncv_wgrad_run(dy, x, dw, g, st, nullptr); }
// dw = wgrad3x3(dy, x) and dw1 = wgrad1x1(dy1, x) - the weight gradients of the two convolutions lamp's residual block applies to its input
// (cnn.scala:16-20) - from one launch that stages x once: the values of the two separate launches (the second gradient tensor rides in the
// spare rows of the MFMA tile, its weight gradient is the centre tap of those rows); false = nothing launched
bool narrow_conv_wgrad_pair(const Tensor* dy, const Tensor* dy1, const Tensor* x, Tensor* dw, Tensor* dw1, const ConvGeom& g, const ConvGeom& g1, hipStream_t st) {
  static const bool on = [] { const char* e = getenv("LAMP_CONV_WGRAD_PAIR"); return !(e && e[0] == '0'); }();
  if (!on || x->dtype != kBF16 || dy->dtype != kBF16 || dy1->dtype != kBF16) return false;
  if (g1.kh != 1 || g1.kw != 1 || g1.ph != 0 || g1.pw != 0) return false;
  if (g.sh != g1.sh || g.sw != g1.sw || g.Ho != g1.Ho || g.Wo != g1.Wo || g.H != g1.H || g.W != g1.W || g.Cin != g1.Cin || g.N != g1.N) return false;
  if (g.groups != 1 || g1.groups != 1 || g.dh != 1 || g.dw != 1 || g1.dh != 1 || g1.dw != 1) return false;
  const NcvSecondWgrad sw{dy1, dw1};
  return ncv_wgrad_run(dy, x, dw, g, st, &sw);
}

}  // namespace lamp
