// Packed weight fragments of the narrow matrix-core convolutions (conv_narrow.hip): the fragment layout, its device-side builder and the argument
// block of the multi-image pack kernel - in a header so that the optimiser's re-pack hook can pack these images in the SAME launch as the
// implicit-GEMM ones (conv_igemm.hip, ig_ncv_pack_many_kernel): one launch per step instead of two.
#pragma once
#include "device_utils.h"

namespace lamp {

typedef short nv_s8 __attribute__((ext_vector_type(8)));
typedef __bf16 nv_bf8 __attribute__((ext_vector_type(8)));

// Weight fragments.  Fragment of k-step ks for lane = co + 16*g: pair (c, r) = 4*ks + g, element j = filter column (zero for
// j >= kw);   fprop: W[co][c][r][j]      dgrad (c = conv Cout, "co" = conv Cin): W[c][co][kh-1-r][kw-1-j].
// They are packed ONCE per weight version into a [NCV_NKMAX][64 lanes][8] image (12 KiB; ncv_pack_kernel, cached per
// (storage, view, stream, direction) like the implicit-GEMM images and re-packed in one launch by the optimiser step), so a
// workgroup's prologue is NK 16-byte loads per lane.  Gathering them per lane from the filter tensor (8 two-byte loads and two
// integer divisions per k-step) took 5.5 - 6.5 k of the ~22 k cycles a workgroup lives (scripts/ncv_stamp_probe.py).
constexpr int NCV_NKMAX = 16;
// ns = 2 ("two-shift" images, for at most 8 output channels and kw + sw <= 8): MFMA column n = 8*s + co carries the filter of
// channel co moved s*sw taps to the right inside the 8-wide window, i.e. ONE MFMA produces the output pixels of two neighbouring
// window phases - half the MFMAs (and half the funnel shifts) per output pixel; the 6-channel layers used 6 of 16 columns before.
struct NcvW {
  const bf16_t* w;
  int Cout, Cin, kh, kw, dgrad;
  int ns, sw;            // shifts per MFMA (1 or 2) and the window stride between them
  // round 5, fprop only: a SIBLING 1x1 filter [Cout2][Cin] of the same input (the shortcut of lamp's residual block, cnn.scala:16-20) as
  // output columns Cout .. Cout + Cout2 - 1 whose only non-zero tap is the centre one - the two convolutions are then ONE product over the
  // staged image (the 6-channel layers use 6 of the MFMA's 16 columns: the second convolution rides in the padding)
  // dgrad: w2 = the sibling's filter too, but as EXTRA K pairs (its output gradient is a second source of the staged image, NcvGeom::C1)
  const bf16_t* w2;
  int Cout2;
  // round 6, input gradients of stride-2 convolutions (kh = 3; ncv_fwd2_kernel<.., PAR>): nke > 0 = the K pairs are ordered by the PARITY of their
  // filter row - k-steps [0, nke) hold the pairs (c, r) with r even (pair p: c = p / 2, r = 2 (p % 2)), k-steps [nke, 2 nke) those with r = 1
  // (pair p': c = p') followed by the second source's centre-row pairs.  An output row of the zero-dilated image meets non-zero rows under
  // one parity of r only, so a super-tile of rows of one parity runs half the k-steps.
  int nke;
};
__device__ __forceinline__ nv_bf8 ncv_weight_frag(const NcvW& wq, int ks, int lane) {
  const int n = lane & 15, pair = ks * 4 + (lane >> 4);
  const int co = wq.ns == 2 ? (n & 7) : n, shift = wq.ns == 2 ? (n >> 3) * wq.sw : 0;
  int c = pair / wq.kh, r = pair - c * wq.kh;
  // dgrad of a pair: behind the Cout * kh pairs of the first filter come Cout2 pairs (c2, centre row) of the sibling 1x1 filter [Cout2][Cin]
  bool second_k = wq.dgrad && wq.w2 && pair >= wq.Cout * wq.kh;
  if (second_k) { c = pair - wq.Cout * wq.kh; r = wq.kh / 2; }
  if (wq.nke > 0) {                           // parity order (dgrad, kh == 3): see NcvW::nke
    second_k = false;
    if (ks < wq.nke) { c = pair >> 1; r = 2 * (pair & 1); if (c >= wq.Cout) c = 1 << 20; }                 // (beyond the filter: zero weights)
    else {
      const int p2 = pair - wq.nke * 4;
      if (p2 < wq.Cout) { c = p2; r = 1; }
      else if (wq.w2 && p2 < wq.Cout + wq.Cout2) { second_k = true; c = p2 - wq.Cout; r = 1; }
      else c = 1 << 20;
    }
  }
  nv_s8 v;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    unsigned short e = 0;
    const int t = j - shift;                  // filter column
    if (t >= 0 && t < wq.kw) {
      if (!wq.dgrad) {
        if (co < wq.Cout && c < wq.Cin) e = wq.w[((co * wq.Cin + c) * wq.kh + r) * wq.kw + t].bits;
        else if (wq.w2 && co < wq.Cout + wq.Cout2 && c < wq.Cin && r == wq.kh / 2 && t == wq.kw / 2) e = wq.w2[(co - wq.Cout) * wq.Cin + c].bits;
      }
      else if (second_k) { if (co < wq.Cin && c < wq.Cout2 && t == wq.kw / 2) e = wq.w2[c * wq.Cin + co].bits; }   // (the centre tap is its own mirror image)
      else { if (co < wq.Cin && c < wq.Cout) e = wq.w[((c * wq.Cin + co) * wq.kh + (wq.kh - 1 - r)) * wq.kw + (wq.kw - 1 - t)].bits; }
    }
    v[j] = (short)e;
  }
  return __builtin_bit_cast(nv_bf8, v);
}
constexpr int NCV_PACK_MAX = 16;
struct NcvPackMany {
  NcvW w[NCV_PACK_MAX];
  nv_bf8* dst[NCV_PACK_MAX];
};
// grid (3, entries) x 256 threads: thread = (k-step, lane) of one fragment image
__device__ __forceinline__ void ncv_pack_body(const NcvPackMany& a, int e, int t) {
  if (t >= NCV_NKMAX * 64) return;
  a.dst[e][t] = ncv_weight_frag(a.w[e], t >> 6, t & 63);
}

}  // namespace lamp
