// Device-side helpers shared by all gfx950 kernels: scalar traits (bf16/f32/f64 with their
// accumulation types), 16-byte vector access, 64-lane wavefront reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#include "../core/tensor.h"

namespace lamp {

// bf16 storage type. Conversion f32->bf16 is a plain cast on the device (hipcc emits
// v_cvt_pk_bf16_f32 on gfx950, round-to-nearest-even, NaN preserving).
struct alignas(2) bf16_t {
  uint16_t bits;
  bf16_t() = default;
  __host__ __device__ explicit bf16_t(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    __bf16 h = (__bf16)f;
    bits = __builtin_bit_cast(uint16_t, h);
#else
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) { bits = (uint16_t)((u >> 16) | 0x40); }
    else { bits = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
#endif
  }
  __host__ __device__ explicit operator float() const {
    uint32_t u = ((uint32_t)bits) << 16;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(float, u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
  }
};

// round(a + b) on packed bf16 pairs / groups of eight: the values of the elementwise add kernel (f32 sum, round to nearest even)
__device__ __forceinline__ unsigned add_bf16x2(unsigned a, unsigned b) {
  const float lo = __uint_as_float(a << 16) + __uint_as_float(b << 16);
  const float hi = __uint_as_float(a & 0xffff0000u) + __uint_as_float(b & 0xffff0000u);
  return (unsigned)bf16_t(lo).bits | ((unsigned)bf16_t(hi).bits << 16);
}
__device__ __forceinline__ uint4 add_bf16x8(const uint4& a, const uint4& b) {
  return make_uint4(add_bf16x2(a.x, b.x), add_bf16x2(a.y, b.y), add_bf16x2(a.z, b.z), add_bf16x2(a.w, b.w));
}

// 16-byte load / store with the non-temporal hint: data that is read exactly once (or written for a reader far in the future) should not
// displace what the next kernels will find in L2 / Infinity Cache
__device__ __forceinline__ uint4 nt_load16(const uint4* p) {
  typedef unsigned int nt_u4 __attribute__((ext_vector_type(4)));
  const nt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u4*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store16(uint4* p, const uint4& v) {
  typedef unsigned int nt_u4 __attribute__((ext_vector_type(4)));
  const nt_u4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<nt_u4*>(p));
}

// f16 (IEEE half) storage type: lamp's HalfPrecision (STen.scala:726-731 scalar-type byte 5; AdamW's mixed-precision KAT runs on it,
// adamw.test.scala:96-127).  Arithmetic happens in f32 (acc_t), the conversions are the hardware's round-to-nearest-even casts.
struct alignas(2) f16_t {
  uint16_t bits;
  f16_t() = default;
  __host__ __device__ explicit f16_t(float f) { _Float16 h = (_Float16)f; bits = __builtin_bit_cast(uint16_t, h); }
  __host__ __device__ explicit operator float() const { return (float)__builtin_bit_cast(_Float16, bits); }
};

template <class T> struct acc_type { using type = T; };
template <> struct acc_type<f16_t> { using type = float; };
template <> struct acc_type<bf16_t> { using type = float; };
template <> struct acc_type<float> { using type = float; };
template <> struct acc_type<double> { using type = double; };
template <> struct acc_type<int64_t> { using type = int64_t; };
template <> struct acc_type<int32_t> { using type = int64_t; };
template <> struct acc_type<uint8_t> { using type = int64_t; };
template <class T> using acc_t = typename acc_type<T>::type;

template <class A, class T> __host__ __device__ inline A load_as(const T& v) { return (A)v; }
template <> __host__ __device__ inline float load_as<float, bf16_t>(const bf16_t& v) { return (float)v; }
template <> __host__ __device__ inline double load_as<double, bf16_t>(const bf16_t& v) { return (double)(float)v; }
template <> __host__ __device__ inline int64_t load_as<int64_t, bf16_t>(const bf16_t& v) { return (int64_t)(float)v; }

template <> __host__ __device__ inline float load_as<float, f16_t>(const f16_t& v) { return (float)v; }
template <> __host__ __device__ inline double load_as<double, f16_t>(const f16_t& v) { return (double)(float)v; }
template <> __host__ __device__ inline int64_t load_as<int64_t, f16_t>(const f16_t& v) { return (int64_t)(float)v; }

template <class T, class A> __host__ __device__ inline T store_as(A v) { return (T)v; }
template <> __host__ __device__ inline f16_t store_as<f16_t, float>(float v) { return f16_t(v); }
template <> __host__ __device__ inline f16_t store_as<f16_t, double>(double v) { return f16_t((float)v); }
template <> __host__ __device__ inline f16_t store_as<f16_t, int64_t>(int64_t v) { return f16_t((float)v); }
template <> __host__ __device__ inline bf16_t store_as<bf16_t, float>(float v) { return bf16_t(v); }
template <> __host__ __device__ inline bf16_t store_as<bf16_t, double>(double v) { return bf16_t((float)v); }
template <> __host__ __device__ inline bf16_t store_as<bf16_t, int64_t>(int64_t v) { return bf16_t((float)v); }

// 16-byte packets
template <class T, int N> struct alignas(sizeof(T) * N) Vec {
  T v[N];
};
template <class T> constexpr int vec_width() { return 16 / sizeof(T); }

// ---- dtype dispatch ---------------------------------------------------------------------------
#define LAMP_DISPATCH_FLOAT(DT, T, ...)                                                           \
  switch (DT) {                                                                                   \
    case ::lamp::kF32: { using T = float; __VA_ARGS__; } break;                                   \
    case ::lamp::kF64: { using T = double; __VA_ARGS__; } break;                                  \
    case ::lamp::kBF16: { using T = ::lamp::bf16_t; __VA_ARGS__; } break;                         \
    case ::lamp::kF16: { using T = ::lamp::f16_t; __VA_ARGS__; } break;                           \
    default: throw ::lamp::Error(std::string(__func__) + ": unsupported floating dtype " +        \
                                 ::lamp::dtype_name(DT));                                         \
  }

#define LAMP_DISPATCH_ALL(DT, T, ...)                                                             \
  switch (DT) {                                                                                   \
    case ::lamp::kF32: { using T = float; __VA_ARGS__; } break;                                   \
    case ::lamp::kF64: { using T = double; __VA_ARGS__; } break;                                  \
    case ::lamp::kBF16: { using T = ::lamp::bf16_t; __VA_ARGS__; } break;                         \
    case ::lamp::kF16: { using T = ::lamp::f16_t; __VA_ARGS__; } break;                           \
    case ::lamp::kI64: { using T = int64_t; __VA_ARGS__; } break;                                 \
    case ::lamp::kI32: { using T = int32_t; __VA_ARGS__; } break;                                 \
    case ::lamp::kU8: case ::lamp::kBool: { using T = uint8_t; __VA_ARGS__; } break;              \
    default: throw ::lamp::Error(std::string(__func__) + ": unsupported dtype " +                 \
                                 ::lamp::dtype_name(DT));                                         \
  }

// ---- wavefront (64 lanes) reductions ----------------------------------------------------------
template <class T> __device__ inline T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
template <class T> __device__ inline T wave_max(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    T o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}
template <class T> __device__ inline T wave_min(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    T o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}

// block-wide sum through LDS; `smem` must hold (blockDim.x/64) elements. All threads get the result.
template <class T> __device__ inline T block_sum(T v, T* smem) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  T r = (lane < nw) ? smem[lane] : T(0);
  r = wave_sum(r);
  return r;
}
template <class T> __device__ inline T block_max(T v, T* smem) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  T r = smem[lane < nw ? lane : 0];
  r = wave_max(r);
  return r;
}

#define LAMP_LAUNCH_CHECK() HIP_CHECK(hipGetLastError())

}  // namespace lamp
