// pco_dev.h -- device-side helpers shared by the gfx950 decode and encode kernels.
// Written for CDNA4 only: wavefront = 64 lanes, LDS staging, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pco_gfx.h"

namespace pcogfx {

// format constants (reference: pco/src/constants.rs:10-62, standalone/constants.rs:4-9)
constexpr uint32_t kBatchN = 256;
constexpr uint32_t kAnsInterleaving = 4;
constexpr uint32_t kMaxAnsBits = 14;
constexpr uint32_t kMaxEntries = 1u << 24;
constexpr uint32_t kBitsAnsSizeLog = 4, kBitsModeVariant = 4, kBitsDeltaVariant = 4, kBitsDeltaOrder = 3;
constexpr uint32_t kBitsLookbackWindowLog = 5, kBitsLookbackStateLog = 4, kBitsNBins = 15, kBitsQuantK = 8;
constexpr uint32_t kBitsDictLen = 25, kBitsNEntries = 24, kBitsVarintPower = 6;
constexpr uint32_t kMaxLookbackWindowLog = 24;
constexpr uint32_t kWave = 64;

enum ModeKind : uint32_t { kClassic = 0, kIntMult = 1, kFloatMult = 2, kFloatQuant = 3, kDict = 4 };
enum DeltaKind : uint32_t { kDeltaNone = 0, kDeltaConsecutive = 1, kDeltaLookback = 2, kDeltaConv1 = 3 };
enum NumKind : uint32_t { kUnsigned = 0, kSigned = 1, kFloat = 2 };

__host__ __device__ inline int dtype_bits(uint32_t d) {
  switch (d) {
    case PCO_TYPE_U32: case PCO_TYPE_I32: case PCO_TYPE_F32: return 32;
    case PCO_TYPE_U64: case PCO_TYPE_I64: case PCO_TYPE_F64: return 64;
    case PCO_TYPE_U16: case PCO_TYPE_I16: case PCO_TYPE_F16: return 16;
    case PCO_TYPE_U8: case PCO_TYPE_I8: return 8;
  }
  return 0;
}
__host__ __device__ inline uint32_t dtype_kind(uint32_t d) {
  switch (d) {
    case PCO_TYPE_I32: case PCO_TYPE_I64: case PCO_TYPE_I16: case PCO_TYPE_I8: return kSigned;
    case PCO_TYPE_F32: case PCO_TYPE_F64: case PCO_TYPE_F16: return kFloat;
    default: return kUnsigned;
  }
}
__host__ __device__ inline uint32_t offset_bits_bits(int latent_bits) {  // bits.rs:24-26
  return latent_bits == 8 ? 4 : latent_bits == 16 ? 5 : latent_bits == 32 ? 6 : 7;
}

template <class L> struct LBits;
template <> struct LBits<uint8_t> { static constexpr uint32_t v = 8; };
template <> struct LBits<uint16_t> { static constexpr uint32_t v = 16; };
template <> struct LBits<uint32_t> { static constexpr uint32_t v = 32; };
template <> struct LBits<uint64_t> { static constexpr uint32_t v = 64; };
template <class L> __host__ __device__ constexpr L lmid() { return (L)((L)1 << (LBits<L>::v - 1)); }

// order preserving bijections (data_types/unsigned.rs:155-161, signed.rs:46-52, float.rs:392-411)
// Branch-free: `kind` is wave-uniform, the masks below stay in SGPRs.
template <class L> struct SignedOf;
template <> struct SignedOf<uint8_t> { typedef int8_t T; };
template <> struct SignedOf<uint16_t> { typedef int16_t T; };
template <> struct SignedOf<uint32_t> { typedef int32_t T; };
template <> struct SignedOf<uint64_t> { typedef int64_t T; };
template <class L> __device__ __forceinline__ L to_latent_ordered(L bits, uint32_t kind) {
  typedef typename SignedOf<L>::T S;
  const L neg = (L)((S)bits >> (LBits<L>::v - 1));                     // all ones for a set sign bit
  const L m = (L)((kind == kFloat ? neg : (L)0) | (kind == kUnsigned ? (L)0 : lmid<L>()));
  return (L)(bits ^ m);   // unsigned: x; signed: x ^ MID; float: sign ? !x : x ^ MID
}
template <class L> __device__ __forceinline__ L from_latent_ordered(L l, uint32_t kind) {
  typedef typename SignedOf<L>::T S;
  const L pos = (L)((S)(L)~l >> (LBits<L>::v - 1));                    // all ones for a clear top bit
  const L m = (L)((kind == kFloat ? pos : (L)0) | (kind == kUnsigned ? (L)0 : lmid<L>()));
  return (L)(l ^ m);      // unsigned: l; signed: l ^ MID; float: top ? l ^ MID : !l
}

__device__ __forceinline__ uint32_t clz_u32(uint32_t x) { return x == 0 ? 32u : (uint32_t)__builtin_clz(x); }
__device__ __forceinline__ uint32_t clz_u64(uint64_t x) { return x == 0 ? 64u : (uint32_t)__builtin_clzll(x); }
template <class L> __device__ __forceinline__ uint32_t bitlen(L x) {
  if constexpr (sizeof(L) == 8) return 64u - clz_u64((uint64_t)x);
  else return 32u - clz_u32((uint32_t)x);
}

// ---- f16 (data_types/float.rs:254-366): the reference's f16 is the `half` crate's, whose arithmetic converts to f32, computes there
//      and rounds back to nearest-even; the same is done here, with NaNs converted by bit manipulation (sign, all-ones exponent, quiet
//      bit, top payload bits) so that they do not depend on the conversion instruction's NaN rule ----
__device__ __forceinline__ float half_bits_to_f32(uint32_t h) {
  if ((h & 0x7c00u) == 0x7c00u) return __uint_as_float(((h & 0x8000u) << 16) | 0x7f800000u | ((h & 0x3ffu) << 13));
  return (float)__builtin_bit_cast(_Float16, (uint16_t)h);
}
__device__ __forceinline__ uint32_t f32_to_half_bits(float f) {
  const uint32_t u = __float_as_uint(f);
  if ((u & 0x7f800000u) == 0x7f800000u) { const uint32_t man = u & 0x7fffffu; return ((u >> 16) & 0x8000u) | 0x7c00u | (man ? (0x200u | (man >> 13)) : 0u); }
  return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)f);   // v_cvt_f16_f32: round to nearest even
}
__device__ __forceinline__ uint32_t half_mul(uint32_t a, uint32_t b) { return f32_to_half_bits(half_bits_to_f32(a) * half_bits_to_f32(b)); }
__device__ __forceinline__ uint32_t half_round(uint32_t a) { return f32_to_half_bits(roundf(half_bits_to_f32(a))); }
// int_float_from_latent / int_float_to_latent for f16 (float.rs:324-361): 11 mantissa digits
__device__ __forceinline__ uint32_t half_int_float_from_latent(uint32_t l) {   // -> f16 bits
  const uint32_t mid = 0x8000u; const bool negative = l < mid;
  const uint32_t abs_int = negative ? mid - 1 - l : l - mid, gpi = 1u << 11;
  uint32_t abs_bits;
  if (abs_int < gpi) abs_bits = f32_to_half_bits((float)abs_int);
  else abs_bits = (f32_to_half_bits((float)gpi) + (abs_int - gpi)) & 0xffffu;
  return negative ? (abs_bits ^ 0x8000u) : abs_bits;
}
__device__ __forceinline__ uint32_t half_int_float_to_latent(uint32_t bits) {   // f16 bits -> latent
  const uint32_t abs_bits = bits & 0x7fffu, gpi = 1u << 11, gpi_bits = 0x6800u;   // 2048.0
  const float absf = half_bits_to_f32(abs_bits);
  uint32_t abs_int;
  if (absf < 2048.0f) abs_int = (uint32_t)absf;   // (NaN compares false and lands in the other branch, like the reference)
  else abs_int = (gpi + (abs_bits - gpi_bits)) & 0xffffu;
  return ((bits & 0x8000u) ? (0x8000u - 1 - abs_int) : (0x8000u + abs_int)) & 0xffffu;
}

// ---- float <-> int-float latents (data_types/float.rs:208-244), f32/f64 ----
template <class L> struct FloatOf;
template <> struct FloatOf<uint32_t> { typedef float F; static constexpr int kMantDigits = 24; };
template <> struct FloatOf<uint64_t> { typedef double F; static constexpr int kMantDigits = 53; };
__device__ __forceinline__ float bits_to_float(uint32_t b) { return __uint_as_float(b); }
__device__ __forceinline__ double bits_to_float(uint64_t b) { return __longlong_as_double((long long)b); }
__device__ __forceinline__ uint32_t float_to_bits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ uint64_t float_to_bits(double f) { return (uint64_t)__double_as_longlong(f); }

template <class L> __device__ __forceinline__ typename FloatOf<L>::F int_float_from_latent(L l) {
  typedef typename FloatOf<L>::F F;
  const L mid = lmid<L>();
  const bool negative = l < mid;
  const L abs_int = negative ? (L)(mid - 1 - l) : (L)(l - mid);
  const L gpi = (L)1 << FloatOf<L>::kMantDigits;
  F abs_float;
  if (abs_int < gpi) abs_float = (F)abs_int;
  else abs_float = bits_to_float((L)(float_to_bits((F)gpi) + (abs_int - gpi)));
  return negative ? -abs_float : abs_float;
}
template <class L> __device__ __forceinline__ L int_float_to_latent(typename FloatOf<L>::F x) {
  typedef typename FloatOf<L>::F F;
  const L bits = float_to_bits(x);
  const L abs_bits = (L)(bits & ~lmid<L>());
  const F abs = bits_to_float(abs_bits);
  const L gpi = (L)1 << FloatOf<L>::kMantDigits;
  const F gpi_float = (F)gpi;
  L abs_int;
  if (abs < gpi_float) abs_int = (L)abs;  // exact: abs is an integer-valued float below 2^mant
  else abs_int = (L)(gpi + (abs_bits - float_to_bits(gpi_float)));  // also NaN / inf
  return (bits & lmid<L>()) ? (L)(lmid<L>() - 1 - abs_int) : (L)(lmid<L>() + abs_int);
}
// Rust f32::round / f64::round: half away from zero
__device__ __forceinline__ float round_half_away(float x) { return roundf(x); }
__device__ __forceinline__ double round_half_away(double x) { return round(x); }

// ---- wave64 primitives ----
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
template <class T> __device__ __forceinline__ T shfl_up(T v, int d) {
  if constexpr (sizeof(T) == 8) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)((uint64_t)v >> 32);
    lo = __shfl_up(lo, d, 64); hi = __shfl_up(hi, d, 64);
    return (T)(((uint64_t)hi << 32) | lo);
  } else return (T)__shfl_up((uint32_t)v, d, 64);
}
template <class T> __device__ __forceinline__ T shfl_idx(T v, int src) {
  if constexpr (sizeof(T) == 8) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)((uint64_t)v >> 32);
    lo = __shfl(lo, src, 64); hi = __shfl(hi, src, 64);
    return (T)(((uint64_t)hi << 32) | lo);
  } else return (T)__shfl((uint32_t)v, src, 64);
}
// inclusive wave scan (wrapping add) on the DPP network: 4 row_shr steps inside each row of 16 lanes, then row_bcast:15
// into rows 1 and 3 and row_bcast:31 into rows 2 and 3.  Lanes that have no source read 0.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true); }
template <int CTRL, int ROW_MASK, class T> __device__ __forceinline__ T dpp0_t(T v) {
  if constexpr (sizeof(T) == 8) return (T)(((uint64_t)dpp0<CTRL, ROW_MASK>((uint32_t)((uint64_t)v >> 32)) << 32) | dpp0<CTRL, ROW_MASK>((uint32_t)v));
  else return (T)dpp0<CTRL, ROW_MASK>((uint32_t)v);
}
template <class T> __device__ __forceinline__ T wave_incl_scan(T v) {
  v = (T)(v + dpp0_t<0x111, 0xf>(v));
  v = (T)(v + dpp0_t<0x112, 0xf>(v));
  v = (T)(v + dpp0_t<0x114, 0xf>(v));
  v = (T)(v + dpp0_t<0x118, 0xf>(v));
  v = (T)(v + dpp0_t<0x142, 0xa>(v));
  v = (T)(v + dpp0_t<0x143, 0xc>(v));
  return v;
}
// value of lane 63, wave-uniform
template <class T> __device__ __forceinline__ T wave_last(T v) {
  if constexpr (sizeof(T) == 8) return (T)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63));
  else return (T)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
}
template <class T> __device__ __forceinline__ T wave_sum(T v) { return wave_last(wave_incl_scan(v)); }
// The value of lane (lane ^ J), J a power of two, on the VALU's own cross-lane paths (DPP inside a row of 16 lanes, gfx950's
// v_permlane16_swap / v_permlane32_swap across rows): a butterfly through __shfl_xor is a ds_bpermute per step, i.e. LDS-pipe
// traffic, and the kernels that sort in registers are LDS-bound as it is.
template <uint32_t J> __device__ __forceinline__ uint32_t xor_lane32(uint32_t v) {
  static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "a power of two below 64");
  if constexpr (J == 1) return dpp0<0xB1, 0xf>(v);                               // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return dpp0<0x4E, 0xf>(v);                          // quad_perm [2,3,0,1]
  else if constexpr (J == 4) return dpp0<0x1B, 0xf>(dpp0<0x141, 0xf>(v));        // row_half_mirror (lane ^ 7), then quad_perm [3,2,1,0] (lane ^ 3)
  else if constexpr (J == 8) return dpp0<0x128, 0xf>(v);                         // row_ror:8
  else if constexpr (J == 16) {   // vdst' = {d0, s0, d2, s2}, vsrc' = {d1, s1, d3, s3} (rows of 16 lanes)
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (threadIdx.x & 16u) ? (uint32_t)r[0] : (uint32_t)r[1];
  } else {                        // vdst' = {d_lo, s_lo}, vsrc' = {d_hi, s_hi} (halves of 32 lanes)
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (threadIdx.x & 32u) ? (uint32_t)r[0] : (uint32_t)r[1];
  }
}
template <uint32_t J, class T> __device__ __forceinline__ T xor_lane(T v) {
  if constexpr (sizeof(T) == 8) return (T)(((uint64_t)xor_lane32<J>((uint32_t)((uint64_t)v >> 32)) << 32) | xor_lane32<J>((uint32_t)v));
  else return (T)xor_lane32<J>((uint32_t)v);
}
// ---- quad (4 consecutive lanes) helpers ----
template <int CTRL> __device__ __forceinline__ uint32_t quad_dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
// 4x4 byte transpose across the four lanes of a quad: lane j ends up with byte j of every lane's dword
__device__ __forceinline__ uint32_t quad_transpose_u8(uint32_t d, uint32_t j) {
  const uint32_t d0 = quad_dpp<0x00>(d), d1 = quad_dpp<0x55>(d), d2 = quad_dpp<0xAA>(d), d3 = quad_dpp<0xFF>(d);
  const uint32_t sel = 0x0c0c0000u | ((4u + j) << 8) | j;
  const uint32_t lo = __builtin_amdgcn_perm(d1, d0, sel), hi = __builtin_amdgcn_perm(d3, d2, sel);
  return lo | (hi << 16);
}
// 4x4 u16 transpose: each lane holds (a: entries 0,1 | b: entries 2,3); lane j ends up with entry j of every lane
__device__ __forceinline__ void quad_transpose_u16(uint32_t& a, uint32_t& b, uint32_t j) {
  const uint32_t a0 = quad_dpp<0x00>(a), a1 = quad_dpp<0x55>(a), a2 = quad_dpp<0xAA>(a), a3 = quad_dpp<0xFF>(a);
  const uint32_t b0 = quad_dpp<0x00>(b), b1 = quad_dpp<0x55>(b), b2 = quad_dpp<0xAA>(b), b3 = quad_dpp<0xFF>(b);
  const bool lo = j < 2;
  const uint32_t x0 = lo ? a0 : b0, x1 = lo ? a1 : b1, x2 = lo ? a2 : b2, x3 = lo ? a3 : b3;
  const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
  a = __builtin_amdgcn_perm(x1, x0, sel); b = __builtin_amdgcn_perm(x3, x2, sel);
}

// f over the whole wave, wave-uniform, for an idempotent f (min / max / or): the scan's DPP steps with lanes that have no source keeping
// their own value, the total read from lane 63.  Eight dependent VALU steps where a ds_bpermute butterfly is six LDS round trips.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_self(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false); }
template <class F> __device__ __forceinline__ uint32_t wave_reduce_u32(uint32_t v, F f) {
  v = f(v, dpp_self<0x111, 0xf>(v)); v = f(v, dpp_self<0x112, 0xf>(v)); v = f(v, dpp_self<0x114, 0xf>(v)); v = f(v, dpp_self<0x118, 0xf>(v));
  v = f(v, dpp_self<0x142, 0xa>(v)); v = f(v, dpp_self<0x143, 0xc>(v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
template <class T, class F> __device__ __forceinline__ T wave_butterfly(T v, F f) {   // every lane ends up with f over the whole wave
  v = f(v, xor_lane<32>(v)); v = f(v, xor_lane<16>(v)); v = f(v, xor_lane<8>(v));
  v = f(v, xor_lane<4>(v)); v = f(v, xor_lane<2>(v)); v = f(v, xor_lane<1>(v));
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return wave_reduce_u32(v, [](uint32_t a, uint32_t b) { return a > b ? a : b; }); }
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) { return wave_reduce_u32(v, [](uint32_t a, uint32_t b) { return a | b; }); }

// ---- explicit global address space (pointers loaded from task structs are generic/flat otherwise;
//      flat loads would also tick lgkmcnt and serialise against the LDS traffic of the walkers) ----
#define PCO_GLOBAL __attribute__((address_space(1)))
typedef const uint8_t PCO_GLOBAL* gcptr_u8;
typedef uint8_t PCO_GLOBAL* gptr_u8;
template <class T> __device__ __forceinline__ const T PCO_GLOBAL* as_global(const T* p) { return (const T PCO_GLOBAL*)p; }
template <class T> __device__ __forceinline__ T PCO_GLOBAL* as_global(T* p) { return (T PCO_GLOBAL*)p; }

// ---- unaligned little-endian loads from global memory (amdhsa enables unaligned-access-mode) ----
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
__device__ __forceinline__ uint64_t load_u64_le(gcptr_u8 p) { return *(const u64_unaligned PCO_GLOBAL*)p; }
struct __attribute__((packed, aligned(1))) u128_unaligned { uint64_t lo, hi; };
__device__ __forceinline__ uint32_t load_u32_le(gcptr_u8 p) { return *(const u32_unaligned PCO_GLOBAL*)p; }
// bounds-safe: bytes at or beyond `len` read as zero
__device__ __forceinline__ uint64_t load_u64_le_safe(gcptr_u8 base, uint64_t byte, uint64_t len) {
  if (byte + 8 <= len) return load_u64_le(base + byte);
  uint64_t v = 0;
  for (int i = 0; i < 8; i++) if (byte + i < len) v |= (uint64_t)base[byte + i] << (8 * i);
  return v;
}
// wave-uniform values: move to SGPRs so loops / branches on them are scalar
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni(uint64_t v) {
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// A slow, bounds-safe serial bit reader used for metadata (a few hundred fields per chunk).
struct MetaReader {
  gcptr_u8 src; uint64_t len; uint64_t bit;
  __device__ __forceinline__ uint64_t peek(uint64_t at_bit, uint32_t n) const {  // n <= 64
    if (n == 0) return 0;
    const uint64_t byte = at_bit >> 3; const uint32_t sh = (uint32_t)(at_bit & 7);
    uint64_t v = load_u64_le_safe(src, byte, len) >> sh;
    if (sh + n > 64) v |= load_u64_le_safe(src, byte + 8, len) << (64 - sh);
    if (n < 64) v &= ((uint64_t)1 << n) - 1;
    return v;
  }
  __device__ __forceinline__ uint64_t read(uint32_t n) { uint64_t v = uni(peek(bit, n)); bit += n; return v; }
  __device__ __forceinline__ bool in_bounds() const { return bit <= len * 8; }
  // bit_reader.rs:237-247: returns false on non-zero padding
  __device__ __forceinline__ bool drain_empty_byte() {
    const uint32_t sh = (uint32_t)(bit & 7);
    if (sh == 0) return true;
    const uint64_t byte = bit >> 3;
    const uint32_t b = byte < len ? src[byte] : 0;
    bit += 8 - sh;
    return (b >> sh) == 0;
  }
};

}  // namespace pcogfx
