// =============================================================================
// pco_oracle_half.hpp -- IEEE binary16 for the ORACLE (test infrastructure), behaving like the `half` crate's f16 the reference
// uses for its f16 number type (pco/Cargo.toml: half; data_types/float.rs:254-366): every arithmetic operation converts to f32,
// computes there and rounds back to nearest-even (half's portable path; with hardware f16 the results are the same -- f32 carries
// 2p + 2 bits for p = 11, so the double rounding is innocuous for + - * /).  from_f64 rounds directly from the double.
// =============================================================================
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace pco_oracle {

struct Half {
  uint16_t b = 0;
  Half() = default;
  static Half from_bits(uint16_t x) { Half h; h.b = x; return h; }
  // round-to-nearest-even from a wider binary format given as (sign, unbiased exponent of the leading 1, 64-bit significand with the
  // leading 1 at bit 63, sticky)
  static uint16_t pack(uint32_t sign, int e, uint64_t sig, bool sticky) {
    if (e > 15) return (uint16_t)(sign | 0x7c00u);                 // overflow -> inf
    int shift;                                                        // bits to drop so that 11 (normal) or fewer (subnormal) bits remain
    if (e >= -14) shift = 63 - 10; else shift = 63 - 10 + (-14 - e);
    if (shift >= 64) {                                                // everything is dropped: rounds to the smallest subnormal only when above half of it
      const bool up = shift == 64 && (sig > (1ull << 63) || (sig == (1ull << 63) && sticky));
      return (uint16_t)(sign | (up ? 1u : 0u));
    }
    uint64_t kept, rem; const bool st = sticky;
    kept = sig >> shift; rem = sig & ((1ull << shift) - 1);
    const uint64_t halfway = 1ull << (shift - 1);
    if (rem > halfway || (rem == halfway && (st || (kept & 1)))) kept++;
    if (e >= -14) {                                                   // kept has the implicit 1 at bit 10 (or carried into bit 11)
      uint32_t exp_field = (uint32_t)(e + 15);
      if (kept >> 11) { kept >>= 1; exp_field++; }
      if (exp_field >= 31) return (uint16_t)(sign | 0x7c00u);
      return (uint16_t)(sign | (exp_field << 10) | ((uint32_t)kept & 0x3ffu));
    }
    return (uint16_t)(sign | (uint32_t)kept);                        // subnormal (a carry into bit 10 is the smallest normal: same encoding)
  }
  static Half from_f32(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u, ex = (u >> 23) & 0xffu, man = u & 0x7fffffu;
    if (ex == 0xff) return from_bits((uint16_t)(sign | 0x7c00u | (man ? (0x200u | (man >> 13)) : 0u)));   // inf / NaN (quiet, payload top bits)
    if (ex == 0) return from_bits((uint16_t)sign);                   // zero and f32 subnormals (< 2^-126: far below half of f16's smallest subnormal)
    return from_bits(pack(sign, (int)ex - 127, ((uint64_t)(man | 0x800000u)) << 40, false));
  }
  static Half from_f64(double d) {
    uint64_t u; std::memcpy(&u, &d, 8);
    const uint32_t sign = (uint32_t)(u >> 48) & 0x8000u; const uint32_t ex = (uint32_t)(u >> 52) & 0x7ffu; const uint64_t man = u & 0xfffffffffffffull;
    if (ex == 0x7ff) return from_bits((uint16_t)(sign | 0x7c00u | (man ? (0x200u | (uint32_t)(man >> 42)) : 0u)));
    if (ex == 0) return from_bits((uint16_t)sign);                   // zero and f64 subnormals
    return from_bits(pack(sign, (int)ex - 1023, (man | (1ull << 52)) << 11, false));
  }
  float to_f32() const {
    const uint32_t sign = (uint32_t)(b & 0x8000u) << 16, ex = (b >> 10) & 0x1fu, man = b & 0x3ffu;
    uint32_t u;
    if (ex == 0x1f) u = sign | 0x7f800000u | (man << 13);
    else if (ex == 0) {
      if (man == 0) u = sign;
      else { const int lz = __builtin_clz(man) - 22; u = sign | ((uint32_t)(112 - lz) << 23) | (((man << (lz + 1)) & 0x3ffu) << 13); }   // man * 2^-24
    } else u = sign | ((ex + 112) << 23) | (man << 13);
    float f; std::memcpy(&f, &u, 4); return f;
  }
  double to_f64() const { return (double)to_f32(); }
  // conversions the generic code writes as casts
  Half(double d) : b(from_f64(d).b) {}
  Half(float f) : b(from_f32(f).b) {}
  Half(int i) : b(from_f32((float)i).b) {}
  Half(unsigned i) : b(from_f32((float)i).b) {}
  Half(uint16_t i) : b(from_f32((float)i).b) {}                    // from_latent_numerical (float.rs:363-365)
  Half(unsigned long i) : b(from_f32((float)i).b) {}
  explicit operator double() const { return to_f64(); }
  explicit operator float() const { return to_f32(); }
  explicit operator uint16_t() const {                              // Rust `f32 as u16`: saturating, NaN -> 0
    const float f = to_f32(); if (!(f == f)) return 0; if (f <= 0.0f) return 0; if (f >= 65535.0f) return 65535; return (uint16_t)f;
  }
};
inline Half operator+(Half a, Half c) { return Half::from_f32(a.to_f32() + c.to_f32()); }
inline Half operator-(Half a, Half c) { return Half::from_f32(a.to_f32() - c.to_f32()); }
inline Half operator*(Half a, Half c) { return Half::from_f32(a.to_f32() * c.to_f32()); }
inline Half operator/(Half a, Half c) { return Half::from_f32(a.to_f32() / c.to_f32()); }
inline Half operator-(Half a) { return Half::from_bits((uint16_t)(a.b ^ 0x8000u)); }
inline Half& operator+=(Half& a, Half c) { a = a + c; return a; }
inline bool operator<(Half a, Half c) { return a.to_f32() < c.to_f32(); }
inline bool operator<=(Half a, Half c) { return a.to_f32() <= c.to_f32(); }
inline bool operator>(Half a, Half c) { return a.to_f32() > c.to_f32(); }
inline bool operator>=(Half a, Half c) { return a.to_f32() >= c.to_f32(); }
inline bool operator==(Half a, Half c) { return a.to_f32() == c.to_f32(); }
inline bool operator!=(Half a, Half c) { return a.to_f32() != c.to_f32(); }

}  // namespace pco_oracle
