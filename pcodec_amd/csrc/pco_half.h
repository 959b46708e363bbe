// pco_half.h -- host-side IEEE binary16 for the O(metadata) float logic on f16 chunks (explicit TryFloatMult bases, Auto mode detection).
// The reference's f16 is the `half` crate's (data_types/float.rs:254-366): arithmetic = convert to f32, operate, round to nearest even;
// from_f64 rounds once, from the double.  Device code does the same with the conversion instructions (pco_dev.h).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace pcogfx {

struct F16 {
  uint16_t bits = 0;
  F16() = default;
  static F16 raw(uint16_t b) { F16 h; h.bits = b; return h; }
  // Round a finite, non-zero magnitude m * 2^e2 (m an integer, exact) to binary16, nearest even: scale onto the target's quantum
  // (2^-24 below the normal range, 2^(e-10) inside it) and let the 64-bit integer arithmetic do the rounding.
  static uint16_t round_mag(uint64_t m, int e2) {
    const int top = 63 - __builtin_clzll(m);            // magnitude in [2^(top+e2), 2^(top+e2+1))
    int e = top + e2;                                    // exponent of the leading one
    if (e > 15) return 0x7c00;
    const int q = e >= -14 ? e - 10 : -24;               // quantum exponent
    const int drop = q - e2;                             // low bits of m below the quantum
    uint64_t k;
    if (drop <= 0) k = m << (-drop);
    else if (drop > 63) k = 0;                           // (m < 2^63 <= half a quantum ... or exactly representable cases never reach here)
    else {
      const uint64_t rem = m & ((1ull << drop) - 1), half = 1ull << (drop - 1);
      k = m >> drop;
      if (rem > half || (rem == half && (k & 1))) k++;
    }
    if (drop > 63) { return 0; }
    if (e >= -14) {                                      // k in [2^10, 2^11]: a carry to 2^11 moves to the next binade
      if (k == (1ull << 11)) { k >>= 1; e++; if (e > 15) return 0x7c00; }
      return (uint16_t)(((uint32_t)(e + 15) << 10) | ((uint32_t)k & 0x3ffu));
    }
    return (uint16_t)k;                                  // subnormal (k == 2^10 is the smallest normal: same bits)
  }
  static F16 from_f32(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u); const uint32_t ex = (u >> 23) & 0xffu, man = u & 0x7fffffu;
    if (ex == 0xff) return raw((uint16_t)(sign | 0x7c00u | (man ? (0x200u | (man >> 13)) : 0u)));
    if (ex == 0) return raw(sign);                      // zeros and f32 subnormals: below half of the smallest f16 subnormal
    return raw((uint16_t)(sign | round_mag((uint64_t)(man | 0x800000u), (int)ex - 127 - 23)));
  }
  static F16 from_f64(double d) {
    uint64_t u; std::memcpy(&u, &d, 8);
    const uint16_t sign = (uint16_t)((u >> 48) & 0x8000u); const uint32_t ex = (uint32_t)(u >> 52) & 0x7ffu; const uint64_t man = u & 0xfffffffffffffull;
    if (ex == 0x7ff) return raw((uint16_t)(sign | 0x7c00u | (man ? (0x200u | (uint32_t)(man >> 42)) : 0u)));
    if (ex == 0) return raw(sign);
    return raw((uint16_t)(sign | round_mag(man | (1ull << 52), (int)ex - 1023 - 52)));
  }
  float to_f32() const {
    const uint32_t sign = (uint32_t)(bits & 0x8000u) << 16, ex = (bits >> 10) & 0x1fu, man = bits & 0x3ffu;
    float mag;
    if (ex == 0x1f) { const uint32_t u = sign | 0x7f800000u | (man << 13); float f; std::memcpy(&f, &u, 4); return f; }
    if (ex == 0) mag = std::ldexp((float)man, -24); else mag = std::ldexp((float)(man | 0x400u), (int)ex - 25);
    return sign ? -mag : mag;
  }
  F16(double d) : bits(from_f64(d).bits) {}
  F16(float f) : bits(from_f32(f).bits) {}
  F16(int i) : bits(from_f32((float)i).bits) {}
  F16(unsigned i) : bits(from_f32((float)i).bits) {}
  F16(uint16_t i) : bits(from_f32((float)i).bits) {}
  F16(unsigned long i) : bits(from_f32((float)i).bits) {}
  explicit operator double() const { return (double)to_f32(); }
  explicit operator float() const { return to_f32(); }
  explicit operator uint16_t() const { const float f = to_f32(); if (!(f == f) || f <= 0.0f) return 0; if (f >= 65535.0f) return 65535; return (uint16_t)f; }   // Rust `as`: saturating
};
inline F16 operator+(F16 a, F16 b) { return F16::from_f32(a.to_f32() + b.to_f32()); }
inline F16 operator-(F16 a, F16 b) { return F16::from_f32(a.to_f32() - b.to_f32()); }
inline F16 operator*(F16 a, F16 b) { return F16::from_f32(a.to_f32() * b.to_f32()); }
inline F16 operator/(F16 a, F16 b) { return F16::from_f32(a.to_f32() / b.to_f32()); }
inline F16 operator-(F16 a) { return F16::raw((uint16_t)(a.bits ^ 0x8000u)); }
inline F16& operator+=(F16& a, F16 b) { a = a + b; return a; }
inline bool operator<(F16 a, F16 b) { return a.to_f32() < b.to_f32(); }
inline bool operator<=(F16 a, F16 b) { return a.to_f32() <= b.to_f32(); }
inline bool operator>(F16 a, F16 b) { return a.to_f32() > b.to_f32(); }
inline bool operator>=(F16 a, F16 b) { return a.to_f32() >= b.to_f32(); }
inline bool operator==(F16 a, F16 b) { return a.to_f32() == b.to_f32(); }
inline bool operator!=(F16 a, F16 b) { return a.to_f32() != b.to_f32(); }

}  // namespace pcogfx
