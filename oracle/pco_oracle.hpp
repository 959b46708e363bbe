// =============================================================================
// pco_oracle.hpp -- CPU restatement of the pcodec (pco v1.0.3) chunk encode /
// decode path.  TEST INFRASTRUCTURE ONLY.
//
// This file is the parity ORACLE for the MI355X-native codec in pcodec_amd/.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// import, link, call or execute anything under oracle/.  The product path
// (pcodec_amd/csrc, libpco_gfx.so) never includes or links this code.
//
// It is a from-scratch C++17 restatement of the reference's Rust algorithm
// (the Rust toolchain is absent here, so the reference itself cannot be built;
// see oracle/README.md).  Every function cites the reference file:line it
// follows (paths relative to /root/reference/pco/src).  Pinning status:
//   * DECODE is pinned by the reference's own golden files pco/assets/*.pco
//     (tests/compatibility.rs) -- see tests/test_oracle_golden.py.
//   * ENCODE is pinned by byte-exact re-encoding of v1_0_0_u8.pco /
//     v1_0_0_i8.pco (written by simple_compress of lib 1.0.0 in the current
//     format) plus the reference's inline per-stage known-answer tests.
//
// Rust semantics mirrored here: wrapping integer arithmetic, f32/f64 round()
// = half away from zero, saturating float->int casts, no FMA contraction
// (build with -ffp-contract=off), Iterator::max_by returns the LAST maximum.
// =============================================================================
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

namespace pco_oracle {

// ----------------------------------------------------------------------------
// errors (errors.rs:8-24)
// ----------------------------------------------------------------------------
enum ErrKind : int {
  kOk = 0,
  kCorruption = 1,
  kInsufficientData = 2,
  kInvalidArgument = 3,
  kUnsupported = 4,  // oracle-only: feature outside the hot-path scope
};
struct PcoErr {
  ErrKind kind;
  std::string msg;
};
[[noreturn]] inline void fail(ErrKind k, const std::string& m) { throw PcoErr{k, m}; }

// ----------------------------------------------------------------------------
// constants (constants.rs:10-62, standalone/constants.rs:4-9)
// ----------------------------------------------------------------------------
typedef uint32_t Bitlen;
constexpr Bitlen BITS_TO_ENCODE_ANS_SIZE_LOG = 4;
constexpr Bitlen BITS_TO_ENCODE_MODE_VARIANT = 4;
constexpr Bitlen BITS_TO_ENCODE_DELTA_ENCODING_VARIANT = 4;
constexpr Bitlen BITS_TO_ENCODE_DELTA_ENCODING_ORDER = 3;
constexpr Bitlen BITS_TO_ENCODE_DELTA_CONV_QUANTIZATION = 5;
constexpr Bitlen BITS_TO_ENCODE_DELTA_CONV_N_WEIGHTS = 5;
constexpr Bitlen BITS_TO_ENCODE_DELTA_LOOKBACK_WINDOW_N_LOG = 5;
constexpr Bitlen BITS_TO_ENCODE_DELTA_LOOKBACK_STATE_N_LOG = 4;
constexpr Bitlen BITS_TO_ENCODE_N_BINS = 15;
constexpr Bitlen BITS_TO_ENCODE_QUANTIZE_K = 8;
constexpr Bitlen BITS_TO_ENCODE_DICT_LEN = 25;
constexpr size_t OVERSHOOT_PADDING = 9;
constexpr Bitlen MAX_ANS_BITS = 14;
constexpr Bitlen LIMITED_UNOPTIMIZED_BINS_LOG = 6;
constexpr size_t MAX_COMPRESSION_LEVEL = 12;
constexpr size_t MAX_CONSECUTIVE_DELTA_ORDER = 7;
constexpr size_t MAX_CONV1_DELTA_ORDER = 32;
constexpr size_t MAX_ENTRIES = 1u << 24;
constexpr Bitlen MAX_DELTA_LOOKBACK_WINDOW_N_LOG = 24;
constexpr double MULT_REQUIRED_BITS_SAVED_PER_NUM = 0.5;
constexpr double QUANT_REQUIRED_BITS_SAVED_PER_NUM = 1.5;
constexpr Bitlen CLASSIC_MEMORIZABLE_BINS_LOG = 8;
constexpr size_t DEFAULT_COMPRESSION_LEVEL = 8;
constexpr size_t DEFAULT_MAX_PAGE_N = 1u << 18;
constexpr size_t ANS_INTERLEAVING = 4;
constexpr size_t FULL_BATCH_N = 256;
// FULL_BATCH_N * (16 + 2) + OVERSHOOT_PADDING (constants.rs:28-29)
constexpr size_t MAX_BATCH_LATENT_VAR_SIZE = FULL_BATCH_N * (16 + 2) + OVERSHOOT_PADDING;

constexpr uint8_t MAGIC_HEADER[4] = {112, 99, 111, 33};
constexpr uint8_t MAGIC_TERMINATION_BYTE = 0;
constexpr Bitlen BITS_TO_ENCODE_N_ENTRIES = 24;
constexpr Bitlen BITS_TO_ENCODE_STANDALONE_VERSION = 8;
constexpr Bitlen BITS_TO_ENCODE_VARINT_POWER = 6;
constexpr size_t CURRENT_STANDALONE_VERSION = 3;
constexpr uint8_t FORMAT_MAJOR = 4, FORMAT_MINOR = 1;  // metadata/format_version.rs:29-34

// number type bytes (pco_c/include/cpcodec.h:10-20, docs/format.md:205-217)
enum DType : uint8_t {
  U32 = 1, U64 = 2, I32 = 3, I64 = 4, F32 = 5, F64 = 6,
  U16 = 7, I16 = 8, F16 = 9, U8 = 10, I8 = 11,
};
enum NumKind { kUnsigned = 0, kSigned = 1, kFloat = 2 };
inline bool dtype_valid(uint8_t d) { return d >= 1 && d <= 11; }
inline int dtype_bits(uint8_t d) {
  switch (d) {
    case U32: case I32: case F32: return 32;
    case U64: case I64: case F64: return 64;
    case U16: case I16: case F16: return 16;
    case U8: case I8: return 8;
  }
  return 0;
}
inline NumKind dtype_kind(uint8_t d) {
  switch (d) {
    case I32: case I64: case I16: case I8: return kSigned;
    case F32: case F64: case F16: return kFloat;
    default: return kUnsigned;
  }
}

// ----------------------------------------------------------------------------
// latent helpers (data_types/latent_priv.rs, unsigned.rs:79-130)
// ----------------------------------------------------------------------------
template <class L> struct LT;
template <> struct LT<uint8_t>  { static constexpr Bitlen BITS = 8;  typedef float F; };
template <> struct LT<uint16_t> { static constexpr Bitlen BITS = 16; typedef float F; };
template <> struct LT<uint32_t> { static constexpr Bitlen BITS = 32; typedef float F; };
template <> struct LT<uint64_t> { static constexpr Bitlen BITS = 64; typedef double F; };
template <class L> constexpr L MID() { return (L)((L)1 << (LT<L>::BITS - 1)); }
template <class L> constexpr L LMAX() { return (L)~(L)0; }

template <class L> inline Bitlen leading_zeros(L x) {
  if (x == 0) return LT<L>::BITS;
  return (Bitlen)__builtin_clzll((unsigned long long)x) - (64 - LT<L>::BITS);
}
inline Bitlen clz32(uint32_t x) { return x == 0 ? 32 : (Bitlen)__builtin_clz(x); }
inline Bitlen ilog2_u64(uint64_t x) { return 63 - (Bitlen)__builtin_clzll(x); }
// bits.rs:20-26
template <class L> inline Bitlen bits_to_encode_offset(L max_offset) {
  return LT<L>::BITS - leading_zeros<L>(max_offset);
}
template <class L> constexpr Bitlen bits_to_encode_offset_bits() {
  return LT<L>::BITS == 8 ? 4 : LT<L>::BITS == 16 ? 5 : LT<L>::BITS == 32 ? 6 : 7;
}
inline Bitlen bits_to_encode_offset_bits_rt(int latent_bits) {
  return latent_bits == 8 ? 4 : latent_bits == 16 ? 5 : latent_bits == 32 ? 6 : 7;
}

// order-preserving bijections number bits <-> latent
// (data_types/unsigned.rs:155-161, signed.rs:46-52, float.rs:392-411)
template <class L> inline L to_latent_ordered(L bits, NumKind k) {
  switch (k) {
    case kUnsigned: return bits;
    case kSigned: return (L)(bits ^ MID<L>());
    default: return (bits & MID<L>()) ? (L)~bits : (L)(bits ^ MID<L>());
  }
}
template <class L> inline L from_latent_ordered(L l, NumKind k) {
  switch (k) {
    case kUnsigned: return l;
    case kSigned: return (L)(l ^ MID<L>());
    default: return (l & MID<L>()) ? (L)(l ^ MID<L>()) : (L)~l;
  }
}

// float helpers (data_types/float.rs:135-252); only f32/f64 do arithmetic.
}  // namespace pco_oracle
#include "pco_oracle_half.hpp"
namespace pco_oracle {
template <class L> struct FloatOps;
template <> struct FloatOps<uint32_t> {
  typedef float F; typedef uint32_t L;
  static constexpr Bitlen PRECISION_BITS = 23; static constexpr int MANTISSA_DIGITS = 24;
  static constexpr int EXP_OFFSET = 127;
  static F from_bits(L b) { F f; std::memcpy(&f, &b, 4); return f; }
  static L to_bits(F f) { L b; std::memcpy(&b, &f, 4); return b; }
  static F round(F x) { return ::roundf(x); }
  static F fabs_(F x) { return ::fabsf(x); }
  static F max_value() { return std::numeric_limits<float>::max(); }
};
template <> struct FloatOps<uint64_t> {
  typedef double F; typedef uint64_t L;
  static constexpr Bitlen PRECISION_BITS = 52; static constexpr int MANTISSA_DIGITS = 53;
  static constexpr int EXP_OFFSET = 1023;
  static F from_bits(L b) { F f; std::memcpy(&f, &b, 8); return f; }
  static L to_bits(F f) { L b; std::memcpy(&b, &f, 8); return b; }
  static F round(F x) { return ::round(x); }
  static F fabs_(F x) { return ::fabs(x); }
  static F max_value() { return std::numeric_limits<double>::max(); }
};
template <> struct FloatOps<uint16_t> {   // f16 through the `half` crate's semantics (pco_oracle_half.hpp; data_types/float.rs:254-366)
  typedef Half F; typedef uint16_t L;
  static constexpr Bitlen PRECISION_BITS = 10; static constexpr int MANTISSA_DIGITS = 11;
  static constexpr int EXP_OFFSET = 15;
  static F from_bits(L b) { return Half::from_bits(b); }
  static L to_bits(F f) { return f.b; }
  static F round(F x) { return Half::from_f32(::roundf(x.to_f32())); }
  static F fabs_(F x) { return Half::from_bits((uint16_t)(x.b & 0x7fffu)); }
  static F max_value() { return Half::from_bits(0x7bffu); }
};
template <class L> inline typename FloatOps<L>::F float_from_latent_ordered(L l) {
  return FloatOps<L>::from_bits(from_latent_ordered<L>(l, kFloat));
}
template <class L> inline L float_to_latent_ordered(typename FloatOps<L>::F f) {
  return to_latent_ordered<L>(FloatOps<L>::to_bits(f), kFloat);
}
// float.rs:208-226
template <class L> inline typename FloatOps<L>::F int_float_from_latent(L l) {
  typedef FloatOps<L> FO; typedef typename FO::F F;
  const L mid = MID<L>();
  bool negative; L abs_int;
  if (l >= mid) { negative = false; abs_int = l - mid; } else { negative = true; abs_int = mid - 1 - l; }
  const L gpi = (L)1 << FO::MANTISSA_DIGITS;
  F abs_float;
  if (abs_int < gpi) abs_float = (F)abs_int;
  else abs_float = FO::from_bits(FO::to_bits((F)gpi) + (abs_int - gpi));
  return negative ? -abs_float : abs_float;
}
// float.rs:229-244 (Rust `as` cast saturates, NaN -> 0)
template <class L> inline L int_float_to_latent(typename FloatOps<L>::F x) {
  typedef FloatOps<L> FO; typedef typename FO::F F;
  F abs = FO::fabs_(x);
  const L gpi = (L)1 << FO::MANTISSA_DIGITS;
  F gpi_float = (F)gpi;
  L abs_int;
  if (abs < gpi_float) abs_int = (L)abs;  // abs < 2^53, exact and in range
  else abs_int = gpi + (FO::to_bits(abs) - FO::to_bits(gpi_float));  // NaN lands here too
  bool sign_positive = (FO::to_bits(x) & MID<L>()) == 0;
  return sign_positive ? (L)(MID<L>() + abs_int) : (L)(MID<L>() - 1 - abs_int);
}
template <class L> inline typename FloatOps<L>::F float_exp2(int power) {
  typedef FloatOps<L> FO;
  return FO::from_bits((L)(FO::EXP_OFFSET + power) << FO::PRECISION_BITS);
}
template <class L> inline int float_exponent(typename FloatOps<L>::F x) {
  typedef FloatOps<L> FO;
  return (int)(FO::to_bits(FO::fabs_(x)) >> FO::PRECISION_BITS) - FO::EXP_OFFSET;
}
template <class L> inline uint32_t float_trailing_zeros(typename FloatOps<L>::F x) {
  L b = FloatOps<L>::to_bits(x);
  return b == 0 ? LT<L>::BITS : (uint32_t)__builtin_ctzll((unsigned long long)b);
}
template <class L> inline bool float_is_normal(typename FloatOps<L>::F x) {
  if constexpr (sizeof(L) == 2) { const uint32_t e = (x.b >> 10) & 0x1fu; return e != 0 && e != 0x1f; }
  else return std::isnormal(x);
}

// ----------------------------------------------------------------------------
// bit writer (bit_writer.rs:22-165).  Fields are LSB-first into a little
// endian byte stream.  The reference ORs `val << bits_past_byte` without
// masking; every call site passes val < 2^n, so masking is equivalent.
// ----------------------------------------------------------------------------
struct BitWriter {
  std::vector<uint8_t> buf;
  uint64_t bit_pos = 0;
  void ensure(size_t nbytes) { if (buf.size() < nbytes) buf.resize(nbytes + nbytes / 2 + 64, 0); }
  void write_uint(uint64_t val, Bitlen n) {
    if (n == 0) return;
    if (n < 64) val &= ((uint64_t)1 << n) - 1;
    size_t byte = bit_pos >> 3; Bitlen sh = bit_pos & 7;
    ensure(byte + 17);
    uint64_t cur; std::memcpy(&cur, &buf[byte], 8);
    cur |= val << sh;
    std::memcpy(&buf[byte], &cur, 8);
    if (sh + n > 64) {  // spill (only when sh > 0)
      buf[byte + 8] |= (uint8_t)(val >> (64 - sh));
    }
    bit_pos += n;
  }
  void write_bool(bool b) { write_uint(b ? 1 : 0, 1); }
  void write_aligned_bytes(const uint8_t* p, size_t n) {
    if (bit_pos & 7) fail(kInvalidArgument, "cannot write aligned bytes to unaligned writer");
    size_t byte = bit_pos >> 3; ensure(byte + n + 1);
    std::memcpy(&buf[byte], p, n); bit_pos += 8 * (uint64_t)n;
  }
  void finish_byte() { bit_pos = (bit_pos + 7) & ~(uint64_t)7; }
  size_t byte_len() const { return (size_t)((bit_pos + 7) >> 3); }
};

// ----------------------------------------------------------------------------
// bit reader (bit_reader.rs:30-247).  `src` must be followed by >=
// MAX_BATCH_LATENT_VAR_SIZE zero bytes (the reference's eof_buffer padding,
// bit_reader.rs:270-300); `check_in_bounds` reproduces bit_idx_safe.
// ----------------------------------------------------------------------------
struct BitReader {
  const uint8_t* src; size_t unpadded_bits; size_t padded_len; uint64_t bit_pos;
  inline uint64_t u64_at(size_t byte) const {
    uint64_t v = 0;
    if (byte + 8 <= padded_len) std::memcpy(&v, src + byte, 8);
    else if (byte < padded_len) std::memcpy(&v, src + byte, padded_len - byte);
    return v;
  }
  uint64_t read_uint(Bitlen n) {  // n <= 64
    if (n == 0) return 0;
    size_t byte = bit_pos >> 3; Bitlen sh = bit_pos & 7;
    uint64_t v = u64_at(byte) >> sh;
    if (sh + n > 64) v |= u64_at(byte + 8) << (64 - sh);
    if (n < 64) v &= ((uint64_t)1 << n) - 1;
    bit_pos += n;
    return v;
  }
  bool read_bool() { return read_uint(1) != 0; }
  void check_in_bounds() const {
    if (bit_pos > unpadded_bits) fail(kInsufficientData, "[BitReader] out of bounds");
  }
  const uint8_t* read_aligned_bytes(size_t n) {
    if (bit_pos & 7) fail(kInvalidArgument, "misaligned bit reader");
    const uint8_t* p = src + (bit_pos >> 3); bit_pos += 8 * (uint64_t)n; return p;
  }
  // bit_reader.rs:237-247
  void drain_empty_byte(const char* msg) {
    check_in_bounds();
    Bitlen sh = bit_pos & 7;
    if (sh != 0) {
      size_t byte = bit_pos >> 3;
      uint8_t b = byte < padded_len ? src[byte] : 0;
      if ((b >> sh) > 0) fail(kCorruption, msg);
      bit_pos += 8 - sh;
    }
  }
  size_t aligned_byte_idx() const { return (size_t)(bit_pos >> 3); }
};

// ----------------------------------------------------------------------------
// metadata structs (metadata/*.rs)
// ----------------------------------------------------------------------------
struct DynBin { uint32_t weight; uint64_t lower; Bitlen offset_bits; };
struct LatentVarMeta {  // metadata/chunk_latent_var.rs:86-96
  int latent_bits = 0;
  Bitlen ans_size_log = 0;
  std::vector<DynBin> bins;
  bool present = false;
};
enum ModeKind { kClassic = 0, kIntMult = 1, kFloatMult = 2, kFloatQuant = 3, kDict = 4 };
struct Mode {  // metadata/mode.rs:54-62
  ModeKind kind = kClassic;
  uint64_t base_latent = 0;  // IntMult: base; FloatMult: ordered latent of the float base
  Bitlen k = 0;              // FloatQuant
  std::vector<uint64_t> dict;
};
enum DeltaKind { kDeltaNone = 0, kDeltaConsecutive = 1, kDeltaLookback = 2, kDeltaConv1 = 3 };
struct DeltaEncoding {  // metadata/delta_encoding.rs:86-99
  DeltaKind kind = kDeltaNone;
  size_t order = 0;
  bool secondary_uses_delta = false;
  Bitlen window_n_log = 0, state_n_log = 0;
  // conv1 (decode only)
  Bitlen quantization = 0; int64_t bias = 0; std::vector<int64_t> weights;
};
struct LatentVarDelta {  // LatentVarDeltaEncoding, delta_encoding.rs:61-79
  DeltaKind kind = kDeltaNone; size_t order = 0; Bitlen window_n_log = 0, state_n_log = 0;
  Bitlen quantization = 0; int64_t bias = 0; std::vector<int64_t> weights;
  size_t n_latents_per_state() const {
    switch (kind) {
      case kDeltaNone: return 0;
      case kDeltaConsecutive: return order;
      case kDeltaLookback: return (size_t)1 << state_n_log;
      default: return weights.size();
    }
  }
};
enum VarKey { kVarDelta = 0, kVarPrimary = 1, kVarSecondary = 2 };
// delta_encoding.rs:263-306
inline LatentVarDelta delta_for_latent_var(const DeltaEncoding& d, VarKey key) {
  LatentVarDelta r;
  if (d.kind == kDeltaNone || key == kVarDelta) return r;
  if (key == kVarSecondary && !(d.kind != kDeltaConv1 && d.secondary_uses_delta)) return r;
  r.kind = d.kind; r.order = d.order; r.window_n_log = d.window_n_log; r.state_n_log = d.state_n_log;
  r.quantization = d.quantization; r.bias = d.bias; r.weights = d.weights;
  return r;
}
struct ChunkMeta {  // metadata/chunk.rs:19-30
  Mode mode; DeltaEncoding delta;
  LatentVarMeta vars[3];  // delta, primary, secondary
};

constexpr size_t DELTA_ENCODING_MAX_BIT_SIZE = 4 + 5 + 5 + 64 + MAX_CONV1_DELTA_ORDER * 32;  // delta_encoding.rs:103-107

inline size_t mode_max_bit_size(const Mode& m, int latent_bits) {  // mode.rs:217-228
  size_t payload = 0;
  switch (m.kind) {
    case kClassic: payload = 0; break;
    case kDict: payload = BITS_TO_ENCODE_DICT_LEN + 7 + m.dict.size() * (size_t)latent_bits; break;
    case kFloatMult: case kIntMult: payload = (size_t)latent_bits; break;
    case kFloatQuant: payload = BITS_TO_ENCODE_QUANTIZE_K; break;
  }
  return BITS_TO_ENCODE_MODE_VARIANT + payload;
}
inline size_t bin_exact_bit_size(int latent_bits, Bitlen ans_size_log) {  // metadata/bin.rs:20-22
  return ans_size_log + (size_t)latent_bits + bits_to_encode_offset_bits_rt(latent_bits);
}
inline size_t var_exact_bit_size(const LatentVarMeta& v) {  // chunk_latent_var.rs:170-178
  return BITS_TO_ENCODE_ANS_SIZE_LOG + BITS_TO_ENCODE_N_BINS + v.bins.size() * bin_exact_bit_size(v.latent_bits, v.ans_size_log);
}
inline size_t chunk_meta_max_size(const ChunkMeta& m, int number_latent_bits) {  // chunk.rs:105-113
  size_t bits = mode_max_bit_size(m.mode, number_latent_bits) + DELTA_ENCODING_MAX_BIT_SIZE;
  for (int v = 0; v < 3; v++) if (m.vars[v].present) bits += var_exact_bit_size(m.vars[v]);
  return (bits + 7) / 8;
}
inline size_t chunk_meta_exact_page_meta_size(const ChunkMeta& m) {  // chunk.rs:115-125, chunk_latent_var.rs:180-187
  size_t bits = 0;
  for (int v = 0; v < 3; v++) if (m.vars[v].present) {
    LatentVarDelta d = delta_for_latent_var(m.delta, (VarKey)v);
    bits += (size_t)m.vars[v].ans_size_log * ANS_INTERLEAVING + (size_t)m.vars[v].latent_bits * d.n_latents_per_state();
  }
  return (bits + 7) / 8;
}

// metadata/mode.rs:169-195
inline void write_mode(const Mode& m, int latent_bits, BitWriter& w) {
  w.write_uint((uint64_t)m.kind, BITS_TO_ENCODE_MODE_VARIANT);
  switch (m.kind) {
    case kClassic: break;
    case kIntMult: case kFloatMult: w.write_uint(m.base_latent, (Bitlen)latent_bits); break;
    case kFloatQuant: w.write_uint(m.k, BITS_TO_ENCODE_QUANTIZE_K); break;
    case kDict:
      w.write_uint(m.dict.size(), BITS_TO_ENCODE_DICT_LEN); w.finish_byte();
      for (uint64_t x : m.dict) w.write_uint(x, (Bitlen)latent_bits);
      break;
  }
}
// metadata/delta_encoding.rs:204-254
inline void write_delta_encoding(const DeltaEncoding& d, BitWriter& w) {
  w.write_uint((uint64_t)d.kind, BITS_TO_ENCODE_DELTA_ENCODING_VARIANT);
  switch (d.kind) {
    case kDeltaNone: break;
    case kDeltaConsecutive:
      w.write_uint(d.order, BITS_TO_ENCODE_DELTA_ENCODING_ORDER); w.write_bool(d.secondary_uses_delta); break;
    case kDeltaLookback:
      w.write_uint(d.window_n_log - 1, BITS_TO_ENCODE_DELTA_LOOKBACK_WINDOW_N_LOG);
      w.write_uint(d.state_n_log, BITS_TO_ENCODE_DELTA_LOOKBACK_STATE_N_LOG);
      w.write_bool(d.secondary_uses_delta); break;
    case kDeltaConv1: fail(kUnsupported, "conv1 encode is out of scope");
  }
}
// metadata/chunk_latent_var.rs:55-71,158-168
inline void write_latent_var_meta(const LatentVarMeta& v, BitWriter& w) {
  w.write_uint(v.ans_size_log, BITS_TO_ENCODE_ANS_SIZE_LOG);
  w.write_uint(v.bins.size(), BITS_TO_ENCODE_N_BINS);
  Bitlen obb = bits_to_encode_offset_bits_rt(v.latent_bits);
  for (const DynBin& b : v.bins) {
    w.write_uint(b.weight - 1, v.ans_size_log);
    w.write_uint(b.lower, (Bitlen)v.latent_bits);
    w.write_uint(b.offset_bits, obb);
  }
}
// metadata/chunk.rs:176-189
inline void write_chunk_meta(const ChunkMeta& m, int number_latent_bits, BitWriter& w) {
  write_mode(m.mode, number_latent_bits, w);
  write_delta_encoding(m.delta, w);
  for (int v = 0; v < 3; v++) if (m.vars[v].present) write_latent_var_meta(m.vars[v], w);
  w.finish_byte();
}

// metadata/mode.rs:102-167
inline Mode read_mode(BitReader& r, uint8_t format_major, int latent_bits) {
  Mode m;
  uint64_t variant = r.read_uint(BITS_TO_ENCODE_MODE_VARIANT);
  switch (variant) {
    case 0: m.kind = kClassic; break;
    case 1:
      if (format_major == 0) fail(kCorruption, "unable to decompress data from yanked v0.0.0 of pco with different GCD encoding");
      m.kind = kIntMult; m.base_latent = r.read_uint((Bitlen)latent_bits); break;
    case 2: m.kind = kFloatMult; m.base_latent = r.read_uint((Bitlen)latent_bits); break;
    case 3: m.kind = kFloatQuant; m.k = (Bitlen)r.read_uint(BITS_TO_ENCODE_QUANTIZE_K); break;
    case 4: {
      m.kind = kDict;
      size_t n_unique = (size_t)r.read_uint(BITS_TO_ENCODE_DICT_LEN);
      r.drain_empty_byte("expected zeros between dict mode length and values");
      r.check_in_bounds();
      m.dict.assign(n_unique, 0);
      for (size_t start = 0; start < n_unique; start += 512) {  // dyn_latents.rs:30-51
        size_t end = std::min(start + 512, n_unique);
        for (size_t i = start; i < end; i++) m.dict[i] = r.read_uint((Bitlen)latent_bits);
        r.check_in_bounds();
      }
      return m;
    }
    default: fail(kCorruption, "unknown mode variant");
  }
  r.check_in_bounds();
  return m;
}
// metadata/delta_encoding.rs:118-202
inline DeltaEncoding read_delta_encoding(BitReader& r, uint8_t format_major) {
  DeltaEncoding d;
  if (format_major < 3) {  // read_from_pre_v3
    size_t order = (size_t)r.read_uint(BITS_TO_ENCODE_DELTA_ENCODING_ORDER);
    if (order != 0) { d.kind = kDeltaConsecutive; d.order = order; }
    r.check_in_bounds();
    return d;
  }
  uint64_t variant = r.read_uint(BITS_TO_ENCODE_DELTA_ENCODING_VARIANT);
  switch (variant) {
    case 0: break;
    case 1: {
      size_t order = (size_t)r.read_uint(BITS_TO_ENCODE_DELTA_ENCODING_ORDER);
      if (order == 0) fail(kCorruption, "Consecutive delta encoding order must not be 0");
      d.kind = kDeltaConsecutive; d.order = order; d.secondary_uses_delta = r.read_bool(); break;
    }
    case 2: {
      Bitlen window_n_log = 1 + (Bitlen)r.read_uint(BITS_TO_ENCODE_DELTA_LOOKBACK_WINDOW_N_LOG);
      Bitlen state_n_log = (Bitlen)r.read_uint(BITS_TO_ENCODE_DELTA_LOOKBACK_STATE_N_LOG);
      if (window_n_log > MAX_DELTA_LOOKBACK_WINDOW_N_LOG) fail(kCorruption, "LZ delta encoding window size log exceeds max");
      if (state_n_log > window_n_log) fail(kCorruption, "LZ delta encoding state size log exceeded window size log");
      d.kind = kDeltaLookback; d.window_n_log = window_n_log; d.state_n_log = state_n_log;
      d.secondary_uses_delta = r.read_bool(); break;
    }
    case 3: {
      d.kind = kDeltaConv1;
      d.quantization = (Bitlen)r.read_uint(BITS_TO_ENCODE_DELTA_CONV_QUANTIZATION);
      d.bias = (int64_t)(r.read_uint(64) ^ ((uint64_t)1 << 63));
      size_t order = 1 + (size_t)r.read_uint(BITS_TO_ENCODE_DELTA_CONV_N_WEIGHTS);
      for (size_t i = 0; i < order; i++) d.weights.push_back((int64_t)(int32_t)((uint32_t)r.read_uint(32) ^ 0x80000000u));
      break;
    }
    default: fail(kCorruption, "unknown delta encoding value");
  }
  r.check_in_bounds();
  return d;
}
// metadata/chunk_latent_var.rs:21-53,102-156
inline LatentVarMeta read_latent_var_meta(BitReader& r, int latent_bits) {
  LatentVarMeta v; v.present = true; v.latent_bits = latent_bits;
  v.ans_size_log = (Bitlen)r.read_uint(BITS_TO_ENCODE_ANS_SIZE_LOG);
  size_t n_bins = (size_t)r.read_uint(BITS_TO_ENCODE_N_BINS);
  r.check_in_bounds();
  if (((size_t)1 << v.ans_size_log) < n_bins) fail(kCorruption, "ANS size log is too small for number of bins");
  if (n_bins == 1 && v.ans_size_log > 0) fail(kCorruption, "Only 1 bin but ANS size log is > 0");
  if (v.ans_size_log > MAX_ANS_BITS) fail(kCorruption, "ANS size log too large");
  Bitlen obb = bits_to_encode_offset_bits_rt(latent_bits);
  v.bins.reserve(n_bins);
  for (size_t start = 0; start < n_bins; start += 128) {
    size_t end = std::min(start + 128, n_bins);
    for (size_t i = start; i < end; i++) {
      DynBin b;
      b.weight = (uint32_t)r.read_uint(v.ans_size_log) + 1;
      b.lower = r.read_uint((Bitlen)latent_bits);
      b.offset_bits = (Bitlen)r.read_uint(obb);
      if (b.offset_bits > (Bitlen)latent_bits) { r.check_in_bounds(); fail(kCorruption, "offset bits exceeds type"); }
      v.bins.push_back(b);
    }
    r.check_in_bounds();
  }
  return v;
}
// metadata/chunk.rs:33-103 (validation), :127-174 (read)
inline void validate_chunk_meta(const ChunkMeta& m) {
  if (m.delta.kind == kDeltaLookback) {
    uint64_t window_n = (uint64_t)1 << m.delta.window_n_log;
    for (const DynBin& b : m.vars[kVarDelta].bins)
      if (b.lower < 1 || b.lower > window_n) fail(kCorruption, "delta lookback bin had invalid lower bound");
  } else if (m.delta.kind == kDeltaConv1) {  // chunk.rs:58-94: the primary latent type bounds the Conv type (data_types/unsigned.rs:132-138)
    const int l_bits = m.vars[kVarPrimary].latent_bits;
    if (l_bits > 32) fail(kCorruption, "Conv1 delta encodings are not supported on types larger than 32 bits");
    const int conv_bits = l_bits == 32 ? 64 : 2 * l_bits;
    const Bitlen max_quantization = std::min<Bitlen>((1u << BITS_TO_ENCODE_DELTA_CONV_QUANTIZATION) - 1, (Bitlen)(conv_bits - 1));
    if (m.delta.quantization > max_quantization) fail(kCorruption, "Conv1 delta encoding quantization exceeds max");
    double sum_abs = 0.0;   // (i64::abs as f64, summed in order)
    for (int64_t wgt : m.delta.weights) sum_abs += (double)(wgt < 0 ? (uint64_t)0 - (uint64_t)wgt : (uint64_t)wgt);
    const double bias_abs = std::fabs((double)m.delta.bias);
    const double max_pred = bias_abs + std::ldexp(1.0, l_bits) * sum_abs;
    if (max_pred >= std::ldexp(1.0, conv_bits - 1)) fail(kCorruption, "Conv1 delta encoding weights and bias risk overflowing");
  }
}
inline ChunkMeta read_chunk_meta(BitReader& r, uint8_t format_major, int latent_bits) {
  ChunkMeta m;
  m.mode = read_mode(r, format_major, latent_bits);
  m.delta = read_delta_encoding(r, format_major);
  if (m.delta.kind == kDeltaLookback) m.vars[kVarDelta] = read_latent_var_meta(r, 32);
  int primary_bits = m.mode.kind == kDict ? 32 : latent_bits;  // mode.rs:197-202
  m.vars[kVarPrimary] = read_latent_var_meta(r, primary_bits);
  if (m.mode.kind == kIntMult || m.mode.kind == kFloatMult || m.mode.kind == kFloatQuant)
    m.vars[kVarSecondary] = read_latent_var_meta(r, latent_bits);
  r.drain_empty_byte("nonzero bits in end of final byte of chunk metadata");
  validate_chunk_meta(m);
  return m;
}

// mode validity (data_types/unsigned.rs:65-71, float.rs:377-390)
inline bool mode_is_valid(const Mode& m, uint8_t dtype) {
  NumKind k = dtype_kind(dtype); int bits = dtype_bits(dtype);
  switch (m.kind) {
    case kClassic: case kDict: return true;
    case kIntMult: return k != kFloat && m.base_latent > 0;
    case kFloatQuant: {
      if (k != kFloat) return false;
      Bitlen prec = bits == 64 ? 52 : bits == 32 ? 23 : 10;
      return m.k > 0 && m.k <= prec;
    }
    case kFloatMult: {
      if (k != kFloat) return false;
      if (bits == 64) { double b = float_from_latent_ordered<uint64_t>(m.base_latent); return std::isfinite(b) && std::fabs(b) > 0.0; }
      if (bits == 32) { float b = float_from_latent_ordered<uint32_t>((uint32_t)m.base_latent); return std::isfinite(b) && std::fabs(b) > 0.0f; }
      // f16: finite and nonzero by bit inspection
      uint16_t hb = from_latent_ordered<uint16_t>((uint16_t)m.base_latent, kFloat);
      return ((hb >> 10) & 0x1f) != 0x1f && (hb & 0x7fff) != 0;
    }
  }
  return false;
}

// ----------------------------------------------------------------------------
// tANS (ans/spec.rs, ans/encoding.rs, ans/decoding.rs)
// ----------------------------------------------------------------------------
inline uint32_t choose_stride(uint32_t table_size) {  // ans/spec.rs:24-30
  uint32_t res = (3 * table_size) / 5;
  if (res % 2 == 0) res += 1;
  return res;
}
// ans/spec.rs:37-59 (+ from_weights :61-75: empty weights -> [1])
inline std::vector<uint32_t> spread_state_symbols(Bitlen size_log, const std::vector<uint32_t>& weights_in) {
  std::vector<uint32_t> w1{1};
  const std::vector<uint32_t>& weights = weights_in.empty() ? w1 : weights_in;
  uint64_t table_size = 0;
  for (uint32_t w : weights) table_size += w;
  if (table_size != ((uint64_t)1 << size_log)) fail(kCorruption, "table size log does not agree with total weight");
  std::vector<uint32_t> res((size_t)table_size, 0);
  uint32_t step = 0, stride = choose_stride((uint32_t)table_size);
  uint32_t mod_table_size = (0xFFFFFFFFu >> 1) >> (32 - 1 - size_log);
  for (size_t s = 0; s < weights.size(); s++)
    for (uint32_t k = 0; k < weights[s]; k++) { res[(stride * step) & mod_table_size] = (uint32_t)s; step++; }
  return res;
}
struct AnsNode { uint16_t next_state_idx_base; uint8_t offset_bits; uint8_t bits_to_read; };  // ans/decoding.rs:15-19
// ans/decoding.rs:27-47
inline std::vector<AnsNode> build_decoder_nodes(Bitlen size_log, const std::vector<uint32_t>& weights_in,
                                                const std::vector<uint32_t>& state_symbols,
                                                const std::vector<Bitlen>& bin_offset_bits) {
  std::vector<uint32_t> symbol_x_s = weights_in.empty() ? std::vector<uint32_t>{1} : weights_in;
  uint32_t table_size = 1u << size_log;
  std::vector<AnsNode> nodes; nodes.reserve(table_size);
  for (uint32_t symbol : state_symbols) {
    uint32_t base = symbol_x_s[symbol];
    Bitlen bits_to_read = clz32(base) - clz32(table_size);
    base <<= bits_to_read;
    Bitlen ob = symbol < bin_offset_bits.size() ? bin_offset_bits[symbol] : 0;
    nodes.push_back(AnsNode{(uint16_t)(base - table_size), (uint8_t)ob, (uint8_t)bits_to_read});
    symbol_x_s[symbol]++;
  }
  return nodes;
}
// ans/encoding.rs:8-91
struct AnsEncoder {
  struct SymbolInfo { uint32_t renorm_bit_cutoff; Bitlen min_renorm_bits; uint32_t weight; uint32_t ns_off; };
  std::vector<SymbolInfo> infos; std::vector<uint32_t> next_states; Bitlen size_log = 0;
  void init(Bitlen size_log_, const std::vector<uint32_t>& weights_in, const std::vector<uint32_t>& state_symbols) {
    std::vector<uint32_t> weights = weights_in.empty() ? std::vector<uint32_t>{1} : weights_in;
    size_log = size_log_; uint32_t table_size = 1u << size_log;
    infos.resize(weights.size()); uint32_t off = 0;
    for (size_t s = 0; s < weights.size(); s++) {
      uint32_t weight = weights[s];
      uint32_t max_x_s = 2 * weight - 1;
      Bitlen min_renorm_bits = size_log - (31 - clz32(max_x_s));
      infos[s] = SymbolInfo{(uint32_t)(2 * weight * (1u << min_renorm_bits)), min_renorm_bits, weight, off};
      off += weight;
    }
    next_states.assign(table_size, 0);
    std::vector<uint32_t> fill(weights.size(), 0);
    for (uint32_t state_idx = 0; state_idx < table_size; state_idx++) {
      uint32_t s = state_symbols[state_idx];
      next_states[infos[s].ns_off + fill[s]++] = table_size + state_idx;
    }
  }
  inline void encode(uint32_t state, uint32_t symbol, uint32_t& new_state, Bitlen& bits) const {
    const SymbolInfo& si = infos[symbol];
    bits = state >= si.renorm_bit_cutoff ? si.min_renorm_bits + 1 : si.min_renorm_bits;
    new_state = next_states[si.ns_off + ((state >> bits) - si.weight)];
  }
  uint32_t default_state() const { return 1u << size_log; }
};

// Rust `f32::round() as u32`: round half away from zero, saturating, NaN -> 0
inline uint32_t f32_round_to_u32_sat(float x) {
  float r = ::roundf(x);
  if (!(r == r)) return 0;
  if (r <= 0.0f) return 0;
  if (r >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t)r;
}
// ans/encoding.rs:95-151
inline std::vector<uint32_t> quantize_weights_to(const std::vector<uint32_t>& counts, size_t total_count, Bitlen size_log) {
  if (size_log == 0) return {1};
  uint32_t required_weight_sum = 1u << size_log;
  float multiplier = (float)required_weight_sum / (float)total_count;
  std::vector<float> desired(counts.size());
  float desired_surplus = 0.0f;
  for (size_t i = 0; i < counts.size(); i++) {
    float v = (float)counts[i] * multiplier - 1.0f;
    desired[i] = v > 0.0f ? v : 0.0f;  // f32::max(0.0) (NaN-free here)
  }
  for (float d : desired) desired_surplus += d;
  uint32_t required_surplus = required_weight_sum - (uint32_t)counts.size();
  float surplus_mult = desired_surplus == 0.0f ? 0.0f : (float)required_surplus / desired_surplus;
  std::vector<float> float_weights(counts.size());
  std::vector<uint32_t> weights(counts.size());
  uint32_t weight_sum = 0;
  for (size_t i = 0; i < counts.size(); i++) {
    float_weights[i] = 1.0f + desired[i] * surplus_mult;
    weights[i] = f32_round_to_u32_sat(float_weights[i]);
    weight_sum += weights[i];
  }
  size_t i = 0;
  while (weight_sum > required_weight_sum) {
    if (i >= weights.size()) fail(kInvalidArgument, "quantize_weights_to: index out of bounds (reference would panic)");
    if (weights[i] > 1 && (float)weights[i] > float_weights[i]) { weights[i]--; weight_sum--; }
    i++;
  }
  i = 0;
  while (weight_sum < required_weight_sum) {
    if (i >= weights.size()) fail(kInvalidArgument, "quantize_weights_to: index out of bounds (reference would panic)");
    if ((float)weights[i] < float_weights[i]) { weights[i]++; weight_sum++; }
    i++;
  }
  return weights;
}
// ans/encoding.rs:156-175
inline std::pair<Bitlen, std::vector<uint32_t>> quantize_weights(const std::vector<uint32_t>& counts, size_t total_count, Bitlen max_size_log) {
  if (counts.size() == 1) return {0, {1}};
  if (counts.empty()) fail(kInvalidArgument, "quantize_weights: no counts (reference would panic)");
  Bitlen min_size_log = 64 - (Bitlen)__builtin_clzll((unsigned long long)(counts.size() - 1));  // size >= 2 here
  Bitlen size_log = std::max(min_size_log, max_size_log);
  std::vector<uint32_t> weights = quantize_weights_to(counts, total_count, size_log);
  Bitlen power_of_2 = 32;
  for (uint32_t w : weights) power_of_2 = std::min(power_of_2, w == 0 ? (Bitlen)32 : (Bitlen)__builtin_ctz(w));
  size_log -= power_of_2;
  for (uint32_t& w : weights) w >>= power_of_2;
  return {size_log, weights};
}

}  // namespace pco_oracle
