// =============================================================================
// pco_oracle_decode.hpp -- ORACLE (test infrastructure), decode side.
// Restates standalone/decompressor.rs, wrapped/{file,chunk,page}_decompressor.rs,
// chunk_latent_decompressor.rs, page_latent_decompressor.rs, delta/*.rs (decode)
// and mode/*.rs (join) of pco v1.0.3.  See pco_oracle.hpp for the rules.
// =============================================================================
#pragma once
#include "pco_oracle.hpp"

namespace pco_oracle {

// per latent variable decoder: ChunkLatentDecompressor + PageLatentDecompressor
// (chunk_latent_decompressor.rs:30-76, page_latent_decompressor.rs:69-86)
template <class L> struct LatentDecoder {
  bool present = false;
  LatentVarDelta delta;
  size_t n_bins = 0;
  size_t bytes_per_offset = 0;
  std::vector<AnsNode> nodes;
  std::vector<L> state_lowers;
  // page state
  uint32_t state_idxs[4] = {0, 0, 0, 0};
  std::vector<L> delta_state; size_t delta_state_pos = 0;
  // scratch
  uint32_t offset_bits_csum[FULL_BATCH_N]; uint32_t offset_bits[FULL_BATCH_N]; L latents[FULL_BATCH_N];

  void init_chunk(const LatentVarMeta& meta, const LatentVarDelta& d) {
    present = true; delta = d; n_bins = meta.bins.size();
    Bitlen max_ob = 0; std::vector<Bitlen> bin_ob; std::vector<uint32_t> weights;
    for (const DynBin& b : meta.bins) { max_ob = std::max(max_ob, b.offset_bits); bin_ob.push_back(b.offset_bits); weights.push_back(b.weight); }
    bytes_per_offset = max_ob == 0 ? 0 : (max_ob + 14) / 8;  // read_write_uint.rs:9-16
    std::vector<uint32_t> state_symbols = spread_state_symbols(meta.ans_size_log, weights);
    state_lowers.resize(state_symbols.size());
    for (size_t i = 0; i < state_symbols.size(); i++)
      state_lowers[i] = state_symbols[i] < meta.bins.size() ? (L)meta.bins[state_symbols[i]].lower : (L)0;
    nodes = build_decoder_nodes(meta.ans_size_log, weights, state_symbols, bin_ob);
    std::memset(offset_bits_csum, 0, sizeof(offset_bits_csum));
    std::memset(offset_bits, 0, sizeof(offset_bits));
    for (size_t i = 0; i < FULL_BATCH_N; i++) latents[i] = 0;
    if (meta.bins.size() == 1) {  // chunk_latent_decompressor.rs:53-62
      uint32_t csum = 0;
      for (size_t i = 0; i < FULL_BATCH_N; i++) {
        offset_bits[i] = meta.bins[0].offset_bits; offset_bits_csum[i] = csum; latents[i] = (L)meta.bins[0].lower;
        csum += meta.bins[0].offset_bits;
      }
    }
  }
  // delta/mod.rs:86-100, lookback.rs:187-197
  void init_page(const uint32_t final_state_idxs[4], const std::vector<L>& stored_state) {
    for (int j = 0; j < 4; j++) state_idxs[j] = final_state_idxs[j];
    if (delta.kind == kDeltaLookback) {
      size_t window_n = (size_t)1 << delta.window_n_log;
      size_t buffer_n = std::max(window_n, FULL_BATCH_N) * 2;
      delta_state.assign(buffer_n, 0);
      std::copy(stored_state.begin(), stored_state.end(), delta_state.begin() + (window_n - stored_state.size()));
      delta_state_pos = window_n;
    } else { delta_state = stored_state; delta_state_pos = 0; }
  }

  // page_latent_decompressor.rs:89-177 (the two variants are semantically identical)
  void read_ans_symbols(BitReader& r, size_t batch_n) {
    uint64_t bit_pos = r.bit_pos; uint32_t offset_bit_idx = 0;
    uint32_t st[4] = {state_idxs[0], state_idxs[1], state_idxs[2], state_idxs[3]};
    const AnsNode* nd = nodes.data(); const L* lowers = state_lowers.data();
    const size_t n_nodes = nodes.size();
    for (size_t i = 0; i < batch_n; i++) {
      size_t j = i % ANS_INTERLEAVING;
      uint32_t s = st[j];
      if (s >= n_nodes) fail(kCorruption, "oracle: ANS state index out of table (reference would read out of bounds)");
      uint64_t packed = r.u64_at((size_t)(bit_pos >> 3));
      const AnsNode node = nd[s];
      Bitlen btr = node.bits_to_read;
      uint32_t ans_val = (uint32_t)(packed >> (bit_pos & 7)) & ((1u << btr) - 1);
      offset_bits_csum[i] = offset_bit_idx; offset_bits[i] = node.offset_bits; latents[i] = lowers[s];
      bit_pos += btr; offset_bit_idx += node.offset_bits;
      st[j] = (uint32_t)node.next_state_idx_base + ans_val;
    }
    r.bit_pos = bit_pos;
    for (int j = 0; j < 4; j++) state_idxs[j] = st[j];
  }
  // page_latent_decompressor.rs:15-44 + bit_reader.rs:30-106
  void read_offsets(BitReader& r, size_t n) {
    const uint64_t base = r.bit_pos;
    for (size_t i = 0; i < n; i++) {
      Bitlen ob = offset_bits[i];
      uint64_t bit_idx = base + offset_bits_csum[i];
      size_t byte = (size_t)(bit_idx >> 3); Bitlen sh = bit_idx & 7;
      uint64_t v = r.u64_at(byte) >> sh;
      if (sh + ob > 64) v |= r.u64_at(byte + 8) << (64 - sh);
      if (ob < 64) v &= ((uint64_t)1 << ob) - 1;
      latents[i] = (L)(latents[i] + (L)v);
    }
    r.bit_pos = base + offset_bits_csum[n - 1] + offset_bits[n - 1];
  }
  // page_latent_decompressor.rs:181-235
  void read_batch_pre_delta(BitReader& r, size_t batch_n) {
    if (batch_n == 0) return;
    if (n_bins > 1) read_ans_symbols(r, batch_n);
    else for (size_t i = 0; i < batch_n; i++) latents[i] = state_lowers[0];
    if (bytes_per_offset != 0) read_offsets(r, batch_n);
  }
  // page_latent_decompressor.rs:237-257, delta/mod.rs:125-159
  void read_batch(BitReader& r, const uint32_t* delta_latents, size_t n_remaining_in_page) {
    size_t nlps = delta.n_latents_per_state();
    size_t n_remaining_pre_delta = n_remaining_in_page > nlps ? n_remaining_in_page - nlps : 0;
    size_t pre_delta_len = std::min(FULL_BATCH_N, n_remaining_pre_delta);
    read_batch_pre_delta(r, pre_delta_len);
    size_t dst_n = std::min(n_remaining_in_page, FULL_BATCH_N);
    switch (delta.kind) {
      case kDeltaNone: break;
      case kDeltaConsecutive: {  // delta/consecutive.rs:35-50
        for (size_t i = 0; i < dst_n; i++) latents[i] = (L)(latents[i] + MID<L>());
        for (size_t m = delta_state.size(); m-- > 0;) {
          L moment = delta_state[m];
          for (size_t i = 0; i < dst_n; i++) { L tmp = latents[i]; latents[i] = moment; moment = (L)(moment + tmp); }
          delta_state[m] = moment;
        }
        break;
      }
      case kDeltaLookback: {  // delta/lookback.rs:200-246
        for (size_t i = 0; i < dst_n; i++) latents[i] = (L)(latents[i] + MID<L>());
        size_t window_n = (size_t)1 << delta.window_n_log, state_n = (size_t)1 << delta.state_n_log;
        size_t start_pos = delta_state_pos;
        if (start_pos + dst_n > delta_state.size()) {
          std::copy(delta_state.begin() + (start_pos - window_n), delta_state.begin() + start_pos, delta_state.begin());
          start_pos = window_n;
        }
        bool oob = false;
        // the reference zips with the delta variable's full 256-entry scratch
        for (size_t i = 0; i < dst_n; i++) {
          size_t pos = start_pos + i; uint32_t lb = delta_latents[i]; size_t lookback;
          if (lb <= (uint32_t)window_n) lookback = lb; else { oob = true; lookback = 1; }
          delta_state[pos] = (L)(latents[i] + delta_state[pos - lookback]);
        }
        size_t end_pos = start_pos + dst_n;
        for (size_t i = 0; i < dst_n; i++) latents[i] = delta_state[start_pos - state_n + i];
        delta_state_pos = end_pos;
        if (oob) fail(kCorruption, "delta lookback exceeded window n");
        break;
      }
      case kDeltaConv1: {  // delta/conv1.rs:463-484 (decode_in_place), :148-161 (predict_one), :231-251 (decode_residuals)
        // Conv = i16 / i32 / i64 for 8- / 16- / 32-bit latents (data_types/unsigned.rs:132-134); the sums are formed in 64 bits
        // and wrapped to the Conv width, which is what wrapping arithmetic in the narrower type yields
        if (LT<L>::BITS > 32) fail(kCorruption, "Conv1 delta encoding cannot be used with 64-bit latents");
        const int conv_bits = LT<L>::BITS == 32 ? 64 : 2 * LT<L>::BITS;
        auto wrap = [&](uint64_t x) -> int64_t { return conv_bits == 64 ? (int64_t)x : (int64_t)(x << (64 - conv_bits)) >> (64 - conv_bits); };
        const size_t order = delta.weights.size();
        for (size_t i = 0; i < dst_n; i++) latents[i] = (L)(latents[i] + MID<L>());
        std::vector<L> residuals(dst_n + order);
        for (size_t i = 0; i < order; i++) residuals[i] = delta_state[i];
        for (size_t i = 0; i < dst_n; i++) residuals[order + i] = latents[i];
        for (size_t i = order; i < residuals.size(); i++) {
          uint64_t sum = (uint64_t)wrap((uint64_t)delta.bias);
          for (size_t k = 0; k < order; k++) sum += (uint64_t)wrap((uint64_t)delta.weights[k]) * (uint64_t)(int64_t)(uint64_t)residuals[i - order + k];
          int64_t sconv = wrap(sum);
          if (sconv < 0) sconv = 0;
          residuals[i] = (L)(residuals[i] + (L)(uint64_t)(sconv >> delta.quantization));
        }
        for (size_t i = 0; i < dst_n; i++) latents[i] = residuals[i];
        for (size_t i = 0; i < order; i++) delta_state[i] = residuals[dst_n + i];
        break;
      }
      default: fail(kCorruption, "unknown delta encoding");
    }
  }
};

// join (mode/classic.rs:14-24, int_mult.rs:38-54, float_mult.rs:17-36, float_quant.rs:13-39)
template <class L> void join_latents(const Mode& mode, NumKind kind, const L* primary, const L* secondary, L* dst_bits, size_t n) {
  switch (mode.kind) {
    case kClassic:
      for (size_t i = 0; i < n; i++) dst_bits[i] = from_latent_ordered<L>(primary[i], kind);
      break;
    case kIntMult: {
      L base = (L)mode.base_latent;
      for (size_t i = 0; i < n; i++) dst_bits[i] = from_latent_ordered<L>((L)((L)(primary[i] * base) + secondary[i]), kind);
      break;
    }
    case kFloatQuant: {
      Bitlen k = mode.k;
      L sign_cutoff = (L)(MID<L>() >> k);
      L lowest_k_bits_max = (L)(((L)1 << k) - 1);
      for (size_t i = 0; i < n; i++) {
        L y = primary[i], m = secondary[i];
        L low = y >= sign_cutoff ? m : (L)(lowest_k_bits_max - m);
        dst_bits[i] = from_latent_ordered<L>((L)((L)(y << k) + low), kFloat);
      }
      break;
    }
    case kFloatMult: {
      if constexpr (LT<L>::BITS >= 16) {
        typedef FloatOps<L> FO; typedef typename FO::F F;
        F base = float_from_latent_ordered<L>((L)mode.base_latent);
        for (size_t i = 0; i < n; i++) {
          F unadjusted = int_float_from_latent<L>(primary[i]) * base;
          L l = (L)(float_to_latent_ordered<L>(unadjusted) + secondary[i] + MID<L>());
          dst_bits[i] = from_latent_ordered<L>(l, kFloat);
        }
      } else fail(kCorruption, "float mult on an 8-bit type");
      break;
    }
    default: fail(kCorruption, "dict mode is joined by the chunk decoder");
  }
}

// One chunk's decoder: wrapped::ChunkDecompressor + PageDecompressorState
// (wrapped/chunk_decompressor.rs:46-80, wrapped/page_decompressor.rs:36-191)
template <class L> struct ChunkDecoder {
  ChunkMeta meta; NumKind kind;
  LatentDecoder<uint32_t> dvar; LatentDecoder<L> pvar, svar;
  LatentDecoder<uint32_t> pdict;   // Dict mode: the primary variable holds u32 dictionary indices (mode.rs:197-202)
  bool dict = false;
  size_t n_remaining = 0;

  void init(const ChunkMeta& m, uint8_t dtype) {
    meta = m; kind = dtype_kind(dtype);
    if (!mode_is_valid(m.mode, dtype)) fail(kCorruption, "invalid mode for number type");
    dict = m.mode.kind == kDict;
    if (m.vars[kVarDelta].present) dvar.init_chunk(m.vars[kVarDelta], delta_for_latent_var(m.delta, kVarDelta));
    if (dict) pdict.init_chunk(m.vars[kVarPrimary], delta_for_latent_var(m.delta, kVarPrimary));
    else pvar.init_chunk(m.vars[kVarPrimary], delta_for_latent_var(m.delta, kVarPrimary));
    if (m.vars[kVarSecondary].present) svar.init_chunk(m.vars[kVarSecondary], delta_for_latent_var(m.delta, kVarSecondary));
  }
  size_t n_latents_per_delta_state() const { return delta_for_latent_var(meta.delta, kVarPrimary).n_latents_per_state(); }

  template <class LL> static void read_page_var_meta(BitReader& r, LatentDecoder<LL>& d, Bitlen ans_size_log) {
    // metadata/page_latent_var.rs:28-49
    size_t nlps = d.delta.n_latents_per_state();
    std::vector<LL> st(nlps);
    for (size_t i = 0; i < nlps; i++) st[i] = (LL)r.read_uint(LT<LL>::BITS);
    uint32_t fs[4];
    for (int j = 0; j < 4; j++) fs[j] = (uint32_t)r.read_uint(ans_size_log);
    d.init_page(fs, st);
  }
  // wrapped/page_decompressor.rs:72-90 (+ make_latent_decompressors :36-70)
  void start_page(BitReader& r, size_t n) {
    if (dvar.present) read_page_var_meta(r, dvar, meta.vars[kVarDelta].ans_size_log);
    if (dict) read_page_var_meta(r, pdict, meta.vars[kVarPrimary].ans_size_log);
    else read_page_var_meta(r, pvar, meta.vars[kVarPrimary].ans_size_log);
    if (svar.present) read_page_var_meta(r, svar, meta.vars[kVarSecondary].ans_size_log);
    r.drain_empty_byte("non-zero bits at end of data page metadata");
    r.check_in_bounds();
    size_t nlps = n_latents_per_delta_state();
    size_t n_in_body = n > nlps ? n - nlps : 0;
    if (n_in_body > 0) {
      if ((dvar.present && dvar.n_bins == 0) || (dict ? pdict.n_bins : pvar.n_bins) == 0 || (svar.present && svar.n_bins == 0))
        fail(kCorruption, "unable to decompress chunk with no bins");
    }
    n_remaining = n;
  }
  // wrapped/page_decompressor.rs:115-191
  void read_batch(BitReader& r, L* dst_bits, size_t batch_n) {
    if (dvar.present) {
      size_t nlps = n_latents_per_delta_state();
      size_t limit = std::min(n_remaining > nlps ? n_remaining - nlps : 0, batch_n);
      dvar.read_batch_pre_delta(r, limit);
      r.check_in_bounds();
    }
    if (dict) {  // mode/dict.rs:70-90 (join_latents)
      pdict.read_batch(r, dvar.present ? dvar.latents : nullptr, n_remaining);
      r.check_in_bounds();
      for (size_t i = 0; i < batch_n; i++) if (pdict.latents[i] >= meta.mode.dict.size()) fail(kCorruption, "dict index exceeded dict length");
      for (size_t i = 0; i < batch_n; i++) dst_bits[i] = from_latent_ordered<L>((L)meta.mode.dict[pdict.latents[i]], kind);
      n_remaining -= batch_n;
      if (n_remaining == 0) { r.drain_empty_byte("expected trailing bits at end of page to be empty"); r.check_in_bounds(); }
      return;
    }
    pvar.read_batch(r, dvar.present ? dvar.latents : nullptr, n_remaining);
    r.check_in_bounds();
    if (svar.present) { svar.read_batch(r, dvar.present ? dvar.latents : nullptr, n_remaining); r.check_in_bounds(); }
    join_latents<L>(meta.mode, kind, pvar.latents, svar.present ? svar.latents : nullptr, dst_bits, batch_n);
    n_remaining -= batch_n;
    if (n_remaining == 0) { r.drain_empty_byte("expected trailing bits at end of page to be empty"); r.check_in_bounds(); }
  }
};

// standalone file header (standalone/decompressor.rs:85-137, metadata/format_version.rs:54-78)
struct FileHeader { size_t standalone_version; uint8_t uniform_type; uint64_t n_hint; uint8_t fmt_major, fmt_minor; };
inline FileHeader read_file_header(BitReader& r) {
  FileHeader h{};
  const uint8_t* magic = r.read_aligned_bytes(4);
  r.check_in_bounds();
  if (std::memcmp(magic, MAGIC_HEADER, 4) != 0) fail(kCorruption, "magic header does not match");
  h.standalone_version = (size_t)r.read_uint(BITS_TO_ENCODE_STANDALONE_VERSION);
  if (h.standalone_version < 2) {
    r.bit_pos -= BITS_TO_ENCODE_STANDALONE_VERSION;
  } else {
    if (h.standalone_version >= 3) {
      uint8_t byte = r.read_aligned_bytes(1)[0];
      if (byte != MAGIC_TERMINATION_BYTE) {
        if (!dtype_valid(byte)) fail(kCorruption, "unknown number type byte");
        h.uniform_type = byte;
      }
    }
    Bitlen power = 1 + (Bitlen)r.read_uint(BITS_TO_ENCODE_VARINT_POWER);
    h.n_hint = r.read_uint(power);
    r.drain_empty_byte("standalone size hint");
  }
  r.check_in_bounds();
  if (h.standalone_version > CURRENT_STANDALONE_VERSION) fail(kCorruption, "file's standalone version exceeds max supported");
  // wrapped header
  h.fmt_major = r.read_aligned_bytes(1)[0];
  h.fmt_minor = h.fmt_major >= 4 ? r.read_aligned_bytes(1)[0] : 0;
  if (h.fmt_major > FORMAT_MAJOR) fail(kCorruption, "file's format version definitely cannot be decompressed");
  r.check_in_bounds();
  return h;
}

// standalone::simple_decompress (standalone/simple.rs:149-152, decompressor.rs:190-284)
template <class L> std::vector<L> simple_decompress_t(const uint8_t* src, size_t len, uint8_t dtype) {
  std::vector<uint8_t> padded(len + MAX_BATCH_LATENT_VAR_SIZE + 64, 0);
  if (len) std::memcpy(padded.data(), src, len);
  BitReader r{padded.data(), len * 8, padded.size(), 0};
  FileHeader h = read_file_header(r);
  std::vector<L> res;
  for (;;) {
    // chunk_preamble (decompressor.rs:190-231)
    uint8_t tb = r.read_aligned_bytes(1)[0];
    r.check_in_bounds();
    if (tb == MAGIC_TERMINATION_BYTE) break;
    if (h.uniform_type && h.uniform_type != tb) fail(kCorruption, "chunk's number type does not match file's uniform number type");
    if (tb != dtype) fail(kCorruption, "requested chunk decompression does not match chunk's number type");
    size_t n = (size_t)r.read_uint(BITS_TO_ENCODE_N_ENTRIES) + 1;
    r.check_in_bounds();
    ChunkMeta meta = read_chunk_meta(r, h.fmt_major, LT<L>::BITS);
    ChunkDecoder<L>* cd = new ChunkDecoder<L>();
    try {
      cd->init(meta, dtype);
      cd->start_page(r, n);
      size_t base = res.size(); res.resize(base + n);
      size_t done = 0;
      while (done < n) {
        size_t bn = std::min(FULL_BATCH_N, n - done);
        cd->read_batch(r, res.data() + base + done, bn);
        done += bn;
      }
    } catch (...) { delete cd; throw; }
    delete cd;
  }
  return res;
}

}  // namespace pco_oracle
