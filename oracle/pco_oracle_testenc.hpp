// =============================================================================
// pco_oracle_testenc.hpp -- ORACLE (test infrastructure): a TEST-ONLY stream generator.
//
// NOT a restatement of the reference's encoder.  The reference can write chunks the restated
// encoder (pco_oracle_encode.hpp) does not: Dict mode (mode/dict.rs:12-33), Conv1 delta
// (delta/conv1.rs:424-461), and the format allows more than any encoder writes (a secondary
// variable that is delta'd too, lookback state_n_log > 0: metadata/delta_encoding.rs:86-99).
// Decoding is deterministic, so ANY valid stream exercises the decoders; this file writes such
// streams from a caller-chosen spec (dictionary order, lookbacks, Conv1 weights) with the format
// writers of pco_oracle.hpp and the restated bin training / tANS machinery.  The decode side
// they are checked against (pco_oracle_decode.hpp) IS a restatement and is pinned by the
// reference's own v1_0_0_dict.pco / v1_0_0_conv1.pco assets.
// Only tests/ may reach this (through pco_oracle_capi.cpp); the product never does.
// =============================================================================
#pragma once
#include "pco_oracle_encode.hpp"

namespace pco_oracle {

struct TestEncSpec {
  uint32_t mode_kind;           // ModeSpecKind numbering: 1 classic, 2 float mult (mode_f64), 3 float quant (mode_u64), 4 int mult (mode_u64), 5 dict
  uint32_t delta_kind;          // DeltaKind: 0 none, 1 consecutive, 2 lookback, 3 conv1
  double mode_f64;
  uint64_t mode_u64;
  uint32_t order;               // consecutive order / number of Conv1 weights
  uint32_t secondary_uses_delta;
  uint32_t window_n_log, state_n_log;
  uint32_t lookback_seed;       // 0: choose_lookbacks (the reference's search); else random valid lookbacks from this seed
  uint32_t quantization;
  int64_t bias;
  int32_t weights[32];
  uint32_t level;
  uint32_t dict_first_appearance;  // dictionary in first-appearance order instead of sorted
};

// Conv1 residuals in place (delta/conv1.rs:424-461 semantics, :148-161 predict_one): latents[i] -= prediction(latents[i-order..i]), + MID;
// returns the state (the first `order` latents, zero-padded when the page is shorter).  The prediction arithmetic is the decoder's.
template <class P> std::vector<P> test_conv1_encode_in_place(const LatentVarDelta& d, P* latents, size_t len) {
  if (LT<P>::BITS > 32) fail(kInvalidArgument, "Conv1 needs a latent type of at most 32 bits");
  const int conv_bits = LT<P>::BITS == 32 ? 64 : 2 * LT<P>::BITS;
  auto wrap = [&](uint64_t x) -> int64_t { return conv_bits == 64 ? (int64_t)x : (int64_t)(x << ((64 - conv_bits) & 63)) >> ((64 - conv_bits) & 63); };
  const size_t order = d.weights.size();
  std::vector<P> state(order, 0);
  for (size_t i = 0; i < std::min(order, len); i++) state[i] = latents[i];
  for (size_t i = len; i-- > order;) {
    uint64_t sum = (uint64_t)wrap((uint64_t)d.bias);
    for (size_t k = 0; k < order; k++) sum += (uint64_t)wrap((uint64_t)d.weights[k]) * (uint64_t)latents[i - order + k];
    int64_t s = wrap(sum); if (s < 0) s = 0;
    latents[i] = (P)(latents[i] - (P)(uint64_t)(s >> d.quantization) + MID<P>());
  }
  return state;
}

template <class L, class P> struct TestChunk {
  ChunkMeta meta; uint8_t dtype = 0;
  LatentCompressor<uint32_t> dvar; LatentCompressor<P> pvar; LatentCompressor<L> svar;
  struct PageInfo { size_t page_n; PageVarInfo v[3]; };
  std::vector<PageInfo> page_infos;

  void write_page(size_t page_idx, BitWriter& w) const {   // (the layout of wrapped/chunk_compressor.rs:659-705)
    const PageInfo& pi = page_infos[page_idx];
    DissectedVar dd, dp, ds;
    if (dvar.present) dd = dvar.dissect_page(pi.v[0].start, pi.v[0].end);
    dp = pvar.dissect_page(pi.v[1].start, pi.v[1].end);
    if (svar.present) ds = svar.dissect_page(pi.v[2].start, pi.v[2].end);
    auto write_var_meta = [&](const PageVarInfo& v, const DissectedVar& d, int latent_bits, Bitlen ans_size_log, uint32_t default_state) {
      for (uint64_t x : v.delta_state) w.write_uint(x, (Bitlen)latent_bits);
      for (int j = 0; j < 4; j++) w.write_uint(d.ans_final_states[j] - default_state, ans_size_log);
    };
    if (dvar.present) write_var_meta(pi.v[0], dd, 32, dvar.encoder.size_log, dvar.encoder.default_state());
    write_var_meta(pi.v[1], dp, LT<P>::BITS, pvar.encoder.size_log, pvar.encoder.default_state());
    if (svar.present) write_var_meta(pi.v[2], ds, LT<L>::BITS, svar.encoder.size_log, svar.encoder.default_state());
    w.finish_byte();
    for (size_t batch_start = 0; batch_start < pi.page_n; batch_start += FULL_BATCH_N) {
      if (dvar.present) dvar.write_dissected_batch(dd, batch_start, w);
      pvar.write_dissected_batch(dp, batch_start, w);
      if (svar.present) svar.write_dissected_batch(ds, batch_start, w);
    }
    w.finish_byte();
  }
};

// Conv1 delta encoding in the ChunkMeta (metadata/delta_encoding.rs:239-252); write_delta_encoding refuses it (no restated encoder).
inline void test_write_delta_encoding(const DeltaEncoding& d, BitWriter& w) {
  if (d.kind != kDeltaConv1) { write_delta_encoding(d, w); return; }
  w.write_uint(3, BITS_TO_ENCODE_DELTA_ENCODING_VARIANT);
  w.write_uint(d.quantization, BITS_TO_ENCODE_DELTA_CONV_QUANTIZATION);
  w.write_uint((uint64_t)d.bias ^ ((uint64_t)1 << 63), 64);
  w.write_uint(d.weights.size() - 1, BITS_TO_ENCODE_DELTA_CONV_N_WEIGHTS);
  for (int64_t x : d.weights) w.write_uint((uint64_t)((uint32_t)(int32_t)x ^ 0x80000000u), 32);
}

template <class L, class P> void test_build_chunk(TestChunk<L, P>& tc, std::vector<P> primary, std::vector<L> secondary, bool has_secondary,
                                                  const std::vector<size_t>& pages, const Mode& mode, const DeltaEncoding& de, const TestEncSpec& spec, uint8_t dtype) {
  tc = TestChunk<L, P>(); tc.dtype = dtype;
  const size_t n = primary.size();
  const Bitlen ubl = choose_unoptimized_bins_log(spec.level, n);
  std::vector<uint32_t> delta_latents;
  const LatentVarDelta dprim = delta_for_latent_var(de, kVarPrimary), dsec = delta_for_latent_var(de, kVarSecondary);
  Xoroshiro128PlusPlus rng(spec.lookback_seed);
  size_t start_idx = 0;
  for (size_t page_n : pages) {
    const size_t end_idx = start_idx + page_n;
    typename TestChunk<L, P>::PageInfo pi; pi.page_n = page_n;
    std::vector<uint32_t> lbs;
    if (de.kind == kDeltaLookback) {
      const size_t state_n = (size_t)1 << de.state_n_log, window_n = (size_t)1 << de.window_n_log;
      // (lookback.rs:166-185 pads a short page's state at the front and the decoder returns the first n state slots: unwritable)
      if (page_n < state_n) fail(kInvalidArgument, "a page shorter than the lookback state cannot be represented");
      if (spec.lookback_seed == 0) lbs = choose_lookbacks<P>(de.window_n_log, de.state_n_log, primary.data() + start_idx, page_n);
      else if (page_n > state_n) {   // any lookback in [1, min(window_n, i)] is valid for element i (lookback.rs:166-185 reads latents[i - lookback])
        lbs.resize(page_n - state_n);
        for (size_t i = state_n; i < page_n; i++) {
          const uint64_t r = rng.next_u64(); const size_t lim = std::min(window_n, i);
          // a mix of short, repeated and far lookbacks, the window's far end included
          size_t lb = (r & 3) == 0 ? 1 + (r >> 8) % lim : ((r & 3) == 1 ? lim : 1 + (r >> 8) % std::min<size_t>(lim, 7));
          lbs[i - state_n] = (uint32_t)lb;
        }
      }
    }
    auto encode_var = [&](auto& v, const LatentVarDelta& d, PageVarInfo& out) {
      typedef typename std::remove_reference<decltype(v)>::type::value_type V;
      std::vector<V> st;
      if (d.kind == kDeltaConsecutive) st = consecutive_encode_in_place<V>(d.order, v.data() + start_idx, page_n);
      else if (d.kind == kDeltaLookback) st = lookback_encode_in_place<V>(d.state_n_log, lbs.data(), v.data() + start_idx, page_n);
      else if (d.kind == kDeltaConv1) { if constexpr (sizeof(V) <= 4) st = test_conv1_encode_in_place<V>(d, v.data() + start_idx, page_n); else fail(kInvalidArgument, "Conv1 on a 64-bit latent"); }
      for (V x : st) out.delta_state.push_back((uint64_t)x);
      out.start = std::min(start_idx + d.n_latents_per_state(), end_idx); out.end = end_idx;
    };
    encode_var(primary, dprim, pi.v[1]);
    if (has_secondary) encode_var(secondary, dsec, pi.v[2]);
    if (de.kind == kDeltaLookback) { pi.v[0].start = delta_latents.size(); pi.v[0].end = delta_latents.size() + lbs.size(); delta_latents.insert(delta_latents.end(), lbs.begin(), lbs.end()); }
    tc.page_infos.push_back(pi);
    start_idx = end_idx;
  }
  auto contiguous = [&](auto& v, int key) {
    typename std::remove_reference<decltype(v)>::type res;
    for (auto& pi : tc.page_infos) res.insert(res.end(), v.begin() + pi.v[key].start, v.begin() + pi.v[key].end);
    return res;
  };
  tc.meta.mode = mode; tc.meta.delta = de;
  if (de.kind == kDeltaLookback) {
    auto t = train_infos<uint32_t>(contiguous(delta_latents, 0), ubl);
    tc.meta.vars[kVarDelta] = var_meta_from_trained(t);
    tc.dvar.init(t, tc.meta.vars[kVarDelta], std::move(delta_latents));
  }
  {
    auto t = train_infos<P>(contiguous(primary, 1), ubl);
    tc.meta.vars[kVarPrimary] = var_meta_from_trained(t);
    tc.pvar.init(t, tc.meta.vars[kVarPrimary], std::move(primary));
  }
  if (has_secondary) {
    auto t = train_infos<L>(contiguous(secondary, 2), std::min(ubl, LIMITED_UNOPTIMIZED_BINS_LOG));
    tc.meta.vars[kVarSecondary] = var_meta_from_trained(t);
    tc.svar.init(t, tc.meta.vars[kVarSecondary], std::move(secondary));
  }
  validate_chunk_meta(tc.meta);
}

// One standalone file: header | one chunk (one page) per entry of `chunks` | terminator.
template <class L> std::vector<uint8_t> test_encode_file(const L* bits, size_t n, uint8_t dtype, const TestEncSpec& spec, const std::vector<size_t>& chunks) {
  const NumKind kind = dtype_kind(dtype);
  DeltaEncoding de;
  de.kind = (DeltaKind)spec.delta_kind; de.secondary_uses_delta = spec.secondary_uses_delta != 0;
  if (de.kind == kDeltaConsecutive) { de.order = spec.order; if (de.order == 0 || de.order > MAX_CONSECUTIVE_DELTA_ORDER) fail(kInvalidArgument, "consecutive order"); }
  else if (de.kind == kDeltaLookback) {
    de.window_n_log = spec.window_n_log; de.state_n_log = spec.state_n_log;
    if (de.window_n_log < 1 || de.window_n_log > MAX_DELTA_LOOKBACK_WINDOW_N_LOG || de.state_n_log > de.window_n_log) fail(kInvalidArgument, "lookback window / state");
  } else if (de.kind == kDeltaConv1) {
    if (spec.order < 1 || spec.order > 32) fail(kInvalidArgument, "conv1 order");
    de.quantization = spec.quantization; de.bias = spec.bias; de.weights.assign(spec.weights, spec.weights + spec.order); de.secondary_uses_delta = false;
  } else if (de.kind != kDeltaNone) fail(kInvalidArgument, "delta kind");
  BitWriter w;
  write_standalone_header(w, n, 0);
  size_t sum = 0; for (size_t c : chunks) { if (c == 0) fail(kInvalidArgument, "empty chunk"); sum += c; }
  if (sum != n) fail(kInvalidArgument, "chunk sizes do not sum to n");
  size_t start = 0;
  for (size_t cn : chunks) {
    const L* src = bits + start;
    const uint32_t n_m1 = (uint32_t)cn - 1;
    w.write_aligned_bytes(&dtype, 1);
    w.write_uint(n_m1, BITS_TO_ENCODE_N_ENTRIES);
    const std::vector<size_t> pages = {cn};
    if (spec.mode_kind == kModeTryDict) {   // mode/dict.rs:12-33: u32 indices into the dictionary of distinct latents
      std::vector<L> lat(cn);
      for (size_t i = 0; i < cn; i++) lat[i] = to_latent_ordered<L>(src[i], kind);
      std::vector<L> uniq;
      if (spec.dict_first_appearance) { std::vector<L> sorted(lat); std::sort(sorted.begin(), sorted.end()); sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
        std::vector<uint8_t> seen(sorted.size(), 0);
        for (L x : lat) { size_t k = std::lower_bound(sorted.begin(), sorted.end(), x) - sorted.begin(); if (!seen[k]) { seen[k] = 1; uniq.push_back(x); } }
      } else { uniq = lat; std::sort(uniq.begin(), uniq.end()); uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end()); }
      std::vector<std::pair<L, uint32_t>> index; for (size_t k = 0; k < uniq.size(); k++) index.push_back({uniq[k], (uint32_t)k});
      std::sort(index.begin(), index.end());
      std::vector<uint32_t> idx(cn);
      for (size_t i = 0; i < cn; i++) idx[i] = std::lower_bound(index.begin(), index.end(), std::make_pair(lat[i], (uint32_t)0))->second;
      Mode mode; mode.kind = kDict; for (L x : uniq) mode.dict.push_back((uint64_t)x);
      TestChunk<L, uint32_t>* tc = new TestChunk<L, uint32_t>();
      try {
        test_build_chunk<L, uint32_t>(*tc, std::move(idx), {}, false, pages, mode, de, spec, dtype);
        write_mode(tc->meta.mode, LT<L>::BITS, w); test_write_delta_encoding(tc->meta.delta, w);
        for (int v = 0; v < 3; v++) if (tc->meta.vars[v].present) write_latent_var_meta(tc->meta.vars[v], w);
        w.finish_byte();
        tc->write_page(0, w);
      } catch (...) { delete tc; throw; }
      delete tc;
    } else {
      ChunkConfig cfg; cfg.mode_kind = (ModeSpecKind)spec.mode_kind; cfg.mode_f64 = spec.mode_f64; cfg.mode_u64 = spec.mode_u64; cfg.enable_8_bit = true;
      if (cfg.mode_kind == kModeAuto) fail(kInvalidArgument, "the test encoder takes explicit modes");
      Mode mode;
      SplitLatents<L> lat = choose_mode_and_split<L>(src, cn, dtype, cfg, mode);
      TestChunk<L, L>* tc = new TestChunk<L, L>();
      try {
        test_build_chunk<L, L>(*tc, std::move(lat.primary), std::move(lat.secondary), lat.has_secondary, pages, mode, de, spec, dtype);
        write_mode(tc->meta.mode, LT<L>::BITS, w); test_write_delta_encoding(tc->meta.delta, w);
        for (int v = 0; v < 3; v++) if (tc->meta.vars[v].present) write_latent_var_meta(tc->meta.vars[v], w);
        w.finish_byte();
        tc->write_page(0, w);
      } catch (...) { delete tc; throw; }
      delete tc;
    }
    start += cn;
  }
  const uint8_t term = MAGIC_TERMINATION_BYTE;
  w.write_aligned_bytes(&term, 1);
  w.buf.resize(w.byte_len());
  return w.buf;
}

}  // namespace pco_oracle
