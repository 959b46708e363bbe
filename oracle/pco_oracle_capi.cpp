// =============================================================================
// pco_oracle_capi.cpp -- C entry points of the ORACLE (test infrastructure).
// Loaded with ctypes by tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg ONLY.  Never linked into the product library.
// =============================================================================
#include "pco_oracle.hpp"
#include "pco_oracle_decode.hpp"
#include "pco_oracle_encode.hpp"
#include "pco_oracle_testenc.hpp"
#include <memory>

using namespace pco_oracle;

static thread_local std::string g_last_error;

template <class F> static int guard(F&& f) {
  try { f(); return 0; }
  catch (const PcoErr& e) { g_last_error = e.msg; return (int)e.kind; }
  catch (const std::exception& e) { g_last_error = e.what(); return 99; }
}
template <class F> static void dispatch_bits(int bits, F&& f) {
  switch (bits) {
    case 8: f(uint8_t{}); break;
    case 16: f(uint16_t{}); break;
    case 32: f(uint32_t{}); break;
    case 64: f(uint64_t{}); break;
    default: fail(kInvalidArgument, "invalid dtype / latent bits");
  }
}


extern "C" {

typedef struct PcoOracleConfig {
  uint32_t compression_level;
  uint32_t mode_kind;    // ModeSpecKind
  double mode_f64;       // TryFloatMult base
  uint64_t mode_u64;     // TryIntMult base / TryFloatQuant k
  uint32_t delta_kind;   // DeltaSpecKind
  uint32_t delta_order;  // TryConsecutive order
  uint64_t max_page_n;   // 0 => 2^18
  uint32_t enable_8_bit;
  uint32_t reserved;
} PcoOracleConfig;

const char* pco_oracle_last_error() { return g_last_error.c_str(); }

static ChunkConfig to_cfg(const PcoOracleConfig* c) {
  ChunkConfig cfg;
  if (!c) { cfg.enable_8_bit = true; return cfg; }
  cfg.compression_level = c->compression_level;
  cfg.mode_kind = (ModeSpecKind)c->mode_kind; cfg.mode_f64 = c->mode_f64; cfg.mode_u64 = c->mode_u64;
  cfg.delta_kind = (DeltaSpecKind)c->delta_kind; cfg.delta_order = c->delta_order;
  cfg.max_page_n = c->max_page_n == 0 ? DEFAULT_MAX_PAGE_N : (size_t)c->max_page_n;
  cfg.enable_8_bit = c->enable_8_bit != 0;
  return cfg;
}

size_t pco_oracle_file_size_bound(size_t n, uint8_t dtype, uint64_t max_page_n) {
  if (!dtype_valid(dtype)) return 0;
  try {
    ChunkConfig c; c.max_page_n = max_page_n == 0 ? DEFAULT_MAX_PAGE_N : (size_t)max_page_n;
    return standalone_file_size(dtype_bits(dtype), n, c);
  } catch (...) { return 0; }
}

int pco_oracle_simple_compress(const void* nums, size_t n, uint8_t dtype, const PcoOracleConfig* config,
                               int uniform_type, uint8_t* dst, size_t dst_cap, size_t* n_written) {
  return guard([&] {
    if (!dtype_valid(dtype)) fail(kInvalidArgument, "invalid dtype");
    ChunkConfig cfg = to_cfg(config);
    std::vector<uint8_t> out;
    dispatch_bits(dtype_bits(dtype), [&](auto tag) {
      typedef decltype(tag) LTYPE;
      out = simple_compress_t<LTYPE>((const LTYPE*)nums, n, dtype, cfg, uniform_type != 0);
    });
    if (out.size() > dst_cap) fail(kInvalidArgument, "destination too small");
    std::memcpy(dst, out.data(), out.size());
    *n_written = out.size();
  });
}

// simple_compress under PagingSpec::Exact (standalone/simple.rs:32-45: one chunk per entry of the paging spec)
int pco_oracle_simple_compress_exact(const void* nums, size_t n, uint8_t dtype, const PcoOracleConfig* config, int uniform_type,
                                     const size_t* exact_pages, size_t n_exact, uint8_t* dst, size_t dst_cap, size_t* n_written) {
  return guard([&] {
    if (!dtype_valid(dtype)) fail(kInvalidArgument, "invalid dtype");
    ChunkConfig cfg = to_cfg(config);
    cfg.paging_exact = true; if (n_exact) cfg.exact_pages.assign(exact_pages, exact_pages + n_exact);
    std::vector<uint8_t> out;
    dispatch_bits(dtype_bits(dtype), [&](auto tag) {
      typedef decltype(tag) LTYPE;
      out = simple_compress_t<LTYPE>((const LTYPE*)nums, n, dtype, cfg, uniform_type != 0);
    });
    if (out.size() > dst_cap) fail(kInvalidArgument, "destination too small");
    std::memcpy(dst, out.data(), out.size());
    *n_written = out.size();
  });
}

// TEST-ONLY stream generator (pco_oracle_testenc.hpp): a standalone file of one chunk per entry of `chunks`, written with features the
// restated encoder lacks (Dict mode, Conv1 delta, delta'd secondary variable, lookback state); any valid stream will do for decode sweeps.
int pco_oracle_test_encode(const void* nums, size_t n, uint8_t dtype, const TestEncSpec* spec, const size_t* chunks, size_t n_chunks,
                           uint8_t* dst, size_t dst_cap, size_t* n_written) {
  return guard([&] {
    if (!dtype_valid(dtype)) fail(kInvalidArgument, "invalid dtype");
    std::vector<size_t> cs(chunks, chunks + n_chunks);
    std::vector<uint8_t> out;
    dispatch_bits(dtype_bits(dtype), [&](auto tag) {
      typedef decltype(tag) LTYPE;
      out = test_encode_file<LTYPE>((const LTYPE*)nums, n, dtype, *spec, cs);
    });
    if (out.size() > dst_cap) fail(kInvalidArgument, "destination too small");
    std::memcpy(dst, out.data(), out.size());
    *n_written = out.size();
  });
}

// wrapped::ChunkCompressor (wrapped/chunk_compressor.rs:442-705): ChunkMeta bytes, then every page's bytes back to back.
// sizes[0] = meta bytes, sizes[1 + i] = bytes of page i, page_ns[i] = numbers in page i; *n_pages <= cap_pages.
static int wrapped_compress_impl(const void* nums, size_t n, uint8_t dtype, const PcoOracleConfig* config, const size_t* exact_pages, size_t n_exact,
                                 uint8_t* dst, size_t dst_cap, size_t* sizes, size_t* page_ns, size_t cap_pages, size_t* n_pages);
int pco_oracle_wrapped_compress(const void* nums, size_t n, uint8_t dtype, const PcoOracleConfig* config, uint8_t* dst, size_t dst_cap,
                                size_t* sizes, size_t* page_ns, size_t cap_pages, size_t* n_pages) {
  return wrapped_compress_impl(nums, n, dtype, config, nullptr, 0, dst, dst_cap, sizes, page_ns, cap_pages, n_pages);
}
// the same with PagingSpec::Exact (chunk_config.rs:124,162-180)
int pco_oracle_wrapped_compress_exact(const void* nums, size_t n, uint8_t dtype, const PcoOracleConfig* config, const size_t* exact_pages, size_t n_exact,
                                      uint8_t* dst, size_t dst_cap, size_t* sizes, size_t* page_ns, size_t cap_pages, size_t* n_pages) {
  static const size_t none = 0;
  return wrapped_compress_impl(nums, n, dtype, config, exact_pages ? exact_pages : &none, n_exact, dst, dst_cap, sizes, page_ns, cap_pages, n_pages);
}
static int wrapped_compress_impl(const void* nums, size_t n, uint8_t dtype, const PcoOracleConfig* config, const size_t* exact_pages, size_t n_exact,
                                 uint8_t* dst, size_t dst_cap, size_t* sizes, size_t* page_ns, size_t cap_pages, size_t* n_pages) {
  return guard([&] {
    if (!dtype_valid(dtype)) fail(kInvalidArgument, "invalid dtype");
    ChunkConfig cfg = to_cfg(config);
    if (exact_pages) { cfg.paging_exact = true; cfg.exact_pages.assign(exact_pages, exact_pages + n_exact); }
    dispatch_bits(dtype_bits(dtype), [&](auto tag) {
      typedef decltype(tag) LTYPE;
      ChunkCompressor<LTYPE>* cc = new ChunkCompressor<LTYPE>();
      try {
        chunk_compressor_new<LTYPE>(*cc, (const LTYPE*)nums, n, dtype, cfg);
        if (cc->n_pages() > cap_pages) fail(kInvalidArgument, "too many pages");
        size_t pos = 0;
        auto emit = [&](BitWriter& w, size_t& out_size) {
          w.buf.resize(w.byte_len());
          if (pos + w.buf.size() > dst_cap) fail(kInvalidArgument, "destination too small");
          std::memcpy(dst + pos, w.buf.data(), w.buf.size()); pos += w.buf.size(); out_size = w.buf.size();
        };
        { BitWriter w; cc->write_meta(w); emit(w, sizes[0]); }
        for (size_t i = 0; i < cc->n_pages(); i++) { BitWriter w; cc->write_page(i, w); emit(w, sizes[1 + i]); page_ns[i] = cc->page_infos[i].page_n; }
        *n_pages = cc->n_pages();
      } catch (...) { delete cc; throw; }
      delete cc;
    });
  });
}

int pco_oracle_simple_decompress(const uint8_t* src, size_t len, uint8_t dtype, void* dst, size_t dst_cap_elems, size_t* n_written) {
  return guard([&] {
    if (!dtype_valid(dtype)) fail(kInvalidArgument, "invalid dtype");
    dispatch_bits(dtype_bits(dtype), [&](auto tag) {
      typedef decltype(tag) LTYPE;
      std::vector<LTYPE> v = simple_decompress_t<LTYPE>(src, len, dtype);
      if (v.size() > dst_cap_elems) fail(kInvalidArgument, "destination too small");
      if (!v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(LTYPE));
      *n_written = v.size();
    });
  });
}

// ---------------------------------------------------------------------------
// chunk inspection: parse the first chunk's metadata of a standalone file
// ---------------------------------------------------------------------------
typedef struct PcoOracleChunkInfo {
  uint32_t mode_kind; uint32_t mode_k; uint64_t mode_base_latent;
  uint32_t delta_kind; uint32_t delta_order; uint32_t window_n_log; uint32_t state_n_log;
  uint32_t n; uint32_t dtype;
  uint32_t var_present[3]; uint32_t ans_size_log[3]; uint32_t n_bins[3];
  uint32_t standalone_version; uint32_t uniform_type; uint32_t fmt_major; uint32_t fmt_minor;
  uint64_t n_hint; uint64_t meta_end_byte;
} PcoOracleChunkInfo;

static void fill_info(const ChunkMeta& m, PcoOracleChunkInfo* info, uint64_t* bins_out, size_t max_bins) {
  info->mode_kind = m.mode.kind; info->mode_k = m.mode.k; info->mode_base_latent = m.mode.base_latent;
  info->delta_kind = m.delta.kind; info->delta_order = (uint32_t)m.delta.order;
  info->window_n_log = m.delta.window_n_log; info->state_n_log = m.delta.state_n_log;
  for (int v = 0; v < 3; v++) {
    info->var_present[v] = m.vars[v].present; info->ans_size_log[v] = m.vars[v].ans_size_log; info->n_bins[v] = (uint32_t)m.vars[v].bins.size();
    if (bins_out) for (size_t b = 0; b < m.vars[v].bins.size() && b < max_bins; b++) {
      uint64_t* o = bins_out + ((size_t)v * max_bins + b) * 3;
      o[0] = m.vars[v].bins[b].weight; o[1] = m.vars[v].bins[b].lower; o[2] = m.vars[v].bins[b].offset_bits;
    }
  }
}

// bins_out: up to max_bins entries of (weight, lower, offset_bits) per var, laid out var-major
int pco_oracle_inspect_first_chunk(const uint8_t* src, size_t len, PcoOracleChunkInfo* info,
                                   uint64_t* bins_out, size_t max_bins) {
  return guard([&] {
    std::vector<uint8_t> padded(len + MAX_BATCH_LATENT_VAR_SIZE + 64, 0);
    if (len) std::memcpy(padded.data(), src, len);
    BitReader r{padded.data(), len * 8, padded.size(), 0};
    FileHeader h = read_file_header(r);
    std::memset(info, 0, sizeof(*info));
    info->standalone_version = (uint32_t)h.standalone_version; info->uniform_type = h.uniform_type;
    info->fmt_major = h.fmt_major; info->fmt_minor = h.fmt_minor; info->n_hint = h.n_hint;
    uint8_t tb = r.read_aligned_bytes(1)[0];
    r.check_in_bounds();
    if (tb == 0) { info->dtype = 0; return; }
    if (!dtype_valid(tb)) fail(kCorruption, "bad dtype byte");
    info->dtype = tb;
    info->n = (uint32_t)r.read_uint(BITS_TO_ENCODE_N_ENTRIES) + 1;
    ChunkMeta m = read_chunk_meta(r, h.fmt_major, dtype_bits(tb));
    info->meta_end_byte = r.bit_pos >> 3;
    fill_info(m, info, bins_out, max_bins);
  });
}

// ---------------------------------------------------------------------------
// per-stage entry points (for the reference's inline known-answer tests and for
// stage-by-stage comparison with the HIP kernels)
// ---------------------------------------------------------------------------
int pco_oracle_spread_state_symbols(uint32_t size_log, const uint32_t* weights, size_t n_weights, uint32_t* out) {
  return guard([&] {
    std::vector<uint32_t> w(weights, weights + n_weights);
    auto s = spread_state_symbols(size_log, w);
    std::memcpy(out, s.data(), s.size() * 4);
  });
}
int pco_oracle_quantize_weights(const uint32_t* counts, size_t n, size_t total_count, uint32_t max_size_log,
                                uint32_t* out_weights, uint32_t* out_size_log) {
  return guard([&] {
    std::vector<uint32_t> c(counts, counts + n);
    auto q = quantize_weights(c, total_count, max_size_log);
    *out_size_log = q.first; std::memcpy(out_weights, q.second.data(), q.second.size() * 4);
  });
}
int pco_oracle_quantize_weights_to(const uint32_t* counts, size_t n, size_t total_count, uint32_t size_log, uint32_t* out_weights) {
  return guard([&] {
    std::vector<uint32_t> c(counts, counts + n);
    auto q = quantize_weights_to(c, total_count, size_log);
    std::memcpy(out_weights, q.data(), q.size() * 4);
  });
}
float pco_oracle_log2_approx(float x) { return log2_approx(x); }

// histogram: rule = 0 literal (mutates a copy), 1 = multiset rule on sorted copy
// wrapped::PageDecompressor, batch after batch (wrapped/page_decompressor.rs:115-221): decodes the page `page` of `page_n` numbers of a chunk
// whose ChunkMeta bytes are `meta` until it is done or a batch fails.  *n_ok = the numbers of the batches decoded before the failure,
// *err = 0 or the ErrKind of the failure, *in_meta = 1 when the page's own metadata already failed (PageDecompressor::new would have).
int pco_oracle_wrapped_page_prefix(const uint8_t* meta, size_t meta_len, const uint8_t* page, size_t page_len, uint8_t dtype, uint8_t fmt_major,
                                   size_t page_n, void* dst, size_t* n_ok, int* err, int* in_meta) {
  return guard([&] {
    dispatch_bits(dtype_bits(dtype), [&](auto tag) {
      typedef decltype(tag) LTYPE;
      *n_ok = 0; *err = 0; *in_meta = 0;
      std::vector<uint8_t> pm(meta_len + MAX_BATCH_LATENT_VAR_SIZE + 64, 0);
      std::memcpy(pm.data(), meta, meta_len);
      BitReader rm{pm.data(), meta_len * 8, pm.size(), 0};
      ChunkMeta cm = read_chunk_meta(rm, fmt_major, LT<LTYPE>::BITS);
      std::vector<uint8_t> pp(page_len + MAX_BATCH_LATENT_VAR_SIZE + 64, 0);
      if (page_len) std::memcpy(pp.data(), page, page_len);
      BitReader r{pp.data(), page_len * 8, pp.size(), 0};
      std::unique_ptr<ChunkDecoder<LTYPE>> cd(new ChunkDecoder<LTYPE>());
      cd->init(cm, dtype);
      try { cd->start_page(r, page_n); } catch (const PcoErr& e) { *err = (int)e.kind; *in_meta = 1; return; }
      size_t done = 0;
      while (done < page_n) {
        const size_t bn = std::min(FULL_BATCH_N, page_n - done);
        try { cd->read_batch(r, (LTYPE*)dst + done, bn); } catch (const PcoErr& e) { *err = (int)e.kind; break; }
        done += bn; *n_ok = done;
      }
    });
  });
}
// TEST HOOK: 0 = the reference's literal histogram (default), 1 = the multiset rule in every encode of this thread (see train_infos)
int pco_oracle_set_hist_rule(int rule) { hist_rule_hook() = rule; return 0; }
int pco_oracle_histogram(const void* latents, size_t n, int latent_bits, uint32_t n_bins_log, int rule,
                         uint64_t* out_count, uint64_t* out_lower, uint64_t* out_upper, size_t* out_n, int* out_fallback) {
  return guard([&] {
    dispatch_bits(latent_bits, [&](auto tag) {
      typedef decltype(tag) LTYPE;
      std::vector<LTYPE> v((const LTYPE*)latents, (const LTYPE*)latents + n);
      std::vector<HistogramBin<LTYPE>> h; bool fb = false;
      if (rule == 0) h = histogram<LTYPE>(v.data(), n, n_bins_log, &fb);
      else if (rule == 2) {   // the heapsort branch's tie rule on its own: apply_sorted (histograms.rs:164-206) over the whole sorted input
        std::sort(v.begin(), v.end());
        HistogramBuilder<LTYPE> hb(n, n_bins_log); hb.apply_sorted(v.data(), n); h = hb.dst; fb = true;
      }
      else { std::sort(v.begin(), v.end()); h = histogram_multiset_rule<LTYPE>(v.data(), n, n_bins_log); }
      for (size_t i = 0; i < h.size(); i++) { out_count[i] = h[i].count; out_lower[i] = h[i].lower; out_upper[i] = h[i].upper; }
      *out_n = h.size(); if (out_fallback) *out_fallback = fb;
    });
  });
}
// optimize_bins on histogram bins -> (weight=count, lower, upper, offset_bits)
int pco_oracle_optimize_bins(const uint64_t* counts, const uint64_t* lowers, const uint64_t* uppers, size_t n_bins,
                             int latent_bits, uint32_t ans_size_log,
                             uint64_t* out_weight, uint64_t* out_lower, uint64_t* out_upper, uint32_t* out_offset_bits, size_t* out_n) {
  return guard([&] {
    dispatch_bits(latent_bits, [&](auto tag) {
      typedef decltype(tag) LTYPE;
      std::vector<HistogramBin<LTYPE>> h;
      for (size_t i = 0; i < n_bins; i++) h.push_back(HistogramBin<LTYPE>{(size_t)counts[i], (LTYPE)lowers[i], (LTYPE)uppers[i]});
      auto o = optimize_bins<LTYPE>(h, ans_size_log);
      for (size_t i = 0; i < o.size(); i++) { out_weight[i] = o[i].weight; out_lower[i] = o[i].lower; out_upper[i] = o[i].upper; out_offset_bits[i] = o[i].offset_bits; }
      *out_n = o.size();
    });
  });
}
int pco_oracle_choose_lookbacks(const void* latents, size_t n, int latent_bits, uint32_t window_n_log, uint32_t state_n_log, uint32_t* out, size_t* out_n) {
  return guard([&] {
    std::vector<uint32_t> lb;
    if (latent_bits == 32) lb = choose_lookbacks<uint32_t>(window_n_log, state_n_log, (const uint32_t*)latents, n);
    else if (latent_bits == 64) lb = choose_lookbacks<uint64_t>(window_n_log, state_n_log, (const uint64_t*)latents, n);
    else fail(kInvalidArgument, "bad latent bits");
    if (!lb.empty()) std::memcpy(out, lb.data(), lb.size() * 4);
    *out_n = lb.size();
  });
}
int pco_oracle_mode_sample_indices(size_t n, uint64_t* out, size_t* out_n) {
  return guard([&] {
    std::vector<size_t> idx;
    if (!choose_mode_sample_indices(n, idx)) { *out_n = 0; return; }
    for (size_t i = 0; i < idx.size(); i++) out[i] = idx[i];
    *out_n = idx.size();
  });
}
// consecutive delta encode in place (u32); returns moments
int pco_oracle_consecutive_encode_u32(uint32_t* latents, size_t n, size_t order, uint32_t* moments) {
  return guard([&] {
    auto m = consecutive_encode_in_place<uint32_t>(order, latents, n);
    for (size_t i = 0; i < m.size(); i++) moments[i] = m[i];
  });
}
// mode split for stage comparison: writes primary/secondary latents (same width as dtype)
int pco_oracle_split_latents(const void* nums, size_t n, uint8_t dtype, const PcoOracleConfig* config,
                             void* primary, void* secondary, uint32_t* mode_kind, uint64_t* mode_payload) {
  return guard([&] {
    ChunkConfig cfg = to_cfg(config);
    dispatch_bits(dtype_bits(dtype), [&](auto tag) {
      typedef decltype(tag) LTYPE;
      Mode mode; auto s = choose_mode_and_split<LTYPE>((const LTYPE*)nums, n, dtype, cfg, mode);
      std::memcpy(primary, s.primary.data(), n * sizeof(LTYPE));
      if (s.has_secondary && secondary) std::memcpy(secondary, s.secondary.data(), n * sizeof(LTYPE));
      *mode_kind = mode.kind; *mode_payload = mode.kind == kFloatQuant ? mode.k : mode.base_latent;
    });
  });
}

// Trained chunk plan for stage-by-stage GPU comparison: encodes one chunk and reports,
// per latent var, the bins (weight, lower, offset_bits), ans_size_log, and whether the
// histogram took the order-dependent heapsort fallback.
int pco_oracle_chunk_plan(const void* nums, size_t n, uint8_t dtype, const PcoOracleConfig* config,
                          PcoOracleChunkInfo* info, uint64_t* bins_out, size_t max_bins, int* hist_fallback) {
  return guard([&] {
    ChunkConfig cfg = to_cfg(config); cfg.paging_exact = true; cfg.exact_pages = {n};
    std::memset(info, 0, sizeof(*info));
    dispatch_bits(dtype_bits(dtype), [&](auto tag) {
      typedef decltype(tag) LTYPE;
      std::unique_ptr<ChunkCompressor<LTYPE>> cc(new ChunkCompressor<LTYPE>());
      chunk_compressor_new<LTYPE>(*cc, (const LTYPE*)nums, n, dtype, cfg);
      info->n = (uint32_t)n; info->dtype = dtype;
      fill_info(cc->meta, info, bins_out, max_bins);
      if (hist_fallback) *hist_fallback = (cc->dvar.present && cc->dvar.hist_fallback) || cc->pvar.hist_fallback || (cc->svar.present && cc->svar.hist_fallback);
    });
  });
}

// int_mult::choose_candidate_base KAT (mode/int_mult.rs:205-213)
int pco_oracle_choose_candidate_base_u32(const uint32_t* sample, size_t n, uint32_t* base, double* score, int* found) {
  return guard([&] {
    std::vector<uint32_t> s(sample, sample + n); uint32_t b = 0; double sc = 0;
    *found = choose_candidate_base<uint32_t>(s, b, sc) ? 1 : 0; *base = b; *score = sc;
  });
}
// float_quant::compute_bid KAT (mode/float_quant.rs:73-91)
int pco_oracle_float_quant_bid_f32(const float* sample, size_t n, uint32_t* k, double* bits_saved, int* found) {
  return guard([&] {
    std::vector<float> s(sample, sample + n); Bitlen kk = 0; double bs = 0;
    *found = FM<uint32_t>::quant_compute_bid(s, kk, bs) ? 1 : 0; *k = kk; *bits_saved = bs;
  });
}
// lookback encode in place (u32) -> state; KAT delta/lookback.rs:254-300
int pco_oracle_lookback_encode_u32(uint32_t* latents, size_t n, uint32_t state_n_log, const uint32_t* lookbacks, uint32_t* state_out) {
  return guard([&] {
    auto st = lookback_encode_in_place<uint32_t>(state_n_log, lookbacks, latents, n);
    for (size_t i = 0; i < st.size(); i++) state_out[i] = st[i];
  });
}

}  // extern "C"

// ---------------------------------------------------------------------------
// cpu_baseline leg of bench.py: the restated reference algorithm timed on the host cores without Python in the loop.
// Each thread owns a private copy of the chunk and loops compress -> decompress (what pco_cli's bench does per
// iteration, pco_cli/src/bench/codecs/mod.rs:150-231) until the deadline.  out[0] = chunks done (all threads),
// out[1] = wall seconds, out[2] / out[3] = summed per-thread seconds inside compress / decompress.
// ---------------------------------------------------------------------------
#include <atomic>
#include <chrono>
#include <thread>
#include <malloc.h>
extern "C" int pco_oracle_bench(const void* nums, size_t n, uint8_t dtype, const PcoOracleConfig* config, uint32_t n_threads,
                                double seconds, double* out) {
  return guard([&] {
    if (!dtype_valid(dtype)) fail(kInvalidArgument, "invalid dtype");
    if (n_threads == 0) n_threads = 1;
    // The restated algorithm allocates its working vectors per call, like the reference (Vec per chunk).  With glibc's defaults a 2 MiB
    // vector is an mmap + page faults + munmap every time, and hundreds of threads then serialise on the process's mmap lock; serve
    // them from the per-thread arenas instead (the equivalent of the reference running under a pooling allocator).
    mallopt(M_MMAP_THRESHOLD, 1 << 30); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_ARENA_MAX, (int)n_threads + 8);
    const ChunkConfig cfg = to_cfg(config);
    const int bits = dtype_bits(dtype);
    const size_t bytes = n * (size_t)(bits / 8);
    std::atomic<uint64_t> done{0}; std::atomic<int> failed{0};
    std::vector<double> t_enc(n_threads, 0.0), t_dec(n_threads, 0.0);
    std::atomic<uint32_t> ready{0}; std::atomic<bool> go{false};
    typedef std::chrono::steady_clock Clock;
    Clock::time_point deadline;
    auto worker = [&](uint32_t k) {
      std::vector<uint8_t> local((const uint8_t*)nums, (const uint8_t*)nums + bytes);
      ready.fetch_add(1);
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      try {
        dispatch_bits(bits, [&](auto tag) {
          typedef decltype(tag) LTYPE;
          uint64_t mine = 0;
          while (Clock::now() < deadline || mine == 0) {
            const auto a = Clock::now();
            std::vector<uint8_t> enc = simple_compress_t<LTYPE>((const LTYPE*)local.data(), n, dtype, cfg, false);
            const auto b = Clock::now();
            std::vector<LTYPE> back = simple_decompress_t<LTYPE>(enc.data(), enc.size(), dtype);
            const auto c = Clock::now();
            if (back.size() != n || std::memcmp(back.data(), local.data(), bytes) != 0) { failed.store(1); return; }
            t_enc[k] += std::chrono::duration<double>(b - a).count(); t_dec[k] += std::chrono::duration<double>(c - b).count();
            mine++;
          }
          done.fetch_add(mine);
        });
      } catch (...) { failed.store(1); }
    };
    std::vector<std::thread> th;
    for (uint32_t k = 0; k < n_threads; k++) th.emplace_back(worker, k);
    while (ready.load() < n_threads) std::this_thread::yield();
    const auto t0 = Clock::now();
    deadline = t0 + std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double>(seconds));
    go.store(true, std::memory_order_release);
    for (auto& t : th) t.join();
    const double wall = std::chrono::duration<double>(Clock::now() - t0).count();
    if (failed.load()) fail(kCorruption, "oracle bench: round trip failed");
    double se = 0, sd = 0; for (uint32_t k = 0; k < n_threads; k++) { se += t_enc[k]; sd += t_dec[k]; }
    out[0] = (double)done.load(); out[1] = wall; out[2] = se; out[3] = sd;
  });
}

// ---------------------------------------------------------------------------
// Parity spot check of bench.py's warm-up step: `n_chunks` chunks of n numbers each (numbers at nums + i * stride_bytes) are
// compressed by the restated reference on `n_threads` host threads and compared with the standalone chunks the GPU produced
// (got + got_off[i], got_len[i] bytes: dtype byte | n - 1 | ChunkMeta | page, i.e. the oracle's file minus header and terminator).
// *n_bad = chunks that differ, *first_bad = the first of them (or -1).
// ---------------------------------------------------------------------------
extern "C" int pco_oracle_verify_chunks(const void* nums, size_t n_chunks, size_t n, size_t stride_bytes, uint8_t dtype, const PcoOracleConfig* config,
                                        const uint8_t* got, const uint64_t* got_off, const uint64_t* got_len, uint32_t n_threads,
                                        uint64_t* n_bad, int64_t* first_bad) {
  return guard([&] {
    if (!dtype_valid(dtype)) fail(kInvalidArgument, "invalid dtype");
    if (n_threads == 0) n_threads = 1;
    mallopt(M_MMAP_THRESHOLD, 1 << 30); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_ARENA_MAX, (int)n_threads + 8);
    const ChunkConfig cfg = to_cfg(config);
    const int bits = dtype_bits(dtype);
    std::atomic<size_t> next{0}; std::atomic<uint64_t> bad{0}; std::atomic<int64_t> first{-1}; std::atomic<int> failed{0};
    auto worker = [&]() {
      try {
        dispatch_bits(bits, [&](auto tag) {
          typedef decltype(tag) LTYPE;
          for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n_chunks) break;
            const std::vector<uint8_t> want = simple_compress_t<LTYPE>((const LTYPE*)((const uint8_t*)nums + i * stride_bytes), n, dtype, cfg, false);
            const size_t len = (size_t)got_len[i];
            const bool same = want.size() >= len + 1 && std::memcmp(want.data() + (want.size() - 1 - len), got + got_off[i], len) == 0 &&
                              want.size() - 1 - len == standalone_header_len(n);
            if (!same) { bad.fetch_add(1); int64_t cur = first.load(); while ((cur < 0 || (int64_t)i < cur) && !first.compare_exchange_weak(cur, (int64_t)i)) {} }
          }
        });
      } catch (...) { failed.store(1); }
    };
    std::vector<std::thread> th;
    for (uint32_t k = 0; k < n_threads; k++) th.emplace_back(worker);
    for (auto& t : th) t.join();
    if (failed.load()) fail(kCorruption, "oracle verify: the oracle failed to compress a chunk");
    *n_bad = bad.load(); *first_bad = first.load();
  });
}
