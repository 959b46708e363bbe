// encode_fast.hip -- the page encoder of the common case, split by what bounds each part:
//
//   enc_dissect_kernel  (HBM-bound)      latent -> bin symbol (u8) + per-batch offset-bit totals
//                                        (compression_table.rs:51-74, chunk_latent_compressor.rs:60-94)
//   enc_walk_kernel     (latency-bound)  the reverse tANS walk: 8 (page, variable) items per wave, four lanes per
//                                        item, one lane per interleaved chain; emits (bits, value) per symbol and
//                                        the final states (chunk_latent_compressor.rs:96-132, ans/encoding.rs:65-91)
//   enc_scan_kernel     (tiny)           bit position of every run of kRunBatches batches, page size, overflow check
//   enc_pack_kernel     (HBM-bound)      metadata + tANS fields + offset fields of one run per wave, joined to its
//                                        neighbours with atomicOr on the boundary dwords
//                                        (wrapped/chunk_compressor.rs:624-705, bit_writer.rs:22-42)
//
// enc_page_kernel (encode_kernels.hip) remains the general path: fallback chunks, ChunkMeta-only tasks and tANS
// tables beyond kFastEncMaxAsl.  Both produce the same bytes.
//
// Symbol / field scratch layout ("quad-transposed", shared with the decoder's symbol scratch): within every block
// of 16 consecutive latents of a page variable, unit j (a dword of 4 symbols, or 4 u16 fields) holds chain j's
// entries of the block's four steps, i.e. latents 16B+j, 16B+4+j, 16B+8+j, 16B+12+j.
#pragma once

namespace pcogfx {

constexpr uint32_t kRunBatches = 64;          // batches per dissect block / pack run

struct EncFast {
  uint8_t* sym;         // [task][slot][n_stride] bin symbols, quad-transposed per page variable
  uint16_t* answ;       // [task][slot][n_stride] bits << 12 | value, quad-transposed
  uint32_t* bat;        // [page][3][bat_stride][2]: offset bits, tANS bits of the batch
  uint64_t* run_start;  // [page][run_stride] first bit of the run, relative to the page task's dst
  uint32_t* fstate;     // [page][3][4] final tANS states
  uint16_t* vlut;       // [task][slot][kDirectHistRange] value -> bin | offset bits << 8 of the variables enc_walkd_kernel takes (enc_vlut_kernel)
  uint32_t bat_stride, run_stride;
  uint32_t pack1;       // enc_pack1_kernel takes the pages it can (pack1_takes)
  uint32_t runs_per_page, fused; // 1-D grids of dissect / pack: block = page * runs_per_page + run.  fused: which variables enc_walkd_kernel takes (wd_takes)
  uint64_t stride;               // elements per (task, slot) in sym / answ: n_stride + 16 per page, so that the 16-latent
                                 // blocks of neighbouring pages never overlap (see fast_at)
  uint64_t* body;       // [page][2] enc_walkp_kernel (encode_walkpack.hip): bits of the page's packed body | 1 << 63, its first bit in `answ`; null: that kernel is off
  uint32_t* wp_block;   // [walk block] 1: enc_walkp_kernel took the block (enc_walkd_kernel leaves it alone)
};
// first scratch index of a page variable: its latent index plus 16 slots per preceding page
__device__ __forceinline__ uint64_t fast_at(const EncPage PCO_GLOBAL* pg, uint32_t skip) { return uni((uint64_t)pg->start) + skip + 16ull * uni(pg->page_idx); }

__device__ __forceinline__ uint8_t PCO_GLOBAL* fsym_ptr(const EncWorkspace& ws, const EncFast& fx, uint32_t task, uint32_t var) {
  return (uint8_t PCO_GLOBAL*)fx.sym + ((uint64_t)task * ws.n_slots + ws.slot_of_var[var]) * fx.stride;
}
__device__ __forceinline__ uint16_t PCO_GLOBAL* fansw_ptr(const EncWorkspace& ws, const EncFast& fx, uint32_t task, uint32_t var) {
  return (uint16_t PCO_GLOBAL*)fx.answ + ((uint64_t)task * ws.n_slots + ws.slot_of_var[var]) * fx.stride;
}

// The page's view of one variable (as in page_task)
struct PageVar { uint32_t present, n_bins, asl, max_ob, needs_ans, trivial, skip, n_lat, compact; uint64_t minv, rel, range; };   // rel: what the compact (16-bit) latents are relative to
__device__ __forceinline__ PageVar page_var(const EncChunk PCO_GLOBAL* ch, uint32_t v, uint32_t page_n) {
  PageVar r;
  r.present = uni(ch->v[v].present); r.n_bins = uni(ch->v[v].n_bins); r.asl = uni(ch->v[v].ans_size_log); r.max_ob = uni(ch->v[v].max_ob);
  r.needs_ans = uni(ch->v[v].needs_ans); r.trivial = uni(ch->v[v].is_trivial);
  r.skip = v == 2 ? 0u : uni(ch->v[v].lat_start);
  if (r.skip > page_n) r.skip = page_n;
  r.n_lat = page_n - r.skip;
  r.compact = uni(ch->v[v].hist_path) == 0 ? 1u : 0u;   // histogram by LDS counting: compact latents exist (clat_ptr)
  r.minv = uni((uint64_t)ch->v[v].minv);
  r.range = uni((uint64_t)ch->v[v].maxv) - r.minv;
  // compact latents: the histogram's copy relative to the minimum, or -- when the split speculated on 16-bit latents and held -- the split's own, relative to c16_ref
  r.rel = v != 0 && uni(ch->c16_ok) == 1 ? uni(ch->c16_ref[v == 2 ? 1 : 0]) : r.minv;
  return r;
}

// The pages of the lean pack kernel (enc_pack1_kernel): ONE or TWO latent variables that write anything (the primary alone: classic mode
// without lookback; primary + secondary: int-mult / float-mult / float-quant; lookback variable + primary), each with at most 256 bins and held
// either as 16-bit latents or at a full width of 32 / 64 bits.  Left to the general kernel: three variables, levels 9-12, full-width latents of
// the 8- / 16-bit types, and the one-bin / full-width-offsets page that is a shifted copy (pack_run).  Returns 0 or the mask of the variables on.
__device__ __forceinline__ uint32_t pack1_mask(const PageVar (&pv)[3], uint32_t latent_bits) {
  uint32_t mask = 0, n_on = 0;
#pragma unroll
  for (int v = 0; v < 3; v++) {
    if (!pv[v].present || pv[v].trivial) continue;
    mask |= 1u << v; n_on++;
    if (pv[v].n_bins > 256 || pv[v].n_bins == 0) return 0;
    if (!pv[v].compact && latent_bits < 32) return 0;   // (full-width latents of the 8- / 16-bit types: no lean instantiation)
  }
  if (!(mask & 2u) || n_on > 2) return 0;
  if (n_on == 1 && pv[1].n_bins == 1 && !pv[1].needs_ans && pv[1].max_ob == latent_bits && !pv[1].compact) return 0;   // the shifted copy
  return mask;
}
__device__ __forceinline__ bool pack1_takes(const PageVar (&pv)[3], uint32_t latent_bits) { return pack1_mask(pv, latent_bits) != 0; }

// Q (page, variable) items per wave: 8 (slot 4608 B: any table of the fast path) or 16 (slot 2304 B, all 64 lanes busy).
// The walk is latency-bound, so what matters is that every item is resident at once: with more than 8192 items the launch
// first walks the items whose tables fit the small slots 16 per wave, then the rest 8 per wave (never worse than two rounds
// of 8 per wave).  Slot: next states u16[T] | info u64[n_bins] (see ew_step) | ... | symbols u8[2][256] of the current / next batch.
template <uint32_t Q> struct EwCfg {
  static constexpr uint32_t kSlotBytes = Q == 16 ? 2304u : 4608u;
  static constexpr uint32_t kSymOff = kSlotBytes - 512;
  static constexpr uint32_t kLdsBytes = Q * kSlotBytes;   // 36864
};
__device__ __forceinline__ uint32_t ew_info_off(uint32_t asl) { return ((2u << asl) + 7u) & ~7u; }
__device__ __forceinline__ bool ew_fits16(uint32_t asl, uint32_t n_bins) { return ew_info_off(asl) + 8u * n_bins <= EwCfg<16>::kSymOff; }
// enc_walkd_kernel (below) walks a page variable when fused != 0 and, in a launch of more than 8192 items (fused == 2), its tables do not fit
// the 16-per-wave slots of enc_walk_kernel<16>; it also FINDS the symbols of the variables it walks whose latents are 16-bit and span fewer
// than 4096 values (wd_takes: enc_dissect_kernel leaves those alone).
constexpr uint32_t kFusedLookups = 0x100u;   // EncFast::fused flag: the value -> bin tables exist (the host allocates them unless they would dwarf the input)
__device__ __forceinline__ bool wd_walks(uint32_t fused, const PageVar& pv) {
  const uint32_t mode = fused & 0xffu;
  return mode != 0 && pv.present && pv.n_bins > 1 && pv.n_lat > 0 && (mode == 1 || !ew_fits16(pv.asl, pv.n_bins));
}
// enc_walkseg_kernel (encode_walkseg.hip) walks the long items, sixteen segments side by side; the kernels here leave those alone
constexpr uint32_t kFusedSegments = 0x200u;   // EncFast::fused flag
constexpr uint32_t kWsSegs = 16, kWsMinBatches = 64, kWsMaxItems = 4096;           // segments per item; items shorter than 64 batches (16 384 latents) stay with the unsegmented kernels
__device__ __forceinline__ bool ws_walks(uint32_t fused, const PageVar& pv) {
  return (fused & 0xffu) != 0 && (fused & kFusedSegments) != 0 && pv.present && pv.n_bins > 1 && pv.n_lat >= kWsMinBatches * kBatchN;
}
__device__ __forceinline__ bool wd_takes(uint32_t fused, const PageVar& pv) { return (fused & kFusedLookups) != 0 && wd_walks(fused, pv) && pv.compact && pv.range < kDirectHistRange; }
// Where a table's window of used slots starts is rotated from table to table: the tables sit 8 KB apart and a narrow variable uses a
// quarter of its table or less -- at the same offset in every table of similar chunks, i.e. in the same few L2 sets.
__device__ __forceinline__ uint32_t vlut_rot(uint32_t table_index) { return (table_index * 1600u) & (kDirectHistRange - 1) & ~31u; }
__device__ __forceinline__ uint16_t PCO_GLOBAL* vlut_ptr(const EncWorkspace& ws, const EncFast& fx, uint32_t task, uint32_t var) {
  return (uint16_t PCO_GLOBAL*)fx.vlut + ((uint64_t)task * ws.n_slots + ws.slot_of_var[var]) * kDirectHistRange;
}

// =========================================================================================================
// dissect
// =========================================================================================================
constexpr uint32_t kDisVarBytes = 2048 + 256 + kDirectHistRange;    // search lowers (8 B stride) + offset bits + value -> bin table
constexpr uint32_t kDisLdsBytes = 3 * kDisVarBytes;

// One batch of one variable.  kLut: the variable's value range is below kDirectHistRange, so the bin of a latent is one
// LDS table lookup away; otherwise the branch-free lower bound over the padded lowers (compression_table.rs:51-74),
// with the four searches of a lane interleaved so that their LDS round trips overlap.
// LV: element type read (the latent type, or uint16_t for compact latents, whose lowers are stored relative to the minimum)
template <class LV, bool kLut, bool kFull>
__device__ __forceinline__ void dissect_batch_t(const uint8_t PCO_LDS* vt, const LV PCO_GLOBAL* lat, uint8_t PCO_GLOBAL* sym_out, uint32_t PCO_GLOBAL* ob_bits_out,
                                                uint32_t cnt, uint32_t n_bins, uint32_t search_log, uint64_t minv) {
  const uint32_t lane = lane_id();
  const uint64_t PCO_LDS* low = (const uint64_t PCO_LDS*)vt;
  const uint8_t PCO_LDS* obs = vt + 2048;
  const uint8_t PCO_LDS* lut = vt + 2048 + 256;
  uint64_t x[4]; uint32_t sym[4] = {0, 0, 0, 0};
  if (kFull) {
    if constexpr (sizeof(LV) == 2) {   // four 16-bit latents: one 8-byte load (2-byte aligned: a page may start at an odd index)
      typedef uint64_t __attribute__((aligned(2))) u64_align2;
      const uint64_t w = *(const u64_align2 PCO_GLOBAL*)(lat + 4 * lane);
      x[0] = w & 0xffffu; x[1] = (w >> 16) & 0xffffu; x[2] = (w >> 32) & 0xffffu; x[3] = w >> 48;
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) x[k] = (uint64_t)lat[4 * lane + k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = 4 * lane + k < cnt ? (uint64_t)lat[4 * lane + k] : minv;
  }
  if (kLut) {
#pragma unroll
    for (int k = 0; k < 4; k++) sym[k] = lut[(uint32_t)(x[k] - minv)];
  } else {
    for (uint32_t depth = 0; depth < search_log; depth++) {
      const uint32_t bis = 1u << (search_log - 1 - depth);
#pragma unroll
      for (int k = 0; k < 4; k++) { const uint64_t probe = low[sym[k] + bis]; sym[k] += x[k] >= probe ? bis : 0u; }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) sym[k] = sym[k] < n_bins - 1 ? sym[k] : n_bins - 1;
  }
  uint32_t packed = 0, t = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {   // unconditional LDS reads + selects: no divergent control flow in the batch body
    const uint32_t ob = obs[sym[k]];
    const bool act = kFull || 4 * lane + k < cnt;
    t += act ? ob : 0u;
    packed |= (act ? sym[k] : 0u) << (8 * k);
  }
  const uint32_t tr = quad_transpose_u8(packed, lane & 3);
  if (kFull || 4 * lane < ((cnt + 15u) & ~15u)) *(u32_unaligned PCO_GLOBAL*)(sym_out + 4 * lane) = tr;   // whole 16-latent blocks
  const uint32_t total = wave_sum(t);
  if (lane == 0) *ob_bits_out = total;
}
template <class LV>
__device__ __forceinline__ void dissect_batch(const uint8_t PCO_LDS* vt, const LV PCO_GLOBAL* lat, uint8_t PCO_GLOBAL* sym_out, uint32_t PCO_GLOBAL* ob_bits_out,
                                              uint32_t cnt, uint32_t n_bins, uint32_t search_log, uint64_t minv, bool use_lut) {
  if (cnt == kBatchN) {
    if (use_lut) dissect_batch_t<LV, true, true>(vt, lat, sym_out, ob_bits_out, cnt, n_bins, search_log, minv);
    else dissect_batch_t<LV, false, true>(vt, lat, sym_out, ob_bits_out, cnt, n_bins, search_log, minv);
  } else {
    if (use_lut) dissect_batch_t<LV, true, false>(vt, lat, sym_out, ob_bits_out, cnt, n_bins, search_log, minv);
    else dissect_batch_t<LV, false, false>(vt, lat, sym_out, ob_bits_out, cnt, n_bins, search_log, minv);
  }
}

constexpr uint32_t kDisRuns = 4;   // runs per dissect block: the tables (bin lowers from the plan, the value -> bin table) are built once per block
template <class L>
__device__ __forceinline__ void dissect_block(const EncWorkspace& ws, const EncFast& fx, uint32_t p, uint32_t run0, EncPage PCO_GLOBAL* pg, EncChunk PCO_GLOBAL* ch) {
  const uint32_t run = run0;   // (first run of the block: the one that decides whether the block has anything to do)
  const uint32_t t = uni(pg->chunk);
  const uint32_t page_n = (uint32_t)uni((uint64_t)pg->n);
  const uint64_t pstart = uni((uint64_t)pg->start);
  uint8_t PCO_LDS* smem = enc_lds_base();
  PageVar pv[3];
  bool any = false;
#pragma unroll
  for (int v = 0; v < 3; v++) {
    pv[v] = page_var(ch, v, page_n);
    if (wd_takes(fx.fused, pv[v])) pv[v].present = 0;   // (enc_walkd_kernel finds this variable's symbols itself)
    if (pv[v].present && pv[v].n_bins > 1 && (uint64_t)run * kRunBatches * kBatchN < pv[v].n_lat) any = true;
  }
  if (!any) return;
  bool use_lut[3]; uint64_t rel0[3];   // rel0: what the element values are relative to (the minimum for compact latents, else 0)
#pragma unroll
  for (int v = 0; v < 3; v++) {
    use_lut[v] = false; rel0[v] = 0;
    if (!pv[v].present || pv[v].n_bins <= 1) continue;
    const PlanRef plan = plan_ref(ws, t, v);
    const uint64_t range = uni((uint64_t)ch->v[v].maxv) - pv[v].minv;
    use_lut[v] = range < kDirectHistRange;
    rel0[v] = pv[v].compact ? pv[v].rel : 0ull;
    uint8_t PCO_LDS* vt = smem + v * kDisVarBytes;
    const uint32_t b = threadIdx.x;  // 256 threads: one padded bin each; lowers as u64 relative to rel0, padded with the maximum
    ((uint64_t PCO_LDS*)vt)[b] = b < pv[v].n_bins ? (uint64_t)plan.blower()[b] - rel0[v] : ~0ull;
    (vt + 2048)[b] = b < pv[v].n_bins ? plan.bob()[b] : 0;
  }
  __syncthreads();
#pragma unroll
  for (int v = 0; v < 3; v++) {   // value -> bin table: each thread fills 16 consecutive values (one search, then a walk over the sorted lowers)
    if (!use_lut[v]) continue;
    uint8_t PCO_LDS* vt = smem + v * kDisVarBytes;
    const uint64_t PCO_LDS* low = (const uint64_t PCO_LDS*)vt;
    const uint32_t u0 = threadIdx.x * (kDirectHistRange / 256);
    const uint64_t x0 = pv[v].minv - rel0[v] + u0;   // value of table slot u0, in the units of `low`
    uint32_t sym = 0;
    { uint32_t lo = 0, hi = pv[v].n_bins; while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (low[mid] <= x0) lo = mid; else hi = mid; } sym = lo; }   // last bin with lower <= x0 (bin 0 below that)
    for (uint32_t k = 0; k < kDirectHistRange / 256; k++) {
      const uint64_t x = x0 + k;
      while (sym + 1 < pv[v].n_bins && low[sym + 1] <= x) sym++;
      (vt + 2048 + 256)[u0 + k] = (uint8_t)sym;
    }
  }
  __syncthreads();
  const uint32_t wave = threadIdx.x >> 6;
  for (uint32_t bb = wave; bb < kDisRuns * kRunBatches; bb += 4) {
    const uint32_t batch = run0 * kRunBatches + bb;
    const uint32_t base = batch * kBatchN;
#pragma unroll
    for (int v = 0; v < 3; v++) {
      if (!pv[v].present || pv[v].n_bins <= 1 || base >= pv[v].n_lat) continue;
      const uint32_t cnt = pv[v].n_lat - base < kBatchN ? pv[v].n_lat - base : kBatchN;
      uint32_t search_log = 0; while ((1u << search_log) < pv[v].n_bins) search_log++;
      const uint64_t at = pstart + pv[v].skip + base, fat = fast_at(pg, pv[v].skip) + base;
      uint32_t PCO_GLOBAL* ob_out = (uint32_t PCO_GLOBAL*)fx.bat + (((uint64_t)p * 3 + v) * fx.bat_stride + batch) * 2;
      const uint8_t PCO_LDS* vt = smem + v * kDisVarBytes;
      uint8_t PCO_GLOBAL* so = fsym_ptr(ws, fx, t, v) + fat;
      const uint64_t m0 = pv[v].minv - rel0[v];   // the table's slot 0 in element units
      if (pv[v].compact) dissect_batch<uint16_t>(vt, clat_ptr(ws, t, v) + at, so, ob_out, cnt, pv[v].n_bins, search_log, m0, use_lut[v]);
      else if (v == 0) dissect_batch<uint32_t>(vt, lat_ptr<uint32_t>(ws, t, 0) + at, so, ob_out, cnt, pv[v].n_bins, search_log, m0, use_lut[v]);
      else dissect_batch<L>(vt, lat_ptr<L>(ws, t, v) + at, so, ob_out, cnt, pv[v].n_bins, search_log, m0, use_lut[v]);
    }
  }
}

// grid pages * ceil(runs_per_page / kDisRuns), 256 threads
__global__ __launch_bounds__(256) void enc_dissect_kernel(EncWorkspace ws, EncFast fx, uint32_t n_pages) {
  const uint32_t bpp = (fx.runs_per_page + kDisRuns - 1) / kDisRuns;
  const uint32_t p = blockIdx.x / bpp, run = (blockIdx.x % bpp) * kDisRuns;
  if (p >= n_pages) return;
  EncPage PCO_GLOBAL* pg = (EncPage PCO_GLOBAL*)ws.pages + p;
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + uni(pg->chunk);
  if (!page_is_fast(ch, pg)) return;
  const int bits = dtype_bits(uni(ch->dtype));
  if (bits == 64) dissect_block<uint64_t>(ws, fx, p, run, pg, ch);
  else if (bits == 32) dissect_block<uint32_t>(ws, fx, p, run, pg, ch);
  else if (bits == 16) dissect_block<uint16_t>(ws, fx, p, run, pg, ch);
  else dissect_block<uint8_t>(ws, fx, p, run, pg, ch);
}

// =========================================================================================================
// reverse tANS walk
// =========================================================================================================
// one reverse step of one chain (ans/encoding.rs:65-87).  info = D | row << 32 with D = ((min_renorm_bits + 1) << 16) - cutoff,
// so that bits = min_renorm_bits + (state >= cutoff) = (state + D) >> 16 (states are below 2^13), and row = the LDS address
// of the symbol's next-state row.  The dependent stretch is v_add, v_lshrrev, v_lshrrev, v_lshl_add -> ds_read_u16; the value
// and the bit count are taken care of after that read has been issued (the wave has nothing else to hide its latency behind).
__device__ __forceinline__ uint32_t ew_step(uint32_t& state, uint32_t& bits_acc, uint64_t info) {
  const uint32_t d = (uint32_t)info, row = (uint32_t)(info >> 32);
  const uint32_t old = state;
  const uint32_t bits = (old + d) >> 16;
  state = *(const uint16_t PCO_LDS*)(uintptr_t)(row + ((old >> bits) << 1));
  __builtin_amdgcn_sched_barrier(0);
  bits_acc += bits;
  return (bits << 12) | __builtin_amdgcn_ubfe(old, 0u, bits);
}

// stage: 0 = every item; 1 = only the items that fit the 16-per-wave slots; 2 = only those that do not
template <uint32_t kEwQ>
__global__ __launch_bounds__(64) void enc_walk_kernel(EncWorkspace ws, EncFast fx, uint32_t n_pages, uint32_t stage) {
  constexpr uint32_t kEwSlotBytes = EwCfg<kEwQ>::kSlotBytes, kEwSymOff = EwCfg<kEwQ>::kSymOff, kEwNsOff = 0;
  const uint32_t lane = lane_id();
  const uint32_t slot = lane >> 2, j = lane & 3;
  uint8_t PCO_LDS* smem = enc_lds_base();
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t n_items = n_pages * ws.n_slots;
  // ---- phase 0: tables of the wave's items into LDS (all lanes cooperate, one item at a time) ----
  uint32_t my_n_lat = 0, my_T = 0, my_p = 0, my_v = 0, my_info_off = 0;
  uint64_t my_at = 0; uint32_t my_task = 0;
  for (uint32_t q = 0; q < kEwQ; q++) {
    const uint32_t item = blockIdx.x * kEwQ + q;
    if (item >= n_items) break;
    const uint32_t p = item / ws.n_slots, s = item % ws.n_slots;
    const uint32_t v = ws.slot_of_var[0] == s ? 0u : (ws.slot_of_var[1] == s ? 1u : 2u);
    EncPage PCO_GLOBAL* pg = (EncPage PCO_GLOBAL*)ws.pages + p;
    const uint32_t t = uni(pg->chunk);
    EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
    if (!page_is_fast(ch, pg)) continue;
    const uint32_t page_n = (uint32_t)uni((uint64_t)pg->n);
    const PageVar pv = page_var(ch, v, page_n);
    if (!pv.present || pv.n_bins <= 1 || pv.n_lat == 0) {
      if (pv.present && lane < 4 && stage != 2) fx.fstate[((uint64_t)p * 3 + v) * 4 + lane] = 1u << pv.asl;
      continue;
    }
    if (wd_walks(fx.fused, pv) || ws_walks(fx.fused, pv)) continue;             // enc_walkd_kernel's / enc_walkseg_kernel's item
    if (stage != 0 && ew_fits16(pv.asl, pv.n_bins) != (stage == 1)) continue;   // the other stage's item
    const uint32_t kEwInfoOff = ew_info_off(pv.asl);
    const PlanRef plan = plan_ref(ws, t, v);
    uint8_t PCO_LDS* sl = smem + q * kEwSlotBytes;
    const uint32_t T = 1u << pv.asl;
    for (uint32_t i = lane; i < T; i += 64) ((uint16_t PCO_LDS*)(sl + kEwNsOff))[i] = plan.next_states()[i];
    for (uint32_t b = lane; b < pv.n_bins; b += 64) {
      const uint32_t si = plan.syminfo()[b];   // cutoff(14) | min_renorm_bits(4) << 14 | (row + 8192)(14) << 18
      const uint32_t cutoff = si & 0x3fffu, minb = (si >> 14) & 15u, row = (si >> 18) - 8192u;   // row may be "negative": wraps mod 2^32 below
      const uint32_t row_addr = lds0 + q * kEwSlotBytes + kEwNsOff + 2u * row;
      ((uint64_t PCO_LDS*)(sl + kEwInfoOff))[b] = (uint64_t)(((minb + 1u) << 16) - cutoff) | ((uint64_t)row_addr << 32);
    }
    if (slot == q) { my_n_lat = pv.n_lat; my_T = T; my_p = p; my_v = v; my_task = t; my_at = fast_at(pg, pv.skip); my_info_off = kEwInfoOff; }
  }
  enc_wave_sync();
  // (wave-level vote at a converged point: idle quads -- trivial variables, the other stage's items, slots beyond the
  // launch -- must stay in the wave, the butterfly maximum over n_full below reads every lane.  Round 1 voted inside
  // `if (my_n_lat == 0)`, where __all only sees the idle lanes: they left, and a wave whose FIRST live quad had fewer
  // full batches than a later one stopped walking early -- the wrong secondary stream of sweep seed 2025 case 188.)
  if (__all(my_n_lat == 0)) return;
  // ---- phase 1: batches in reverse ----
  const uint32_t slice = lds0 + (slot < kEwQ ? slot : 0u) * kEwSlotBytes;
  const uint32_t info_addr = slice + my_info_off, symbuf = slice + kEwSymOff;
  const uint8_t PCO_GLOBAL* gsym = (const uint8_t PCO_GLOBAL*)fsym_ptr(ws, fx, my_task, my_v) + my_at;
  uint16_t PCO_GLOBAL* gans = fansw_ptr(ws, fx, my_task, my_v) + my_at;
  uint32_t PCO_GLOBAL* gbat = (uint32_t PCO_GLOBAL*)fx.bat + ((uint64_t)my_p * 3 + my_v) * fx.bat_stride * 2;
  const uint32_t n_batches = (my_n_lat + kBatchN - 1) / kBatchN;
  uint32_t state = my_T;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;
  typedef uint64_t __attribute__((aligned(2))) u64_align2;
  // the last (possibly partial) batch: staged directly, walked with per-step predicates
  if (n_batches > 0) {
    const uint32_t b = n_batches - 1, base = b * kBatchN, cnt = my_n_lat - base;
    const uint32_t blocks_bytes = (cnt + 15u) & ~15u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t off = 64 * j + 16 * k;
      if (off < blocks_bytes) *(u32x4 PCO_LDS*)(uintptr_t)(symbuf + (b & 1) * 256 + off) = *(const u32x4_unaligned PCO_GLOBAL*)(gsym + base + off);
    }
  }
  enc_wave_sync();
  if (n_batches > 0) {
    const uint32_t b = n_batches - 1, base = b * kBatchN, cnt = my_n_lat - base;
    const uint32_t steps = (cnt + 3) >> 2;
    uint32_t bits_acc = 0;
    for (uint32_t blk = (steps + 3) >> 2; blk-- > 0;) {
      const uint32_t sd = *(const uint32_t PCO_LDS*)(uintptr_t)(symbuf + (b & 1) * 256 + 16 * blk + 4 * j);
      uint64_t out = 0;
#pragma unroll
      for (int k = 3; k >= 0; k--) {
        const uint32_t g = 4 * blk + k;
        if (4 * g + j < cnt) {
          const uint64_t info = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * ((sd >> (8 * k)) & 0xffu));
          out |= (uint64_t)ew_step(state, bits_acc, info) << (16 * k);
        }
      }
      *(u64_align2 PCO_GLOBAL*)(gans + base + 16 * blk + 4 * j) = out;
    }
    bits_acc += quad_dpp<0xB1>(bits_acc); bits_acc += quad_dpp<0x4E>(bits_acc);
    if (j == 0) gbat[(uint64_t)b * 2 + 1] = bits_acc;
  }
  // full batches, newest first; the next batch's symbols are fetched while the current one is walked
  const uint32_t n_full = n_batches > 0 ? n_batches - 1 : 0;
  uint32_t max_full = n_full;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(max_full, d, 64); max_full = max_full > o ? max_full : o; }
  max_full = uni(max_full);
  u32x4 pre[4];
  if (n_full > 0) {
#pragma unroll
    for (int k = 0; k < 4; k++) pre[k] = *(const u32x4_unaligned PCO_GLOBAL*)(gsym + (uint64_t)(n_full - 1) * kBatchN + 64 * j + 16 * k);
  }
  for (uint32_t it = 0; it < max_full; it++) {
    const bool act = it < n_full;
    const uint32_t b = act ? n_full - 1 - it : 0u;
    const uint32_t buf = symbuf + (b & 1) * 256;
    if (act) {
#pragma unroll
      for (int k = 0; k < 4; k++) *(u32x4 PCO_LDS*)(uintptr_t)(buf + 64 * j + 16 * k) = pre[k];
    }
    enc_wave_sync();
    if (act && b > 0) {
#pragma unroll
      for (int k = 0; k < 4; k++) pre[k] = *(const u32x4_unaligned PCO_GLOBAL*)(gsym + (uint64_t)(b - 1) * kBatchN + 64 * j + 16 * k);
    }
    if (act) {
      uint32_t bits_acc = 0;
      uint16_t PCO_GLOBAL* ga = gans + (uint64_t)b * kBatchN + 4 * j;
      // software pipeline: block blk-1's four info words are fetched one per step, each in the shadow of a state read of
      // block blk, and block blk-2's symbol dword with the last of them
      uint32_t sd = *(const uint32_t PCO_LDS*)(uintptr_t)(buf + 16 * 15 + 4 * j);
      uint32_t nsd = *(const uint32_t PCO_LDS*)(uintptr_t)(buf + 16 * 14 + 4 * j);
      uint64_t i0 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * (sd & 0xffu));
      uint64_t i1 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * ((sd >> 8) & 0xffu));
      uint64_t i2 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * ((sd >> 16) & 0xffu));
      uint64_t i3 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * (sd >> 24));
      for (uint32_t blk = 16; blk-- > 0;) {
        const uint32_t nnblk = blk > 1 ? blk - 2 : 0;
        const uint32_t o3 = ew_step(state, bits_acc, i3);
        const uint64_t n3 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * (nsd >> 24));
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t o2 = ew_step(state, bits_acc, i2);
        const uint64_t n2 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * ((nsd >> 16) & 0xffu));
        const uint32_t o23 = o2 | (o3 << 16);
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t o1 = ew_step(state, bits_acc, i1);
        const uint64_t n1 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * ((nsd >> 8) & 0xffu));
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t o0 = ew_step(state, bits_acc, i0);
        const uint64_t n0 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * (nsd & 0xffu));
        nsd = *(const uint32_t PCO_LDS*)(uintptr_t)(buf + 16 * nnblk + 4 * j);
        *(u64_align2 PCO_GLOBAL*)(ga + 16 * blk) = (uint64_t)(o0 | (o1 << 16)) | ((uint64_t)o23 << 32);
        __builtin_amdgcn_sched_barrier(0);
        i0 = n0; i1 = n1; i2 = n2; i3 = n3;
      }
      bits_acc += quad_dpp<0xB1>(bits_acc); bits_acc += quad_dpp<0x4E>(bits_acc);
      if (j == 0) gbat[(uint64_t)b * 2 + 1] = bits_acc;
    }
    enc_wave_sync();
  }
  if (my_n_lat > 0 && slot < kEwQ) fx.fstate[((uint64_t)my_p * 3 + my_v) * 4 + j] = state;
}

// Gathering waves per walk block.  One was right while every lookup went through the texture path (a second changed nothing: the path's rate
// is the bound there).  With the value -> bin tables in LDS (homogeneous blocks, the headline's case) the gathering wave is a chain of LDS
// round trips per item, and the sixteen items split over more waves: enc_walkd_kernel 4.64 ms per 8192 chunks with one, 3.65 with two,
// 3.47 with four (the walk alone: 3.4).  Blocks of mixed items, which keep the texture path, lose a little with four (the mixed stream: 7.4 /
// 7.1 / 7.6 ms).
#ifndef PCO_WD_HELPERS
#define PCO_WD_HELPERS 4
#endif
// =========================================================================================================
// walk + dissect in one block
// =========================================================================================================
// The symbols of a variable whose latents are 16-bit and span fewer than 4096 values need not pass through HBM on their way to the
// walk.  enc_vlut_kernel writes the variable's value -> (bin | offset bits << 8) table once (<= 8 KB, L2-resident); the second wave
// of the walker's block gathers from it, one batch ahead of the walk: it leaves the symbols in the LDS buffer the walker reads and
// (for enc_pack_kernel) in the symbol scratch, and adds up the batch's offset bits.  16 items per block, 4.5 KB of LDS each
// (enc_walk_kernel<8>'s slot): two blocks per CU -- the walker's chain of dependent steps and, beside it, the gathering waves (texture path or
// LDS reads: nothing the walker's steps wait for).
constexpr uint32_t kWdHelpers = PCO_WD_HELPERS, kWdH = 16 / kWdHelpers;   // gathering waves per block, items per gathering wave
constexpr uint32_t kWdQ = 16, kWdSlot = EwCfg<8>::kSlotBytes, kWdSymOff = EwCfg<8>::kSymOff, kWdLdsBytes = kWdQ * kWdSlot;
static_assert(2 * kWdLdsBytes <= 160 * 1024, "two blocks per CU");

// grid = chunks * 3, 256 threads: the table of one variable (each thread 16 consecutive values: one search, then a walk over the sorted lowers)
__global__ __launch_bounds__(256) void enc_vlut_kernel(EncWorkspace ws, EncFast fx, uint32_t n_tasks) {
  const uint32_t t = blockIdx.x / 3, v = blockIdx.x % 3;
  if (t >= n_tasks) return;
  const EncChunk PCO_GLOBAL* ch = (const EncChunk PCO_GLOBAL*)ws.chunks + t;
  if (uni(ch->status) != PCO_GFX_OK || uni(ch->fast_ok) == 0) return;
  const PageVar pv = page_var(ch, v, 0xffffffffu);
  PageVar any = pv; any.n_lat = 1;   // (the page length does not matter here)
  if (!wd_takes(fx.fused, any)) return;
  __shared__ uint64_t low[256]; __shared__ uint8_t obs[256];
  const PlanRef plan = plan_ref(ws, t, v);
  const uint32_t b = threadIdx.x;
  low[b] = b < pv.n_bins ? (uint64_t)plan.blower()[b] - pv.minv : ~0ull;   // relative to the minimum: table slot = value - min
  obs[b] = b < pv.n_bins ? plan.bob()[b] : 0;
  __syncthreads();
  uint16_t PCO_GLOBAL* lut = vlut_ptr(ws, fx, t, v);
  const uint32_t m0 = (uint32_t)(pv.minv - pv.rel) + vlut_rot(t * ws.n_slots + ws.slot_of_var[v]);   // the minimum as a 16-bit latent, plus the table's rotation
  const uint32_t u0 = threadIdx.x * (kDirectHistRange / 256);
  if (u0 > pv.range) return;
  uint32_t sym = 0;
  { uint32_t lo = 0, hi = pv.n_bins; while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (low[mid] <= u0) lo = mid; else hi = mid; } sym = lo; }
  for (uint32_t k = 0; k < kDirectHistRange / 256; k++) {
    const uint64_t x = (uint64_t)u0 + k;
    while (sym + 1 < pv.n_bins && low[sym + 1] <= x) sym++;
    lut[(m0 + u0 + k) & (kDirectHistRange - 1)] = (uint16_t)(sym | ((uint32_t)obs[sym] << 8));   // slot = the 16-bit latent mod 4096
  }
}

// the block's barrier of the walk loop: LDS traffic only is waited for -- the walker's stores and the gathering wave's loads stay in flight across it
// (__syncthreads drains vmcnt: every batch would wait for HBM)
__device__ __forceinline__ void wd_barrier() { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(64 * (1 + kWdHelpers)) void enc_walkd_kernel(EncWorkspace ws, EncFast fx, uint32_t n_pages) {
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  uint8_t PCO_LDS* smem = enc_lds_base();
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t n_items = n_pages * ws.n_slots;
  if (fx.wp_block != nullptr && uni(fx.wp_block[blockIdx.x]) != 0) return;   // walked AND packed by enc_walkp_kernel
  typedef uint64_t __attribute__((aligned(2))) u64_align2;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;
  // ---- phase 0: the items' tables (wave 0: next states + info, as enc_walk_kernel).  A walker lane keeps the item of its quad
  //      (slot = lane >> 2), a gathering lane the item lane & 15 (handed round with v_readlane) ----
  const uint32_t my_q = wave == 0 ? lane >> 2 : (wave - 1) * kWdH + (lane & (kWdH - 1));   // (gathering wave w takes the items (w - 1) * kWdH ..)
  uint32_t my_n_lat = 0, my_T = 0, my_p = 0, my_v = 0, my_task = 0, my_info_off = 0, my_m0 = 0, my_finds = 0;   // (a gathering lane's n_lat is 0 for an item it has nothing to find for)
  uint64_t my_at = 0, my_clat = 0;
  uint32_t max_nb = 0;   // batches of the block's longest item: what every wave's loop runs to (uniform)
  // Value -> bin through LDS (kWdLdsLut): between an item's info words and its symbol buffers the slot has room to spare (1.9 KB at
  // ans_size_log 10 with 20 bins); a variable whose values span at most that many bytes keeps its value -> bin table THERE, a byte per value
  // (copied once from enc_vlut_kernel's table), with the bins' offset-bit counts behind it -- and the gathering wave's four texture-path
  // gathers per lane and item (~88 cycles each whatever the table's size) become eight LDS byte reads.  Taken when EVERY item of the block
  // qualifies (uniform code: a branch per item around a load makes the compiler wait for the load where it is issued).  Worth 0.1 ms of
  // 4.8 on the headline workload: what bounded the gathering wave was not the texture path but the sixteen items' dependent chains, each
  // behind a branch of its own (see the straight-line form below).
  bool all_lds = true; uint32_t my_vtab = 0, my_obtab = 0, my_range = 0;
  for (uint32_t q = 0; q < kWdQ; q++) {
    const uint32_t item = blockIdx.x * kWdQ + q;
    if (item >= n_items) break;
    const uint32_t p = item / ws.n_slots, sl = item % ws.n_slots;
    const uint32_t v = ws.slot_of_var[0] == sl ? 0u : (ws.slot_of_var[1] == sl ? 1u : 2u);
    EncPage PCO_GLOBAL* pg = (EncPage PCO_GLOBAL*)ws.pages + p;
    const uint32_t t = uni(pg->chunk);
    EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
    if (!page_is_fast(ch, pg)) continue;
    const uint32_t page_n = (uint32_t)uni((uint64_t)pg->n);
    const PageVar pv = page_var(ch, v, page_n);
    if (!wd_walks(fx.fused, pv)) {   // (with every walked item here -- fused == 1 -- nobody else writes the states of the variables that need no walk)
      if ((fx.fused & 0xffu) == 1 && pv.present && (pv.n_bins <= 1 || pv.n_lat == 0) && wave == 0 && lane < 4) fx.fstate[((uint64_t)p * 3 + v) * 4 + lane] = 1u << pv.asl;
      continue;
    }
    if (ws_walks(fx.fused, pv)) continue;        // enc_walkseg_kernel's item
    const bool finds = wd_takes(fx.fused, pv);   // its symbols come from the gathering wave; else from enc_dissect_kernel, staged by the walker itself
    const uint32_t info_off = ew_info_off(pv.asl);
    { const uint32_t nbq = (pv.n_lat + kBatchN - 1) / kBatchN; max_nb = nbq > max_nb ? nbq : max_nb; }
    if (wave == 0) {
      const PlanRef plan = plan_ref(ws, t, v);
      uint8_t PCO_LDS* slot = smem + q * kWdSlot;
      const uint32_t T = 1u << pv.asl;
      for (uint32_t i = lane; i < T; i += 64) ((uint16_t PCO_LDS*)slot)[i] = plan.next_states()[i];
      for (uint32_t b = lane; b < pv.n_bins; b += 64) {
        const uint32_t si = plan.syminfo()[b];   // cutoff(14) | min_renorm_bits(4) << 14 | (row + 8192)(14) << 18
        const uint32_t cutoff = si & 0x3fffu, minb = (si >> 14) & 15u, row = (si >> 18) - 8192u;
        const uint32_t row_addr = lds0 + q * kWdSlot + 2u * row;
        ((uint64_t PCO_LDS*)(slot + info_off))[b] = (uint64_t)(((minb + 1u) << 16) - cutoff) | ((uint64_t)row_addr << 32);
      }
    }
    // the slot's slack: [info_off + 8 n_bins, kWdSymOff): offset bits u8[n_bins] | bins u8[range + 1]
    const uint32_t ob_off = info_off + 8u * pv.n_bins, vt_off = ob_off + ((pv.n_bins + 3u) & ~3u);
    const bool fits = finds && pv.n_bins <= 256 && pv.range < 4096 && vt_off + (uint32_t)pv.range + 1u <= kWdSymOff;
    if (!(finds && fits)) all_lds = false;   // (a block of mixed items keeps the global tables for all of them: measured on the mixed stream, the LDS reads of sixteen items for the sake of five cost more than the gathers -- 8.3 against 7.4 ms)
    if (all_lds && wave == 1) {   // (the first gathering wave fills the tables of all sixteen items; kWdHelpers > 1: the others wait at the barrier below)
      const PlanRef plan = plan_ref(ws, t, v);
      uint8_t PCO_LDS* slot = smem + q * kWdSlot;
      for (uint32_t b = lane; b < pv.n_bins; b += 64) slot[ob_off + b] = (uint8_t)plan.bob()[b];
      const uint16_t PCO_GLOBAL* lut = vlut_ptr(ws, fx, t, v);
      const uint32_t base = (uint32_t)(pv.minv - pv.rel) + vlut_rot(t * ws.n_slots + ws.slot_of_var[v]);
      for (uint32_t i = lane; i <= (uint32_t)pv.range; i += 64) slot[vt_off + i] = (uint8_t)lut[(base + i) & (kDirectHistRange - 1)];
    }
    if (my_q == q) {
      my_vtab = lds0 + q * kWdSlot + (fits ? vt_off : 0u); my_obtab = lds0 + q * kWdSlot + (fits ? ob_off : 0u); my_range = fits ? (uint32_t)pv.range : 0u;   // (an item without a table: every index reads the slot's first byte, and nothing looks at it)
      my_finds = finds ? 1u : 0u;
      my_n_lat = wave == 0 || finds ? pv.n_lat : 0u; my_T = 1u << pv.asl; my_p = p; my_v = v; my_task = t; my_info_off = info_off; my_m0 = (uint32_t)(pv.minv - pv.rel);
      my_at = fast_at(pg, pv.skip); my_clat = uni((uint64_t)pg->start) + pv.skip;
    }
  }
  const uint32_t my_nb = (my_n_lat + kBatchN - 1) / kBatchN;
  if (max_nb == 0) return;
  wd_barrier();
  if (wave != 0) {
    const uint32_t q0 = (wave - 1) * kWdH;   // this wave's first item
    // ================= the gathering wave: batch nb - 1 - it of every item, for it = 0 .. max_nb - 1, each one barrier ahead of the walk =================
    // What depends on the item alone stays in the lane that holds the item (lane & 15): the batch's address, its length; the wave takes the
    // items in turn and fetches the two or three scalars it needs with v_readlane.  A lane without an item points at the start of the
    // table area: its loads are issued like everybody's and never used.
    const bool mine = my_n_lat != 0;
    const uint64_t my_clat_p = mine ? (uint64_t)(uintptr_t)(clat_ptr(ws, my_task, my_v) + my_clat) : (uint64_t)(uintptr_t)fx.vlut;
    uint8_t PCO_GLOBAL* my_gsym = mine ? fsym_ptr(ws, fx, my_task, my_v) + my_at : (uint8_t PCO_GLOBAL*)nullptr;
    uint32_t PCO_GLOBAL* my_gbat = (uint32_t PCO_GLOBAL*)fx.bat + (mine ? (uint64_t)(my_p * 3 + my_v) * fx.bat_stride * 2 : 0ull);
    const uint32_t my_lut_off = mine ? (my_task * ws.n_slots + ws.slot_of_var[my_v]) * kDirectHistRange : 0u;   // (u16 elements)
    const uint32_t my_rot2 = vlut_rot(my_lut_off / kDirectHistRange) * 0x10001u;                                     // the table's rotation, for both 16-bit halves of a dword (no carry: latents < 2^15)
    auto bcast = [](uint32_t x, uint32_t q) { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)q); };
    auto bcast64 = [&](uint64_t x, uint32_t q) { return ((uint64_t)bcast((uint32_t)(x >> 32), q) << 32) | bcast((uint32_t)x, q); };
    auto batch_of = [&](uint32_t it) { return it < my_nb ? my_nb - 1 - it : 0u; };   // (no batch at this step: batch 0 is read, and never used)
    // (branch-free on purpose: with a branch per item the compiler waited for every load where it was issued -- sixteen HBM round trips
    //  per batch.  The 8 bytes of a page's last, partial batch may run into the scratch behind the page: those latents count for nothing)
    auto load_batches = [&](uint32_t it, uint64_t (&w)[kWdH]) {   // the 4 latents a lane owns of every item's batch: 16 loads in flight
      const uint64_t my_src = my_clat_p + 2ull * batch_of(it) * kBatchN;
#pragma unroll
      for (uint32_t q = 0; q < kWdH; q++) w[q] = __builtin_nontemporal_load((const u64_align2 PCO_GLOBAL*)((const uint16_t PCO_GLOBAL*)(uintptr_t)bcast64(my_src, q) + 4 * lane));   // (streamed once: must not push the tables out of L2)
    };
    auto gather = [&](const uint64_t (&w)[kWdH], uint32_t (&e)[kWdH][4]) {   // (branch-free, as the loads)
#pragma unroll
      for (uint32_t q = 0; q < kWdH; q++) {
        const uint16_t PCO_GLOBAL* lut = (const uint16_t PCO_GLOBAL*)fx.vlut + bcast(my_lut_off, q);   // (uniform base + 32-bit lane offset)
        const uint32_t rot2 = bcast(my_rot2, q);
        const uint32_t lo = (uint32_t)w[q] + rot2, hi = (uint32_t)(w[q] >> 32) + rot2;
        // the table is indexed by the 16-bit latent mod 4096 (a window of fewer than 4096 consecutive values: no two share a slot; nothing
        // reads beyond the table whatever the scratch holds)
        e[q][0] = lut[lo & (kDirectHistRange - 1)]; e[q][1] = lut[(lo >> 16) & (kDirectHistRange - 1)];
        e[q][2] = lut[hi & (kDirectHistRange - 1)]; e[q][3] = lut[(hi >> 16) & (kDirectHistRange - 1)];
      }
    };
    // the same from the slots' own tables (all_lds): index = latent - the variable's minimum (as a 16-bit latent), clamped into the table
    // (what lies behind a page's last batch is never looked at, but must not index out of the slot)
    auto gather_lds = [&](const uint64_t (&w)[kWdH], uint32_t (&e)[kWdH][4]) {
#pragma unroll
      for (uint32_t q = 0; q < kWdH; q++) {
        const uint32_t vt = bcast(my_vtab, q), ot = bcast(my_obtab, q), rg = bcast(my_range, q), m0 = bcast(my_m0, q);
        const uint32_t lo = (uint32_t)w[q], hi = (uint32_t)(w[q] >> 32);
        uint32_t idx[4] = {(lo & 0xffffu) - m0, (lo >> 16) - m0, (hi & 0xffffu) - m0, (hi >> 16) - m0};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          idx[k] = idx[k] > rg ? 0u : idx[k];
          const uint32_t bin = *(const uint8_t PCO_LDS*)(uintptr_t)(vt + idx[k]);
          e[q][k] = bin | ((uint32_t)*(const uint8_t PCO_LDS*)(uintptr_t)(ot + bin) << 8);
        }
      }
    };
    // software pipeline: at step `it` the table entries of step it + 1 are gathered (from latents loaded during step it - 1) and the
    // latents of step it + 2 requested, in that order -- loads return in order, so nothing below waits for HBM -- while the entries gathered
    // during step it - 1 are turned into symbols
    uint64_t wnxt[kWdH]; uint32_t e[kWdH][4], enxt[kWdH][4];
    const bool lookups = (fx.fused & kFusedLookups) != 0;   // (no tables in this call: the wave only keeps the walker's barriers company)
    if (lookups) {
      load_batches(0, wnxt);
      if (all_lds) gather_lds(wnxt, enxt); else gather(wnxt, enxt);
      if (1 < max_nb) load_batches(1, wnxt);
    }
    for (uint32_t it = 0; it <= max_nb; it++) {   // it == max_nb: nothing left to find, only the barrier
      if (it < max_nb && lookups) {
#pragma unroll
        for (uint32_t q = 0; q < kWdH; q++) { e[q][0] = enxt[q][0]; e[q][1] = enxt[q][1]; e[q][2] = enxt[q][2]; e[q][3] = enxt[q][3]; }
        __builtin_amdgcn_sched_barrier(0);
        if (it + 1 < max_nb) { if (all_lds) gather_lds(wnxt, enxt); else gather(wnxt, enxt); }
        __builtin_amdgcn_sched_barrier(0);
        if (it + 2 < max_nb) load_batches(it + 2, wnxt);
        __builtin_amdgcn_sched_barrier(0);
        const bool my_on = it < my_nb;
        const uint32_t my_hb = batch_of(it), my_base = my_hb * kBatchN;
        const uint32_t my_cnt = my_on ? (my_n_lat - my_base < kBatchN ? my_n_lat - my_base : kBatchN) : 0u;
        const uint32_t my_buf = lds0 + my_q * kWdSlot + kWdSymOff + (my_hb & 1) * 256;   // this step's symbol buffer of the lane's item
        uint32_t my_total = 0;
        if (__all(my_cnt == kBatchN || my_cnt == 0)) {
          // every item the wave looks up has a full batch at this step (all but a page's last): straight-line code -- no branch per item, so that the sixteen
          // items' chains (table reads, byte shuffles, a six-step sum over the wave, a quad transpose) are scheduled into one another
          // instead of each paying its latencies alone
#pragma unroll
          for (uint32_t q = 0; q < kWdH; q++) {
            const uint32_t e01 = e[q][0] | (e[q][1] << 16), e23 = e[q][2] | (e[q][3] << 16);
            const uint32_t packed = __builtin_amdgcn_perm(e23, e01, 0x06040200u);               // the four bin bytes
            const uint32_t obs4 = __builtin_amdgcn_perm(e23, e01, 0x07050301u);                 // the four offset-bit counts (<= 64 each)
            const uint32_t total = wave_sum(__builtin_amdgcn_sad_u8(obs4, 0u, 0u));
            my_total = (lane & (kWdH - 1)) == q ? total : my_total;
            const uint32_t tr = quad_transpose_u8(packed, lane & 3);
            if (bcast(my_cnt, q) != 0) *(uint32_t PCO_LDS*)(uintptr_t)(bcast(my_buf, q) + 4 * lane) = tr;   // (an item the wave does not look up: its buffer is the walker's own)
          }
        } else {
#pragma unroll
        for (uint32_t q = 0; q < kWdH; q++) {
          const uint32_t cnt = bcast(my_cnt, q);
          if (cnt == 0) continue;
          if (cnt < kBatchN) {   // (the page's last batch: a latent beyond it is bin 0 with no bits)
#pragma unroll
            for (int k = 0; k < 4; k++) e[q][k] = 4 * lane + k < cnt ? e[q][k] : 0u;
          }
          const uint32_t e01 = e[q][0] | (e[q][1] << 16), e23 = e[q][2] | (e[q][3] << 16);
          const uint32_t packed = __builtin_amdgcn_perm(e23, e01, 0x06040200u);               // the four bin bytes
          const uint32_t obs4 = __builtin_amdgcn_perm(e23, e01, 0x07050301u);                 // the four offset-bit counts (<= 64 each)
          const uint32_t total = wave_sum(__builtin_amdgcn_sad_u8(obs4, 0u, 0u));
          my_total = (lane & (kWdH - 1)) == q ? total : my_total;
          *(uint32_t PCO_LDS*)(uintptr_t)(bcast(my_buf, q) + 4 * lane) = quad_transpose_u8(packed, lane & 3);
        }
        }
        // the symbols go on to enc_pack_kernel's scratch from the LDS buffers: a lane copies a quarter (64 bytes) of its own item's batch,
        // whole 16-latent blocks as enc_dissect_kernel writes them; lanes 0..15 leave the batch's offset-bit total
        if (my_on) {
          constexpr uint32_t kParts = 64 / kWdH, kPer = 16 / kParts;   // lanes per item, 16-byte blocks per lane
          const uint32_t part = lane / kWdH, blocks = (my_cnt + 15u) >> 4;
#pragma unroll
          for (uint32_t r = 0; r < kPer; r++) {
            const uint32_t blk = part * kPer + r;
            if (blk < blocks) *(u32x4_unaligned PCO_GLOBAL*)(my_gsym + my_base + 16 * blk) = *(const u32x4 PCO_LDS*)(uintptr_t)(my_buf + 16 * blk);
          }
          if (lane < kWdH) my_gbat[(uint64_t)my_hb * 2] = my_total;
        }
      }
      wd_barrier();
    }
    return;
  }
  // ================= the walker wave (enc_walk_kernel's walk, its symbols already in LDS) =================
  const uint32_t j = lane & 3;
  const uint32_t slice = lds0 + my_q * kWdSlot;
  const uint32_t info_addr = slice + my_info_off, symbuf = slice + kWdSymOff;
  uint16_t PCO_GLOBAL* gans = fansw_ptr(ws, fx, my_task, my_v) + my_at;
  uint32_t PCO_GLOBAL* gbat = (uint32_t PCO_GLOBAL*)fx.bat + ((uint64_t)my_p * 3 + my_v) * fx.bat_stride * 2;
  uint32_t state = my_T;
  // an item whose symbols enc_dissect_kernel wrote: the walker's quad stages them itself, a batch ahead, as enc_walk_kernel does
  const bool stages = my_n_lat != 0 && my_finds == 0;
  const uint8_t PCO_GLOBAL* gsym = (const uint8_t PCO_GLOBAL*)fsym_ptr(ws, fx, my_task, my_v) + my_at;
  u32x4 pre[4];
  auto fetch_syms = [&](uint32_t b) {   // whole 16-latent blocks of batch b
    const uint32_t cnt = my_n_lat - b * kBatchN < kBatchN ? my_n_lat - b * kBatchN : kBatchN, blocks_bytes = (cnt + 15u) & ~15u;
#pragma unroll
    for (int k = 0; k < 4; k++) { pre[k] = u32x4{0, 0, 0, 0}; if (64 * j + 16 * k < blocks_bytes) pre[k] = *(const u32x4_unaligned PCO_GLOBAL*)(gsym + (uint64_t)b * kBatchN + 64 * j + 16 * k); }
  };
  if (stages) fetch_syms(my_nb - 1);
  wd_barrier();   // (the gathering wave's it = 0)
  for (uint32_t it = 0; it < max_nb; it++) {
    if (stages && it < my_nb) {
      const uint32_t b = my_nb - 1 - it;
#pragma unroll
      for (int k = 0; k < 4; k++) *(u32x4 PCO_LDS*)(uintptr_t)(symbuf + (b & 1) * 256 + 64 * j + 16 * k) = pre[k];
      if (b > 0) fetch_syms(b - 1);
    }
    enc_wave_sync();
    if (it < my_nb) {
      const uint32_t b = my_nb - 1 - it, base = b * kBatchN, cnt = my_n_lat - base < kBatchN ? my_n_lat - base : kBatchN;
      const uint32_t buf = symbuf + (b & 1) * 256;
      uint32_t bits_acc = 0;
      if (cnt < kBatchN) {   // the last (partial) batch: per-step predicates
        const uint32_t steps = (cnt + 3) >> 2;
        for (uint32_t blk = (steps + 3) >> 2; blk-- > 0;) {
          const uint32_t sd = *(const uint32_t PCO_LDS*)(uintptr_t)(buf + 16 * blk + 4 * j);
          uint64_t out = 0;
#pragma unroll
          for (int k = 3; k >= 0; k--) {
            const uint32_t g = 4 * blk + k;
            if (4 * g + j < cnt) {
              const uint64_t info = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * ((sd >> (8 * k)) & 0xffu));
              out |= (uint64_t)ew_step(state, bits_acc, info) << (16 * k);
            }
          }
          *(u64_align2 PCO_GLOBAL*)(gans + base + 16 * blk + 4 * j) = out;
        }
      } else {               // a full batch, software-pipelined as in enc_walk_kernel
        uint16_t PCO_GLOBAL* ga = gans + (uint64_t)b * kBatchN + 4 * j;
        uint32_t sd = *(const uint32_t PCO_LDS*)(uintptr_t)(buf + 16 * 15 + 4 * j);
        uint32_t nsd = *(const uint32_t PCO_LDS*)(uintptr_t)(buf + 16 * 14 + 4 * j);
        uint64_t i0 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * (sd & 0xffu));
        uint64_t i1 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * ((sd >> 8) & 0xffu));
        uint64_t i2 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * ((sd >> 16) & 0xffu));
        uint64_t i3 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * (sd >> 24));
        for (uint32_t blk = 16; blk-- > 0;) {
          const uint32_t nnblk = blk > 1 ? blk - 2 : 0;
          const uint32_t o3 = ew_step(state, bits_acc, i3);
          const uint64_t n3 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * (nsd >> 24));
          __builtin_amdgcn_sched_barrier(0);
          const uint32_t o2 = ew_step(state, bits_acc, i2);
          const uint64_t n2 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * ((nsd >> 16) & 0xffu));
          const uint32_t o23 = o2 | (o3 << 16);
          __builtin_amdgcn_sched_barrier(0);
          const uint32_t o1 = ew_step(state, bits_acc, i1);
          const uint64_t n1 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * ((nsd >> 8) & 0xffu));
          __builtin_amdgcn_sched_barrier(0);
          const uint32_t o0 = ew_step(state, bits_acc, i0);
          const uint64_t n0 = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * (nsd & 0xffu));
          nsd = *(const uint32_t PCO_LDS*)(uintptr_t)(buf + 16 * nnblk + 4 * j);
          *(u64_align2 PCO_GLOBAL*)(ga + 16 * blk) = (uint64_t)(o0 | (o1 << 16)) | ((uint64_t)o23 << 32);
          __builtin_amdgcn_sched_barrier(0);
          i0 = n0; i1 = n1; i2 = n2; i3 = n3;
        }
      }
      bits_acc += quad_dpp<0xB1>(bits_acc); bits_acc += quad_dpp<0x4E>(bits_acc);
      if (j == 0) gbat[(uint64_t)b * 2 + 1] = bits_acc;
    }
    wd_barrier();
  }
  if (my_n_lat > 0) fx.fstate[((uint64_t)my_p * 3 + my_v) * 4 + j] = state;
}

// =========================================================================================================
// scan: positions of the runs, page size
// =========================================================================================================
__device__ __forceinline__ uint64_t chunk_meta_bits_of(const EncChunk PCO_GLOBAL* ch, uint32_t LB) {  // page_write_chunk_meta, non-fallback
  const uint32_t mode_kind = uni(ch->mode_kind), delta_kind = uni(ch->delta_kind);
  uint64_t bits = kBitsModeVariant;
  if (mode_kind == kIntMult || mode_kind == kFloatMult) bits += LB; else if (mode_kind == kFloatQuant) bits += kBitsQuantK;
  bits += kBitsDeltaVariant;
  if (delta_kind == kDeltaConsecutive) bits += kBitsDeltaOrder + 1;
  else if (delta_kind == kDeltaLookback) bits += kBitsLookbackWindowLog + kBitsLookbackStateLog + 1;
  for (int v = 0; v < 3; v++) {
    if (!uni(ch->v[v].present)) continue;
    const uint32_t lb = v == 0 ? 32u : LB;
    bits += kBitsAnsSizeLog + kBitsNBins + (uint64_t)uni(ch->v[v].n_bins) * (uni(ch->v[v].ans_size_log) + lb + offset_bits_bits(lb));
  }
  return (bits + 7) & ~(uint64_t)7;
}

// grid pages, 64 threads
__global__ __launch_bounds__(64) void enc_scan_kernel(EncWorkspace ws, EncFast fx, PcoGfxTaskResult* results, uint32_t n_pages) {
  const uint32_t p = blockIdx.x;
  if (p >= n_pages) return;
  const uint32_t lane = lane_id();
  EncPage PCO_GLOBAL* pg = (EncPage PCO_GLOBAL*)ws.pages + p;
  const uint32_t t = uni(pg->chunk);
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  if (!page_is_fast(ch, pg)) return;
  const uint32_t LB = (uint32_t)dtype_bits(uni(ch->dtype));
  const uint32_t page_n = (uint32_t)uni((uint64_t)pg->n);
  const uint32_t pflags = uni(pg->flags);
  PageVar pv[3]; uint32_t ob0[3];
  uint64_t head = 0;
  if (pflags & kPageFlagPreamble) head += 8 + kBitsNEntries + chunk_meta_bits_of(ch, LB);
  uint64_t page_meta = 0;
  const uint32_t delta_kind = uni(ch->delta_kind);
#pragma unroll
  for (int v = 0; v < 3; v++) {
    pv[v] = page_var(ch, v, page_n);
    ob0[v] = 0;
    if (!pv[v].present) continue;
    if (pv[v].n_bins == 1) ob0[v] = uni((uint32_t)plan_ref(ws, t, v).bob()[0]);
    if (v == 1) page_meta += (uint64_t)LB * (delta_kind == kDeltaConsecutive ? uni(ch->delta_order) : (delta_kind == kDeltaLookback ? (1u << uni(ch->state_n_log)) : 0u));
    page_meta += 4ull * pv[v].asl;
  }
  head += (page_meta + 7) & ~(uint64_t)7;
  uint32_t PCO_GLOBAL* dst32 = (uint32_t PCO_GLOBAL*)pg->dst;
  const uint64_t cap_bits = uni((uint64_t)pg->dst_cap) * 8;
  const uint32_t n_batches = (page_n + kBatchN - 1) / kBatchN;
  if (fx.body != nullptr) {   // a page enc_walkp_kernel packed: its body is one bit string of known length behind the head (shape 15: no pack kernel but the head's and enc_place_kernel)
    const uint64_t rec = uni(fx.body[2ull * p]);
    if (rec >> 63) {
      const uint64_t end = head + (rec & ~(1ull << 63)), total = (end + 7) & ~(uint64_t)7;
      const bool overflow = total + 64 > cap_bits;
      if (lane == 0) {
        pg->pad = overflow ? 1u : (0x100u | (15u << 4));
        store_result((PcoGfxTaskResult PCO_GLOBAL*)results + p, overflow ? 0 : total >> 3, overflow ? PCO_GFX_INVALID_ARGUMENT : PCO_GFX_OK, 0);
        if (!overflow) { dst32[0] = 0; dst32[total >> 5] = 0; dst32[end >> 5] = 0; dst32[head >> 5] = 0; fx.run_start[(uint64_t)p * fx.run_stride] = head; }
      }
      return;
    }
  }
  uint64_t carry = head;
  // first pass: total size (so that nothing is written to a dst that is too small)
  for (int pass = 0; pass < 2; pass++) {
    carry = head;
    for (uint32_t c0 = 0; c0 < n_batches; c0 += 64) {
      const uint32_t b = c0 + lane;
      uint64_t bits = 0;
      if (b < n_batches) {
        const uint32_t base = b * kBatchN;
#pragma unroll
        for (int v = 0; v < 3; v++) {
          if (!pv[v].present || pv[v].trivial || base >= pv[v].n_lat) continue;
          const uint32_t cnt = pv[v].n_lat - base < kBatchN ? pv[v].n_lat - base : kBatchN;
          if (pv[v].n_bins > 1) { const uint32_t PCO_GLOBAL* r = (const uint32_t PCO_GLOBAL*)fx.bat + (((uint64_t)p * 3 + v) * fx.bat_stride + b) * 2; bits += (uint64_t)r[0] + r[1]; }
          else bits += (uint64_t)cnt * ob0[v];
        }
      }
      const uint64_t incl = wave_incl_scan(bits);
      const uint64_t start = carry + incl - bits;
      if (pass == 1 && b < n_batches && (b % kRunBatches) == 0) {
        fx.run_start[(uint64_t)p * fx.run_stride + b / kRunBatches] = start;
        dst32[start >> 5] = 0;   // the run's first dword is joined with atomicOr by its two writers
      }
      carry += wave_last(incl);
    }
    if (pass == 0) {
      const uint64_t total = (carry + 7) & ~(uint64_t)7;
      const bool overflow = total + 64 > cap_bits;
      // (pad: bit 0 = the page does not fit its dst; else bit 8 and, in bits 4-7, the lean pack kernel's shape for this page -- pack1_mask | full-width << 3,
      //  0 = the general kernel's -- so that the pack kernels of the other shapes leave after one load)
      uint32_t shape = 0;
      if (fx.pack1) { shape = pack1_mask(pv, LB); if (shape) { bool wide = false; for (int v = 0; v < 3; v++) if ((shape >> v) & 1u) wide = wide || !pv[v].compact; shape |= wide ? 8u : 0u; } }
      if (lane == 0) { pg->pad = overflow ? 1u : (0x100u | (shape << 4)); store_result((PcoGfxTaskResult PCO_GLOBAL*)results + p, overflow ? 0 : total >> 3, overflow ? PCO_GFX_INVALID_ARGUMENT : PCO_GFX_OK, 0); }
      if (overflow) return;
      if (lane == 0) { dst32[0] = 0; dst32[total >> 5] = 0; dst32[carry >> 5] = 0; }
    }
  }
}

// =========================================================================================================
// pack
// =========================================================================================================
// Bit sink of one run.  Fields are OR-ed into an LDS staging area (atomics: neighbouring lanes share dwords) that is
// written to dst only when it fills up, so small sections cost no flush.  The run starts at an arbitrary bit of dst:
// the first and the last dword it touches are shared with the neighbouring runs and are merged with atomicOr
// (enc_scan_kernel zeroed them); interior dwords are plain stores.  (bit_writer.rs:22-42 semantics.)
struct PackSink {
  uint32_t PCO_LDS* stg; uint32_t PCO_GLOBAL* dst; uint64_t outbit, first_dw; uint32_t pend;   // pend: staged bits after outbit
  __device__ __forceinline__ void init_at(uint32_t PCO_LDS* s, uint32_t PCO_GLOBAL* d, uint64_t start_bit) {
    stg = s; dst = d; outbit = start_bit; first_dw = start_bit >> 5; pend = 0;
    for (uint32_t i = lane_id(); i < kStgDwords; i += 64) stg[i] = 0;
    enc_wave_sync();
  }
  // nbits <= 64 of val at bit `rel` after the staged bits.  Branch-free: always three ORs (the third is zero for narrow fields).
  __device__ __forceinline__ void put(uint32_t rel, uint64_t val, uint32_t nbits) {
    val &= nbits >= 64 ? ~0ull : (((uint64_t)1 << nbits) - 1);
    const uint32_t pos = (uint32_t)(outbit & 31) + pend + rel;
    const uint32_t dw = pos >> 5, sh = pos & 31;
    const uint64_t lo = val << sh;
#ifdef PCO_PACK_NOATOMIC   // (measurement builds only: what the LDS atomics of the sink cost; the output is garbage)
    stg[dw] = (uint32_t)lo ^ (uint32_t)(lo >> 32) ^ (uint32_t)((val >> 1) >> (63 - sh));
#else
    atomicOr((uint32_t*)&stg[dw], (uint32_t)lo);
    atomicOr((uint32_t*)&stg[dw + 1], (uint32_t)(lo >> 32));
    atomicOr((uint32_t*)&stg[dw + 2], (uint32_t)((val >> 1) >> (63 - sh)));   // val >> (64 - sh), 0 when sh == 0
#endif
  }
  __device__ __forceinline__ void commit(uint32_t total) { pend += total; }
  __device__ __forceinline__ void flush() {
    enc_wave_sync();
    const uint64_t newbit = outbit + pend;
    const uint64_t base_dw = outbit >> 5;
    const uint32_t ncomplete = (uint32_t)((newbit >> 5) - base_dw);
    const uint32_t lane = lane_id();
    for (uint32_t i = lane; i < ncomplete; i += 64) {
      if (base_dw + i == first_dw) atomicOr((uint32_t*)(dst + base_dw + i), stg[i]); else dst[base_dw + i] = stg[i];
    }
    const uint32_t last = stg[ncomplete];
    enc_wave_sync();
    for (uint32_t i = lane; i <= ncomplete; i += 64) stg[i] = 0;
    enc_wave_sync();
    if (lane == 0) stg[0] = last;
    enc_wave_sync();
    outbit = newbit; pend = 0;
  }
  // make room for `bits` more staged bits (plus the 64-bit reach of put)
  __device__ __forceinline__ void reserve(uint32_t bits) { if ((uint32_t)(outbit & 31) + pend + bits + 96 > kStgDwords * 32) flush(); }
  __device__ __forceinline__ void put_uniform(uint64_t val, uint32_t nbits) { reserve(64); if (lane_id() == 0) put(0, val, nbits); commit(nbits); }
  __device__ __forceinline__ void finish_byte() { commit((uint32_t)((8 - ((outbit + pend) & 7)) & 7)); }
  __device__ __forceinline__ void close() {
    flush();
    if (lane_id() == 0 && (outbit & 31)) atomicOr((uint32_t*)(dst + (outbit >> 5)), stg[0]);
  }
};

constexpr uint32_t kPackLdsVar = kPageLdsStg + kStgDwords * 4;   // 2816: per latent slot (lowers u64[256] | offset bits u8[256])
constexpr uint32_t kPackVarBytes = 2048 + 256 + 1024;   // lowers u64[256] | offset bits u8[256] | compact latents: lower | offset bits << 16, u32[256]
__host__ __device__ constexpr uint32_t pack_lds_bytes(uint32_t n_slots) { return kPackLdsVar + n_slots * kPackVarBytes; }

// what one lane holds of one (batch, variable) item: 4 symbols, 4 tANS fields, 4 latents
struct PackItem { uint32_t syms, a, b; uint64_t x[4]; };

// kRaw16: the compact 16-bit latents of a full batch stay packed in x[0] (four per lane: one 8-byte load, cut up where they are used --
// cutting them up here would wait for the load here)
template <class LV, bool kRaw16 = false>
__device__ __forceinline__ void pack_load(PackItem& it, const LV PCO_GLOBAL* lat, const uint8_t PCO_GLOBAL* sym, const uint16_t PCO_GLOBAL* answ,
                                          uint32_t cnt, bool needs_ans, bool single_bin, bool has_offsets) {
  const uint32_t lane = lane_id();
  typedef uint64_t __attribute__((aligned(2))) u64_align2;
  if (cnt == kBatchN) {
    // The common case: every load unconditional, into registers nothing else writes.  (Zeroing the item first and loading under the
    // variable's flags made the compiler wait -- at the zeroing and at the branches -- for every load in flight: the "prefetch" of the next
    // batch was waited for on the spot, and the kernel sat at 2.7 TB/s of its own traffic with four waves per SIMD to hide a batch's
    // round trip to HBM.  What a variable without symbols / tANS fields / offsets reads here is never looked at.)
    it.syms = *(const u32_unaligned PCO_GLOBAL*)(sym + 4 * lane);
    { const uint64_t w = *(const u64_align2 PCO_GLOBAL*)(answ + 4 * lane); it.a = (uint32_t)w; it.b = (uint32_t)(w >> 32); }
    if constexpr (sizeof(LV) == 2) {
      const uint64_t w = *(const u64_align2 PCO_GLOBAL*)(lat + 4 * lane);
      if constexpr (kRaw16) it.x[0] = w;
      else { it.x[0] = w & 0xffffu; it.x[1] = (w >> 16) & 0xffffu; it.x[2] = (w >> 32) & 0xffffu; it.x[3] = w >> 48; }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) it.x[k] = (uint64_t)lat[4 * lane + k];
    }
    return;
  }
  it.syms = 0; it.a = it.b = 0; it.x[0] = it.x[1] = it.x[2] = it.x[3] = 0;
  const bool blk_on = 4 * lane < ((cnt + 15u) & ~15u);
  if (!single_bin && blk_on) it.syms = *(const u32_unaligned PCO_GLOBAL*)(sym + 4 * lane);
  if (needs_ans && blk_on) { const uint64_t w = *(const u64_align2 PCO_GLOBAL*)(answ + 4 * lane); it.a = (uint32_t)w; it.b = (uint32_t)(w >> 32); }
  if (has_offsets) {
#pragma unroll
    for (int k = 0; k < 4; k++) it.x[k] = 4 * lane + k < cnt ? (uint64_t)lat[4 * lane + k] : 0ull;
    if constexpr (kRaw16) it.x[0] = it.x[0] | (it.x[1] << 16) | (it.x[2] << 32) | (it.x[3] << 48);
  }
}

// pack one item: its tANS fields, then its offset fields (chunk_latent_compressor.rs:134-169)
template <bool kFull>
__device__ __forceinline__ void pack_item_t(PackSink& sink, const uint8_t PCO_LDS* vt, const PackItem& it, uint32_t cnt, uint32_t asl, bool needs_ans, uint32_t max_ob, bool single_bin, bool compact) {
  const uint32_t lane = lane_id();
  const uint64_t PCO_LDS* low = (const uint64_t PCO_LDS*)vt;
  const uint8_t PCO_LDS* obs = vt + 2048;
  sink.reserve((needs_ans ? cnt * asl : 0u) + cnt * max_ob);
  const uint32_t syms = single_bin ? 0u : quad_transpose_u8(it.syms, lane & 3);
  if (needs_ans) {
    uint32_t a = it.a, b = it.b;
    quad_transpose_u16(a, b, lane & 3);
    const uint32_t f[4] = {a & 0xffffu, a >> 16, b & 0xffffu, b >> 16};
    uint64_t acc = 0; uint32_t accbits = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool act = kFull || 4 * lane + k < cnt;
      const uint32_t nb = act ? f[k] >> 12 : 0u;
      acc |= (uint64_t)(act ? f[k] & 0xfffu : 0u) << accbits; accbits += nb;
    }
    const uint32_t incl = wave_incl_scan(accbits);
    sink.put(incl - accbits, acc, accbits);  // <= 48 bits
    sink.commit(wave_last(incl));
  }
  if (max_ob != 0 && compact) {   // 16-bit latents relative to the minimum: one packed table word per symbol, 32-bit arithmetic, one put
    const uint32_t PCO_LDS* cpk = (const uint32_t PCO_LDS*)(vt + 2304);
    uint64_t acc = 0; uint32_t t = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t e = cpk[(syms >> (8 * k)) & 0xffu];
      const uint32_t o = (kFull || 4 * lane + k < cnt) ? e >> 16 : 0u;
      acc |= (uint64_t)__builtin_amdgcn_ubfe(((uint32_t)(it.x[0] >> (16 * k)) & 0xffffu) - (e & 0xffffu), 0u, o) << t;   // (compact latents arrive packed, four in x[0])
      t += o;
    }
    const uint32_t incl = wave_incl_scan(t);
    sink.put(incl - t, acc, t);
    sink.commit(wave_last(incl));
  } else if (max_ob != 0) {
    uint32_t ob[4]; uint64_t x[4]; uint32_t t = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t s = (syms >> (8 * k)) & 0xffu;
      const uint32_t o = obs[s];
      ob[k] = (kFull || 4 * lane + k < cnt) ? o : 0u;
      x[k] = it.x[k] - low[s];
      t += ob[k];
    }
    const uint32_t incl = wave_incl_scan(t);
    uint32_t rel = incl - t;
    // the widest offset of THIS batch decides how many puts it takes (a variable with a few far outliers has a large max_ob and narrow
    // offsets in nearly every batch)
    const uint32_t o01 = ob[0] > ob[1] ? ob[0] : ob[1], o23 = ob[2] > ob[3] ? ob[2] : ob[3];
    const uint32_t bmax = max_ob <= 16 ? max_ob : wave_max_u32(o01 > o23 ? o01 : o23);
    if (bmax <= 16) {  // the lane's four fields fit one 64-bit word: one put
      uint64_t acc = 0; uint32_t sh = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) { acc |= (uint64_t)__builtin_amdgcn_ubfe((uint32_t)x[k], 0u, ob[k]) << sh; sh += ob[k]; }
      sink.put(rel, acc, t);
    } else if (bmax <= 32) {  // two fields per 64-bit word: two puts
#pragma unroll
      for (int k = 0; k < 4; k += 2) {
        const uint64_t lo = x[k] & (((uint64_t)1 << ob[k]) - 1), hi = x[k + 1] & (((uint64_t)1 << ob[k + 1]) - 1);
        sink.put(rel, lo | (hi << ob[k]), ob[k] + ob[k + 1]);
        rel += ob[k] + ob[k + 1];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) { sink.put(rel, x[k], ob[k]); rel += ob[k]; }
    }
    sink.commit(wave_last(incl));
  }
}
__device__ __forceinline__ void pack_item(PackSink& sink, const uint8_t PCO_LDS* vt, const PackItem& it, uint32_t cnt, uint32_t asl, bool needs_ans, uint32_t max_ob, bool single_bin, bool compact) {
  if (cnt == kBatchN) pack_item_t<true>(sink, vt, it, cnt, asl, needs_ans, max_ob, single_bin, compact);
  else pack_item_t<false>(sink, vt, it, cnt, asl, needs_ans, max_ob, single_bin, compact);
}

template <class L>
__device__ __forceinline__ void pack_run(const EncWorkspace& ws, const EncFast& fx, uint32_t p, uint32_t run, EncPage PCO_GLOBAL* pg, EncChunk PCO_GLOBAL* ch, bool lean) {
  const uint32_t t = uni(pg->chunk);
  const uint32_t lane = lane_id();
  constexpr uint32_t LB = LBits<L>::v;
  const uint32_t page_n = (uint32_t)uni((uint64_t)pg->n);
  const uint64_t pstart = uni((uint64_t)pg->start);
  if ((uint64_t)run * kRunBatches * kBatchN >= page_n) return;
  uint8_t PCO_LDS* smem = enc_lds_base();
  PageVar pv[3];
#pragma unroll
  for (int v = 0; v < 3; v++) pv[v] = page_var(ch, v, page_n);
  // lean: enc_pack1_kernel packs this page's batches; what is left here is the head of run 0 (preamble, ChunkMeta, page meta)
  PackSink sink;
  sink.init_at((uint32_t PCO_LDS*)(smem + kPageLdsStg), (uint32_t PCO_GLOBAL*)pg->dst, run == 0 ? 0ull : uni(fx.run_start[(uint64_t)p * fx.run_stride + run]));
  if (run == 0) {
    const uint32_t pflags = uni(pg->flags);
    if (pflags & kPageFlagPreamble) {  // standalone/compressor.rs:191-203, then ChunkMeta
      sink.put_uniform(uni(ch->dtype), 8);
      sink.put_uniform(page_n - 1, kBitsNEntries);
      // ChunkMeta (metadata/chunk.rs:176-189, mode.rs:169-195, delta_encoding.rs:204-254, chunk_latent_var.rs:55-71,158-168)
      const uint32_t mode_kind = uni(ch->mode_kind), delta_kind = uni(ch->delta_kind), delta_order = uni(ch->delta_order);
      sink.put_uniform(mode_kind, kBitsModeVariant);
      if (mode_kind == kIntMult || mode_kind == kFloatMult) sink.put_uniform(uni((uint64_t)ch->mode_base), LB);
      else if (mode_kind == kFloatQuant) sink.put_uniform(uni(ch->mode_k), kBitsQuantK);
      sink.put_uniform(delta_kind, kBitsDeltaVariant);
      if (delta_kind == kDeltaConsecutive) { sink.put_uniform(delta_order, kBitsDeltaOrder); sink.put_uniform(0, 1); }
      else if (delta_kind == kDeltaLookback) { sink.put_uniform(uni(ch->window_n_log) - 1, kBitsLookbackWindowLog); sink.put_uniform(uni(ch->state_n_log), kBitsLookbackStateLog); sink.put_uniform(0, 1); }
#pragma unroll
      for (int v = 0; v < 3; v++) {
        if (!pv[v].present) continue;
        const PlanRef plan = plan_ref(ws, t, v);
        const uint32_t asl = pv[v].asl, nbins = pv[v].n_bins;
        const uint32_t lb = v == 0 ? 32u : LB, obb = offset_bits_bits(lb);
        sink.put_uniform(asl, kBitsAnsSizeLog); sink.put_uniform(nbins, kBitsNBins);
        const uint32_t bin_bits = asl + lb + obb;
        for (uint32_t b0 = 0; b0 < nbins; b0 += 64) {
          const uint32_t b = b0 + lane;
          const uint32_t nb = nbins - b0 < 64 ? nbins - b0 : 64;
          sink.reserve(64 * bin_bits);
          if (b < nbins) {
            const uint32_t rel = lane * bin_bits;
            sink.put(rel, plan.bweight()[b] - 1, asl);
            sink.put(rel + asl, plan.blower()[b], lb);
            sink.put(rel + asl + lb, plan.bob()[b], obb);
          }
          sink.commit(nb * bin_bits);
        }
      }
      sink.finish_byte();
    }
    // page meta (metadata/page.rs:22-34, page_latent_var.rs:19-26)
    const uint32_t delta_kind = uni(ch->delta_kind);
#pragma unroll
    for (int v = 0; v < 3; v++) {
      if (!pv[v].present) continue;
      if (v == 1) {
        const uint32_t nlps = delta_kind == kDeltaConsecutive ? uni(ch->delta_order) : (delta_kind == kDeltaLookback ? (1u << uni(ch->state_n_log)) : 0u);
        for (uint32_t i = 0; i < nlps; i++) sink.put_uniform(uni((uint64_t)pg->moments[i]), LB);
      }
      for (int jj = 0; jj < 4; jj++) sink.put_uniform(uni(fx.fstate[((uint64_t)p * 3 + v) * 4 + jj]) - (1u << pv[v].asl), pv[v].asl);
    }
    sink.finish_byte();
  }
  if (lean) { sink.close(); return; }   // (the head ends on a byte, at run_start[0]: enc_scan_kernel zeroed that dword, both writers OR into it)
  // tables for the offsets: lowers widened to u64 (differences wrap the same way once masked to offset_bits)
  bool on[3];
#pragma unroll
  for (int v = 0; v < 3; v++) {
    on[v] = pv[v].present && !pv[v].trivial;
    if (!on[v]) continue;
    const PlanRef plan = plan_ref(ws, t, v);
    uint8_t PCO_LDS* vt = smem + kPackLdsVar + ws.slot_of_var[v] * kPackVarBytes;
    const uint64_t rel0 = pv[v].compact ? pv[v].rel : 0ull;   // compact latents are relative to the minimum
    for (uint32_t b = lane; b < pv[v].n_bins; b += 64) {
      const uint64_t lw = plan.blower()[b] - rel0; const uint32_t ob = plan.bob()[b];
      ((uint64_t PCO_LDS*)vt)[b] = lw; (vt + 2048)[b] = (uint8_t)ob;
      if (pv[v].compact) ((uint32_t PCO_LDS*)(vt + 2304))[b] = ((uint32_t)lw & 0xffffu) | (ob << 16);
    }
  }
  enc_wave_sync();
  // One bin whose offsets are the latent's full width, nothing else to write (uniform random numbers -- BASELINE configs[0] --, the deltas of
  // anything without structure): the page's body is the latents minus the bin's lower bound, byte for byte (chunk_latent_compressor.rs:272-329:
  // no tANS fields, a batch is 256 whole offsets and ends on a byte).  A shifted copy: 16 bytes per lane and step from the full-width
  // latents to wherever the run starts -- through the bit sink it ran at 2.6 TB/s of its own traffic (13.3 ms per 8192 chunks of configs[0]).
  if (on[1] && !on[0] && !on[2] && pv[1].n_bins == 1 && !pv[1].needs_ans && pv[1].max_ob == LB && !pv[1].compact) {
    const uint64_t body_bit = sink.outbit + sink.pend;   // (a whole byte: behind finish_byte, or a run start behind whole batches)
    sink.close();
    constexpr uint32_t kPer = 16 / sizeof(L);            // latents per lane and step
    typedef L lvec __attribute__((ext_vector_type(kPer)));
    typedef lvec __attribute__((aligned(1))) lvec_unaligned;
    const L low0 = (L)plan_ref(ws, t, 1).blower()[0];
    const uint32_t first = run * kRunBatches * kBatchN;
    const uint32_t n_run = pv[1].n_lat - first < kRunBatches * kBatchN ? pv[1].n_lat - first : kRunBatches * kBatchN;
    const L PCO_GLOBAL* src = lat_ptr<L>(ws, t, 1) + pstart + pv[1].skip + first;
    uint8_t PCO_GLOBAL* out8 = (uint8_t PCO_GLOBAL*)pg->dst + (body_bit >> 3);
    // (four steps' loads in flight before the first store: a step at a time the wave waited out a memory round trip per kilobyte)
    uint32_t i = lane * kPer;
    for (; i + 3 * 64 * kPer + kPer <= n_run; i += 4 * 64 * kPer) {
      lvec v[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) v[u] = *(const lvec_unaligned PCO_GLOBAL*)(src + i + u * 64 * kPer);   // (one 16-byte request per lane, whatever the page's first position)
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) {
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) v[u][k] = (L)(v[u][k] - low0);
        *(lvec_unaligned PCO_GLOBAL*)(out8 + (uint64_t)(i + u * 64 * kPer) * sizeof(L)) = v[u];
      }
    }
    for (; i < n_run; i += 64 * kPer) {
      if (i + kPer <= n_run) {
        lvec v;
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) v[k] = (L)(src[i + k] - low0);
        *(lvec_unaligned PCO_GLOBAL*)(out8 + (uint64_t)i * sizeof(L)) = v;
      } else {
        typedef L __attribute__((aligned(1))) l_unaligned;
        for (uint32_t k = 0; i + k < n_run; k++) *(l_unaligned PCO_GLOBAL*)(out8 + (uint64_t)(i + k) * sizeof(L)) = (L)(src[i + k] - low0);
      }
    }
    return;
  }
  // batches of the run; the next batch's loads are issued before the current one is packed
  PackItem cur[3], nxt[3];
  auto load_batch = [&](uint32_t bb, PackItem (&dstv)[3]) {
    const uint32_t base = (run * kRunBatches + bb) * kBatchN;
#pragma unroll
    for (int v = 0; v < 3; v++) {
      if (!on[v] || base >= pv[v].n_lat) continue;
      const uint32_t cnt = pv[v].n_lat - base < kBatchN ? pv[v].n_lat - base : kBatchN;
      const uint64_t at = pstart + pv[v].skip + base, fat = fast_at(pg, pv[v].skip) + base;
      if (pv[v].compact) pack_load<uint16_t, true>(dstv[v], clat_ptr(ws, t, v) + at, fsym_ptr(ws, fx, t, v) + fat, fansw_ptr(ws, fx, t, v) + fat, cnt, pv[v].needs_ans != 0, pv[v].n_bins <= 1, pv[v].max_ob != 0);
      else if (v == 0) pack_load<uint32_t>(dstv[v], lat_ptr<uint32_t>(ws, t, 0) + at, fsym_ptr(ws, fx, t, 0) + fat, fansw_ptr(ws, fx, t, 0) + fat, cnt, pv[v].needs_ans != 0, pv[v].n_bins <= 1, pv[v].max_ob != 0);
      else pack_load<L>(dstv[v], lat_ptr<L>(ws, t, v) + at, fsym_ptr(ws, fx, t, v) + fat, fansw_ptr(ws, fx, t, v) + fat, cnt, pv[v].needs_ans != 0, pv[v].n_bins <= 1, pv[v].max_ob != 0);
    }
  };
  load_batch(0, cur);
  for (uint32_t bb = 0; bb < kRunBatches; bb++) {
    const uint32_t base = (run * kRunBatches + bb) * kBatchN;
    if (base >= page_n) break;
    if (bb + 1 < kRunBatches && base + kBatchN < page_n) load_batch(bb + 1, nxt);
#pragma unroll
    for (int v = 0; v < 3; v++) {
      if (!on[v] || base >= pv[v].n_lat) continue;
      const uint32_t cnt = pv[v].n_lat - base < kBatchN ? pv[v].n_lat - base : kBatchN;
      pack_item(sink, smem + kPackLdsVar + ws.slot_of_var[v] * kPackVarBytes, cur[v], cnt, pv[v].asl, pv[v].needs_ans != 0, pv[v].max_ob, pv[v].n_bins <= 1, pv[v].compact != 0);
    }
#pragma unroll
    for (int v = 0; v < 3; v++) cur[v] = nxt[v];
  }
  if ((uint64_t)(run + 1) * kRunBatches * kBatchN >= page_n) sink.finish_byte();
  sink.close();
}

// ---------------------------------------------------------------------------------------------------------
// enc_pack1_kernel: the batches of the pages with one or two latent variables (pack1_mask), nothing else.  The general kernel above carries
// three variables' worth of per-batch state (two PackItems of eleven registers each per variable, ~124 VGPRs: four waves per SIMD) and decides
// per batch and variable what to do; it ran at 2.7 TB/s of its own traffic, about half VALU-bound, two thirds of its wave cycles waiting
// (profiles/r04_c2_pmc_instruction_mix.txt).  Here the kernel is instantiated per shape -- which variables, 16-bit latents only or also
// full-width ones -- a variable's batch is ONE wave scan for both of its sections (the lane's tANS bits in the low half of a dword, its
// offset bits in the high half: a batch holds at most 256 x 15 of the one and 256 x 64 of the other) and one to five puts; the head of the page (preamble,
// ChunkMeta, page meta) stays with the general kernel, which stops behind it for these pages.  Same bytes
// (chunk_latent_compressor.rs:272-329), same run structure and joins (enc_scan_kernel).  BASELINE configs[1]: 5.03 -> 2.92 + 0.28 ms.
// ---------------------------------------------------------------------------------------------------------
// LDS: the bit sink's staging, then per variable on -- kWide: lowers u64[256] | offset bits u8[256] | lower | offset bits << 16, u32[256]; else the last table alone
__host__ __device__ constexpr uint32_t pack1_var_bytes(bool wide) { return wide ? 2048u + 256u + 1024u : 1024u; }
__host__ __device__ constexpr uint32_t pack1_cpk_off(bool wide) { return wide ? 2304u : 0u; }
__host__ __device__ constexpr uint32_t pack1_lds_bytes(uint32_t mask, bool wide) { return kStgDwords * 4 + (mask == 2u ? 1u : 2u) * pack1_var_bytes(wide); }

// one variable's batch: its tANS fields, then its offset fields (chunk_latent_compressor.rs:134-169)
template <bool kFull, bool kWide>
__device__ __forceinline__ void pack1_item(PackSink& sink, const uint8_t PCO_LDS* vt, const PackItem& it, uint32_t cnt, bool needs_ans, bool single_bin, uint32_t max_ob, bool compact) {
  const uint32_t lane = lane_id();
  const uint32_t syms = single_bin ? 0u : quad_transpose_u8(it.syms, lane & 3);
  uint64_t acc = 0; uint32_t abits = 0;
  if (needs_ans) {
    uint32_t a = it.a, b = it.b;
    quad_transpose_u16(a, b, lane & 3);
    const uint32_t f[4] = {a & 0xffffu, a >> 16, b & 0xffffu, b >> 16};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool act = kFull || 4 * lane + k < cnt;
      const uint32_t nb = act ? f[k] >> 12 : 0u;
      acc |= (uint64_t)(act ? f[k] & 0xfffu : 0u) << abits; abits += nb;      // <= 4 x 15 bits
    }
  }
  uint32_t ob[4] = {0, 0, 0, 0}, obits = 0; uint64_t x[4] = {0, 0, 0, 0};
  if (max_ob != 0) {
    if (!kWide || compact) {   // 16-bit latents relative to rel: one packed table word per symbol, 32-bit arithmetic
      const uint32_t PCO_LDS* cpk = (const uint32_t PCO_LDS*)(vt + pack1_cpk_off(kWide));
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t e = cpk[(syms >> (8 * k)) & 0xffu];
        ob[k] = (kFull || 4 * lane + k < cnt) ? e >> 16 : 0u;
        x[k] = (uint64_t)__builtin_amdgcn_ubfe(((uint32_t)(it.x[0] >> (16 * k)) & 0xffffu) - (e & 0xffffu), 0u, ob[k]);   // (compact latents arrive packed, four in x[0])
        obits += ob[k];
      }
    } else {
      const uint64_t PCO_LDS* low = (const uint64_t PCO_LDS*)vt;
      const uint8_t PCO_LDS* obs = vt + 2048;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t sy = (syms >> (8 * k)) & 0xffu;
        ob[k] = (kFull || 4 * lane + k < cnt) ? (uint32_t)obs[sy] : 0u;
        const uint64_t d = it.x[k] - low[sy];
        x[k] = ob[k] >= 64 ? d : d & (((uint64_t)1 << ob[k]) - 1);
        obits += ob[k];
      }
    }
  }
  const uint32_t both = abits | (obits << 16);                                  // (obits <= 256 per lane, 16384 per batch)
  const uint32_t incl = wave_incl_scan(both), total = wave_last(incl), excl = incl - both;
  const uint32_t ans_total = total & 0xffffu, all = ans_total + (total >> 16);
  sink.reserve(all);
  if (needs_ans) sink.put(excl & 0xffffu, acc, abits);
  if (max_ob != 0) {
    uint32_t rel = ans_total + (excl >> 16);
    // the widest offset of THIS batch decides how many puts it takes (as pack_item_t)
    uint32_t bmax = max_ob;
    if (kWide && max_ob > 16) { const uint32_t o01 = ob[0] > ob[1] ? ob[0] : ob[1], o23 = ob[2] > ob[3] ? ob[2] : ob[3]; bmax = wave_max_u32(o01 > o23 ? o01 : o23); }
    if (!kWide || bmax <= 16) {   // the lane's four fields fit one 64-bit word
      uint64_t w = 0; uint32_t sh = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) { w |= x[k] << sh; sh += ob[k]; }
      sink.put(rel, w, obits);
    } else if (bmax <= 32) {
#pragma unroll
      for (int k = 0; k < 4; k += 2) { sink.put(rel, x[k] | (x[k + 1] << ob[k]), ob[k] + ob[k + 1]); rel += ob[k] + ob[k + 1]; }
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) { sink.put(rel, x[k], ob[k]); rel += ob[k]; }
    }
  }
  sink.commit(all);
}

// kMask: the variables on (2: primary; 6: primary + secondary; 3: lookback variable + primary).  kWide: a variable may hold full-width latents
// (L = uint32_t / uint64_t; the lookback variable's are uint32_t) -- otherwise every variable on has 16-bit latents and L plays no part.
template <class L, uint32_t kMask, bool kWide>
__device__ __forceinline__ void pack1_run(const EncWorkspace& ws, const EncFast& fx, uint32_t p, uint32_t run, EncPage PCO_GLOBAL* pg, uint32_t t, const PageVar (&pv)[3], uint32_t page_n) {
  const uint32_t lane = lane_id();
  uint8_t PCO_LDS* smem = enc_lds_base();
  constexpr int kV0 = (kMask & 1u) ? 0 : 1, kV1 = (kMask & 4u) ? 2 : 1;   // the one or two variables, in stream order (kV0 == kV1: one)
  constexpr int kN = kV0 == kV1 ? 1 : 2;
  const int vs[2] = {kV0, kV1};
#pragma unroll
  for (int i = 0; i < kN; i++) {
    const int v = vs[i];
    const PlanRef plan = plan_ref(ws, t, v);
    uint8_t PCO_LDS* vt = smem + kStgDwords * 4 + i * pack1_var_bytes(kWide);
    const uint64_t rel0 = pv[v].compact ? pv[v].rel : 0ull;   // 16-bit latents are relative to it
    for (uint32_t b = lane; b < pv[v].n_bins; b += 64) {
      const uint64_t lw = plan.blower()[b] - rel0; const uint32_t ob = plan.bob()[b];
      if (kWide) { ((uint64_t PCO_LDS*)vt)[b] = lw; (vt + 2048)[b] = (uint8_t)ob; }
      ((uint32_t PCO_LDS*)(vt + pack1_cpk_off(kWide)))[b] = ((uint32_t)lw & 0xffffu) | (ob << 16);
    }
  }
  PackSink sink;
  sink.init_at((uint32_t PCO_LDS*)smem, (uint32_t PCO_GLOBAL*)pg->dst, uni(fx.run_start[(uint64_t)p * fx.run_stride + run]));   // (run 0 starts behind the head; init_at ends with a wave sync: the tables are visible)
  const uint32_t first = run * kRunBatches * kBatchN;
  const uint64_t pstart = uni((uint64_t)pg->start);
  PackItem cur[kN], nxt[kN];
  auto load = [&](uint32_t bb, PackItem (&dstv)[kN]) {
    const uint32_t base = first + bb * kBatchN;
#pragma unroll
    for (int i = 0; i < kN; i++) {
      const int v = vs[i];
      if (base >= pv[v].n_lat) continue;
      const uint32_t cnt = pv[v].n_lat - base < kBatchN ? pv[v].n_lat - base : kBatchN;
      const uint64_t at = pstart + pv[v].skip + base, fat = fast_at(pg, pv[v].skip) + base;
      const uint8_t PCO_GLOBAL* sy = fsym_ptr(ws, fx, t, v) + fat; const uint16_t PCO_GLOBAL* an = fansw_ptr(ws, fx, t, v) + fat;
      if (!kWide || pv[v].compact) pack_load<uint16_t, true>(dstv[i], clat_ptr(ws, t, v) + at, sy, an, cnt, pv[v].needs_ans != 0, pv[v].n_bins <= 1, pv[v].max_ob != 0);
      else if (v == 0) pack_load<uint32_t>(dstv[i], lat_ptr<uint32_t>(ws, t, 0) + at, sy, an, cnt, pv[v].needs_ans != 0, pv[v].n_bins <= 1, pv[v].max_ob != 0);
      else pack_load<L>(dstv[i], lat_ptr<L>(ws, t, v) + at, sy, an, cnt, pv[v].needs_ans != 0, pv[v].n_bins <= 1, pv[v].max_ob != 0);
    }
  };
  load(0, cur);
  for (uint32_t bb = 0; bb < kRunBatches; bb++) {
    const uint32_t base = first + bb * kBatchN;
    if (base >= page_n) break;
    if (bb + 1 < kRunBatches && base + kBatchN < page_n) load(bb + 1, nxt);
#pragma unroll
    for (int i = 0; i < kN; i++) {
      const int v = vs[i];
      if (base >= pv[v].n_lat) continue;
      const uint32_t cnt = pv[v].n_lat - base < kBatchN ? pv[v].n_lat - base : kBatchN;
      const uint8_t PCO_LDS* vt = smem + kStgDwords * 4 + i * pack1_var_bytes(kWide);
      if (cnt == kBatchN) pack1_item<true, kWide>(sink, vt, cur[i], cnt, pv[v].needs_ans != 0, pv[v].n_bins <= 1, pv[v].max_ob, pv[v].compact != 0);
      else pack1_item<false, kWide>(sink, vt, cur[i], cnt, pv[v].needs_ans != 0, pv[v].n_bins <= 1, pv[v].max_ob, pv[v].compact != 0);
    }
#pragma unroll
    for (int i = 0; i < kN; i++) cur[i] = nxt[i];
  }
  if ((uint64_t)(run + 1) * kRunBatches * kBatchN >= page_n) sink.finish_byte();
  sink.close();
}

// grid pages * runs_per_page, 64 threads; one instantiation per shape (launched only for the shapes a call can have: `shape` = kMask | kWide << 3 | 64-bit << 4)
template <class L, uint32_t kMask, bool kWide>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(kWide ? 5 : 8, kWide ? 5 : 8))) void enc_pack1_kernel(EncWorkspace ws, EncFast fx, uint32_t n_pages) {
  const uint32_t p = blockIdx.x / fx.runs_per_page, run = blockIdx.x % fx.runs_per_page;
  if (p >= n_pages) return;
  EncPage PCO_GLOBAL* pg = (EncPage PCO_GLOBAL*)ws.pages + p;
  const uint32_t t = uni(pg->chunk);
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  const uint32_t pad = uni(pg->pad);   // enc_scan_kernel: bit 0 = does not fit, bits 4-7 = this page's shape
  if ((pad & 1u) != 0 || ((pad >> 4) & 0xfu) != (kMask | (kWide ? 8u : 0u))) return;
  const uint32_t page_n = (uint32_t)uni((uint64_t)pg->n);
  if ((uint64_t)run * kRunBatches * kBatchN >= page_n) return;
  if (kWide && (uint32_t)dtype_bits(uni(ch->dtype)) != LBits<L>::v) return;
  PageVar pv[3];
#pragma unroll
  for (int v = 0; v < 3; v++) pv[v] = page_var(ch, v, page_n);
  pack1_run<L, kMask, kWide>(ws, fx, p, run, pg, t, pv, page_n);
}

// grid pages * runs_per_page, 64 threads
// (left at the ~124 VGPRs it wants: at 80 / 64 -- six / eight waves per SIMD -- it spills and takes 9.1 / 21 ms instead of 5.0)
__global__ __launch_bounds__(64) void enc_pack_kernel(EncWorkspace ws, EncFast fx, uint32_t n_pages) {
  const uint32_t p = blockIdx.x / fx.runs_per_page, run = blockIdx.x % fx.runs_per_page;
  if (p >= n_pages) return;
  EncPage PCO_GLOBAL* pg = (EncPage PCO_GLOBAL*)ws.pages + p;
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + uni(pg->chunk);
  const uint32_t pad = uni(pg->pad);   // enc_scan_kernel: bit 0 = does not fit, bits 4-7 = the lean kernel's shape (0: this kernel packs the page)
  if (!page_is_fast(ch, pg) || (pad & 1u) != 0) return;
  const bool lean = ((pad >> 4) & 0xfu) != 0;
  if (lean && run != 0) return;   // (what is left here of a lean page is the head of run 0: preamble, ChunkMeta, page meta)
  const int bits = dtype_bits(uni(ch->dtype));
  if (bits == 64) pack_run<uint64_t>(ws, fx, p, run, pg, ch, lean);
  else if (bits == 32) pack_run<uint32_t>(ws, fx, p, run, pg, ch, lean);
  else if (bits == 16) pack_run<uint16_t>(ws, fx, p, run, pg, ch, lean);
  else pack_run<uint8_t>(ws, fx, p, run, pg, ch, lean);
}

}  // namespace pcogfx
