// encode_walkpack.hip -- the reverse tANS walk AND the bit-packing of a page in one block (round 6).
//
// enc_walkd_kernel's walker is a chain of n / 4 dependent LDS round trips per page (3.4 ms per 2^18 latents whatever the call holds); its
// gathering waves find the symbols beside it and the SIMDs are otherwise idle.  Until round 6 the page was then packed by a second kernel
// (enc_pack1_kernel, 3.2 ms per 8192 chunks of BASELINE configs[1]) that read the 16-bit latents a second time plus what the walk had left in
// HBM for it -- a symbol byte and a 16-bit tANS field per latent -- 5 bytes per latent to write 1.2.  Here the gathering waves of a block pack
// every batch themselves, two barriers behind the walker: the walker leaves a batch's tANS fields in LDS, the gathering wave that found the
// batch's symbols still holds its latents and bins in registers, looks the bins' lower bounds and offset-bit counts up in an LDS table, and
// writes the batch's bits -- tANS fields, then offsets: chunk_latent_compressor.rs:134-169 -- through a small LDS bit stage.
//
// Where?  The walk runs from the page's last batch to its first (ans/encoding.rs: the decoder reads forwards), so when a batch is packed the
// bits of the batches BEFORE it are not known.  The batches are therefore written downwards from the end of the page's tANS-field scratch
// (which these pages no longer need): batch b ends where batch b + 1 begins, bit-exactly, and when the walk is over the page's body is one
// contiguous, right-aligned bit string of known length.  enc_scan_kernel sizes the page from it, the general pack kernel writes the head
// (preamble, ChunkMeta, page meta with the final tANS states), and enc_place_kernel moves the body behind the head with one funnel shift per
// dword -- 1.2 bytes per latent read and written instead of 5 + 1.2.
//
// Which pages: every walked item of the block is the ONLY variable of its page that writes anything (classic mode without lookback, or
// a two-variable mode whose other variable is trivial), has 16-bit latents spanning fewer than 4096 values (the value -> bin tables of
// enc_vlut_kernel), tables that fit the slot below, and bins whose tANS bits + offset bits never exceed 16 per latent (so that the body fits
// the 2 bytes per latent of the field scratch).  A block with any other item leaves everything to enc_walkd_kernel and the pack kernels.
// Same bytes either way (tests: every encode test runs through here when its pages qualify; PCO_GFX_WALKP=0 switches it off).
#pragma once

namespace pcogfx {

// Eight packing waves of two items each, at most 80 registers (measured, 8192 chunks of BASELINE configs[1]: four waves of four items 5.4 ms -- a
// packing wave's own chain of LDS round trips, scans and flushes takes 8500 cycles per batch step where the walker takes 7100; eight waves at 96
// registers 6.6 ms -- a block's nine waves land on the SIMDs as 3-2-2-2, two blocks may put six on one: the second block did not fit; eight at 80: 4.4 ms)
#ifndef PCO_WP_HELPERS
#define PCO_WP_HELPERS 8
#endif
#ifndef PCO_WP_WAVES
#define PCO_WP_WAVES 6
#endif
constexpr uint32_t kWpQ = 16, kWpHelpers = PCO_WP_HELPERS, kWpH = kWpQ / kWpHelpers;   // items per block, gathering / packing waves, items per such wave
constexpr uint32_t kWpSlot = 5120;                                         // LDS per item: two blocks per CU take all 160 KB
constexpr uint32_t kWpStgDwords = 132;                                     // bit stage: one batch (<= 4096 bits) + the carried dword + the reach of a 64-bit put
// Slot q starts 8 q bytes into its 5120: the slots' strides are a multiple of the 128 bytes the LDS banks repeat after, and sixteen quads (or four
// waves) that read the same offset of sixteen slots -- the same step's symbols, the info word of the most frequent bin -- would all hit one bank.
#ifndef PCO_WP_SKEW
#define PCO_WP_SKEW 8
#endif
constexpr uint32_t kWpSkew = PCO_WP_SKEW, kWpSlotUse = kWpSlot - kWpSkew * (16 - 1);   // 5000 usable bytes
__device__ __forceinline__ uint32_t wp_slot(uint32_t q) { return q * (kWpSlot + kWpSkew); }
constexpr uint32_t kWpStgOff = (kWpSlotUse - kWpStgDwords * 4) & ~15u;     // 4464
constexpr uint32_t kWpAnsOff = kWpStgOff - 1024;                           // tANS fields u16[2][256], quad-transposed (as the walker produces them)
constexpr uint32_t kWpSymOff = kWpAnsOff - 512;                            // symbols u8[2][256], quad-transposed
constexpr uint32_t kWpLdsBytes = kWpQ * kWpSlot;                           // 81920
static_assert(kWpStgOff % 16 == 0 && kWpAnsOff % 16 == 0 && kWpSymOff % 16 == 0 && kWpStgOff + kWpStgDwords * 4 <= kWpSlotUse, "aligned sections");
static_assert(2 * kWpLdsBytes <= 160 * 1024, "two blocks per CU");
constexpr uint64_t kWpBodyFlag = 1ull << 63;
// slot: next states u16[T] | info u64[n_bins] (ew_step) | lower (16 bits, relative) | offset bits << 16, u32[n_bins] | ... | symbols | fields | stage
__device__ __forceinline__ bool wp_fits(uint32_t asl, uint32_t n_bins) { return ew_info_off(asl) + 12u * n_bins <= kWpSymOff; }

// one field of <= 64 bits OR-ed into the stage at bit `pos` (val has no bits beyond its width); three dwords, the third zero for narrow fields
__device__ __forceinline__ void wp_put(uint32_t stg_addr, uint32_t pos, uint64_t val) {
  const uint32_t dw = pos >> 5, sh = pos & 31;
  const uint64_t lo = val << sh;
  uint32_t PCO_LDS* s = (uint32_t PCO_LDS*)(uintptr_t)(stg_addr + 4u * dw);
  const uint32_t hi = (uint32_t)((val >> 1) >> (63 - sh));   // val >> (64 - sh), 0 when sh == 0
#ifdef PCO_WP_PLAINST   // (timing experiments: wrong bytes)
  s[0] = (uint32_t)lo; s[1] = (uint32_t)(lo >> 32);
  if (__any(hi != 0)) s[2] = hi;
#else
  atomicOr((uint32_t*)&s[0], (uint32_t)lo);
  atomicOr((uint32_t*)&s[1], (uint32_t)(lo >> 32));
  atomicOr((uint32_t*)&s[2], hi);   // (unconditionally: a wave-uniform branch around it -- v_cmp, s_cbranch_vccz right behind the ds_or pair -- was one of the two
                                      //  forms with which, at 80 registers, a batch in a few came out garbled in its tANS section; see the note at the kernel)
#endif
}

// LDS accesses of one wave execute in program order: between a wave's own stage atomics, reads and stores nothing has to be waited for -- the compiler
// only must not move them across each other (enc_wave_sync's release fence would also wait for the wave's loads in flight: a round trip to HBM per batch)
#ifdef PCO_WP_FULLORDER
__device__ __forceinline__ void wp_lds_order() { __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
#else
__device__ __forceinline__ void wp_lds_order() { __asm__ volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
#endif

#ifdef PCO_WP_TIMING
__device__ unsigned long long g_wp_timing[16];   // block 7: helper wave 1: find, gather+load, puts, flush, barrier, iterations; walker: walk, barrier
#define WPT_NOW() __builtin_readcyclecounter()
#endif
#ifdef PCO_WP_ASSERT
__device__ uint32_t g_wp_err[8];   // scan mismatch, fields changed under the reader, stage not clean, ...
#endif
#ifdef PCO_WP_DEBUGSUM
__device__ uint32_t g_wp_dbg[16 * 1100 * 4];   // block 0: [item][step]: walker field sum, walker symbol sum, helper field sum, helper symbol sum (each: sum of value * (latent index + 1))
#endif
#ifdef PCO_WP_FULLSYNC
#define wd_barrier() __syncthreads()
#endif
// page_var (encode_fast.hip) for ONE lane's item: the same fields without the wave-uniform loads
__device__ __forceinline__ PageVar page_var_lane(const EncChunk PCO_GLOBAL* ch, uint32_t v, uint32_t page_n) {
  PageVar r;
  r.present = ch->v[v].present; r.n_bins = ch->v[v].n_bins; r.asl = ch->v[v].ans_size_log; r.max_ob = ch->v[v].max_ob;
  r.needs_ans = ch->v[v].needs_ans; r.trivial = ch->v[v].is_trivial;
  r.skip = v == 2 ? 0u : ch->v[v].lat_start;
  if (r.skip > page_n) r.skip = page_n;
  r.n_lat = page_n - r.skip;
  r.compact = ch->v[v].hist_path == 0 ? 1u : 0u;
  r.minv = (uint64_t)ch->v[v].minv;
  r.range = (uint64_t)ch->v[v].maxv - r.minv;
  r.rel = v != 0 && ch->c16_ok == 1 ? ch->c16_ref[v == 2 ? 1 : 0] : r.minv;
  return r;
}

// grid = ceil(items / 16), 320 threads: wave 0 walks, waves 1..4 find the symbols and pack
__global__ __launch_bounds__(64 * (1 + kWpHelpers)) __attribute__((amdgpu_waves_per_eu(PCO_WP_WAVES, PCO_WP_WAVES))) void enc_walkp_kernel(EncWorkspace ws, EncFast fx, uint32_t n_pages) {
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  uint8_t PCO_LDS* smem = enc_lds_base();
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t n_items = n_pages * ws.n_slots;
  typedef uint64_t __attribute__((aligned(2))) u64_align2;
  // ---- pass 1: does the block qualify?  Lane L of EVERY wave inspects item L & 15 (the four lanes L, L + 16, L + 32, L + 48 share an item's bins):
  //      sixteen items' facts in one round of dependent loads.  (Item after item with wave-uniform loads, the set-up of a block was 0.2-0.4 ms --
  //      a tenth of the kernel on BASELINE configs[1], half of it on pages of 16 384 numbers, whose walk is 64 steps.)  Every wave computes the
  //      same verdict. ----
  const uint32_t iq = lane & 15u, sub = lane >> 4;
  const uint32_t l_item = blockIdx.x * kWpQ + iq;
  const bool l_exists = l_item < n_items;
  const uint32_t l_p = l_exists ? l_item / ws.n_slots : 0u, l_sl = l_exists ? l_item % ws.n_slots : 0u;
  const uint32_t l_v = ws.slot_of_var[0] == l_sl ? 0u : (ws.slot_of_var[1] == l_sl ? 1u : 2u);
  const EncPage PCO_GLOBAL* l_pg = (const EncPage PCO_GLOBAL*)ws.pages + l_p;
  const uint32_t l_t = l_pg->chunk;
  const EncChunk PCO_GLOBAL* l_ch = (const EncChunk PCO_GLOBAL*)ws.chunks + l_t;
  const bool l_fast = l_exists && l_ch->status == PCO_GFX_OK && l_ch->fast_ok != 0 && !(l_pg->flags & kPageFlagMetaOnly);   // (page_is_fast, per lane)
  const uint32_t l_page_n = (uint32_t)l_pg->n;
  const PageVar l_pv = page_var_lane(l_ch, l_v, l_page_n);
  const bool l_writes = l_fast && l_pv.present && !l_pv.trivial;   // something of this item reaches the page body
  bool l_bad = false;
  if (l_writes) {
    // it must be the page's primary variable, walked here, looked up here, and alone
    if (l_v != 1 || !wd_walks(fx.fused, l_pv) || ws_walks(fx.fused, l_pv) || !wd_takes(fx.fused, l_pv) || !l_pv.needs_ans || l_pv.n_bins > 256 || !wp_fits(l_pv.asl, l_pv.n_bins)) l_bad = true;
    else if ((l_ch->v[0].present && !l_ch->v[0].is_trivial) || (l_ch->v[2].present && !l_ch->v[2].is_trivial)) l_bad = true;
  }
  {   // tANS bits + offset bits of a latent: at most min_renorm_bits + 1 + the bin's offset bits
    uint32_t worst = 0;
    if (l_writes && !l_bad) {
      const PlanRef plan = plan_ref(ws, l_t, l_v);
      for (uint32_t b = sub; b < l_pv.n_bins; b += 4) { const uint32_t w = ((plan.syminfo()[b] >> 14) & 15u) + 1u + plan.bob()[b]; worst = w > worst ? w : worst; }
    }
    { const uint32_t o = (uint32_t)__shfl_xor((int)worst, 16, 64); worst = worst > o ? worst : o; }
    { const uint32_t o = (uint32_t)__shfl_xor((int)worst, 32, 64); worst = worst > o ? worst : o; }
    if (worst > 16u) l_bad = true;
  }
  const bool ok = (fx.fused & kFusedLookups) != 0 && !__any(l_bad);
  if (threadIdx.x == 0) fx.wp_block[blockIdx.x] = ok ? 1u : 0u;
  if (!ok) {   // enc_walkd_kernel takes the block; the pages whose primary variable sits here are marked "no body"
    if (threadIdx.x < kWpQ) {
      const uint32_t item = blockIdx.x * kWpQ + threadIdx.x;
      if (item < n_items && ws.slot_of_var[1] == item % ws.n_slots) fx.body[2ull * (item / ws.n_slots)] = 0;
    }
    return;
  }
  // ---- pass 2: the items' tables (item q by wave q mod 9), and what a lane keeps of its item: a walker lane the item of its quad (lane >> 2), a
  //      lane of packing wave w the item (w - 1) * kWpH + (lane & (kWpH - 1)) -- fetched from the lane that inspected the item ----
  const uint32_t l_nb = l_writes ? (l_pv.n_lat + kBatchN - 1) / kBatchN : 0u;
  const uint32_t max_nb = wave_max_u32(l_nb);
  if (l_fast && !l_writes && sub == 0 && wave == 0) {
    // (as enc_walkd_kernel: nobody else writes the states of a variable that needs no walk; a page whose primary variable is trivial has no body of ours)
    if (l_pv.present && (l_pv.n_bins <= 1 || l_pv.n_lat == 0)) for (uint32_t jj = 0; jj < 4; jj++) fx.fstate[((uint64_t)l_p * 3 + l_v) * 4 + jj] = 1u << l_pv.asl;
    if (l_v == 1) fx.body[2ull * l_p] = 0;
  }
  const uint32_t l_info_off = ew_info_off(l_pv.asl), l_cpk_off = l_info_off + 8u * l_pv.n_bins;
  const uint64_t l_start = (uint64_t)l_pg->start;
  const uint64_t l_at = l_start + l_pv.skip + 16ull * l_pg->page_idx, l_clat = l_start + l_pv.skip;   // (fast_at, per lane)
  auto rl = [](uint32_t x, uint32_t q) { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)q); };
  for (uint32_t q = wave; q < kWpQ; q += 1 + kWpHelpers) {   // (uniform per wave) this wave's share of the tables
    if (rl(l_writes ? 1u : 0u, q) == 0) continue;
    const uint32_t t = rl(l_t, q), n_bins = rl(l_pv.n_bins, q), asl = rl(l_pv.asl, q);
    const uint64_t rel = ((uint64_t)rl((uint32_t)(l_pv.rel >> 32), q) << 32) | rl((uint32_t)l_pv.rel, q);
    const uint32_t info_off = ew_info_off(asl), cpk_off = info_off + 8u * n_bins;
    const PlanRef plan = plan_ref(ws, t, 1);
    uint8_t PCO_LDS* slot = smem + wp_slot(q);
    const uint32_t T = 1u << asl;
    for (uint32_t i = lane; i < T; i += 64) ((uint16_t PCO_LDS*)slot)[i] = plan.next_states()[i];
    for (uint32_t b = lane; b < n_bins; b += 64) {
      const uint32_t si = plan.syminfo()[b];   // cutoff(14) | min_renorm_bits(4) << 14 | (row + 8192)(14) << 18
      const uint32_t cutoff = si & 0x3fffu, minb = (si >> 14) & 15u, row = (si >> 18) - 8192u;
      const uint32_t row_addr = lds0 + wp_slot(q) + 2u * row;
      ((uint64_t PCO_LDS*)(slot + info_off))[b] = (uint64_t)(((minb + 1u) << 16) - cutoff) | ((uint64_t)row_addr << 32);
      ((uint32_t PCO_LDS*)(slot + cpk_off))[b] = ((uint32_t)(plan.blower()[b] - rel) & 0xffffu) | ((uint32_t)plan.bob()[b] << 16);   // (16-bit latents are relative to rel)
    }
    for (uint32_t i = lane; i < kWpStgDwords; i += 64) ((uint32_t PCO_LDS*)(slot + kWpStgOff))[i] = 0;
  }
  const uint32_t my_q = wave == 0 ? lane >> 2 : (wave - 1) * kWpH + (lane & (kWpH - 1));
  auto from_item = [&](uint32_t x) { return (uint32_t)__shfl((int)x, (int)my_q, 64); };   // (lane q < 16 inspected item q)
  const uint32_t my_n_lat = from_item(l_writes ? l_pv.n_lat : 0u), my_T = 1u << from_item(l_pv.asl), my_p = from_item(l_p), my_task = from_item(l_t);
  const uint32_t my_info_off = from_item(l_info_off), my_cpk_off = from_item(l_cpk_off);
  const uint64_t my_at = ((uint64_t)from_item((uint32_t)(l_at >> 32)) << 32) | from_item((uint32_t)l_at);
  const uint64_t my_clat = ((uint64_t)from_item((uint32_t)(l_clat >> 32)) << 32) | from_item((uint32_t)l_clat);
  const uint32_t my_nb = (my_n_lat + kBatchN - 1) / kBatchN;
  if (max_nb == 0) return;
  wd_barrier();
  if (wave != 0) {
    // ================= a gathering + packing wave: four items =================
    // Everything that depends on the item alone is wave-uniform and lives in SGPRs (the item loops are unrolled): the kernel is bound by the
    // VALU instructions of these waves -- two walkers and eight of them share a CU's four SIMDs, ~7100 issue cycles per batch of the walk --
    // so nothing is fetched from an owner lane per step, the loads take scalar bases, and full batches (all but a page's last) take a
    // path without per-latent predicates.
    const bool mine = my_n_lat != 0;
    const uint64_t my_clat_p = mine ? (uint64_t)(uintptr_t)(clat_ptr(ws, my_task, 1) + my_clat) : (uint64_t)(uintptr_t)fx.vlut;
    const uint32_t my_lut_off = mine ? (my_task * ws.n_slots + ws.slot_of_var[1]) * kDirectHistRange : 0u;   // (u16 elements)
    // the page's body region: the dwords of its tANS-field scratch (and of the 32-byte gap behind it), written downwards from the end
    const uint64_t my_reg = ((uint64_t)(uintptr_t)(fansw_ptr(ws, fx, my_task, 1) + my_at) + 3ull) & ~3ull;
    auto reg_bits_of = [](uint32_t n_lat) { return n_lat != 0 ? (uint32_t)(((2ull * n_lat + 8ull) & ~3ull) * 8ull) : 0u; };
    auto bcast = [](uint32_t x, uint32_t q) { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)q); };
    auto bcast64 = [&](uint64_t x, uint32_t q) { return ((uint64_t)bcast((uint32_t)(x >> 32), q) << 32) | bcast((uint32_t)x, q); };
    const uint32_t q0 = (wave - 1) * kWpH;
    uint32_t s_nlat[kWpH], s_nb[kWpH], s_cpk[kWpH], s_rot2[kWpH], s_low[kWpH], s_carry[kWpH], s_slot[kWpH];
    uint64_t s_clat[kWpH], s_lut[kWpH], s_reg[kWpH];
#pragma unroll
    for (uint32_t q = 0; q < kWpH; q++) {
      s_nlat[q] = bcast(my_n_lat, q); s_nb[q] = (s_nlat[q] + kBatchN - 1) / kBatchN;
      s_slot[q] = lds0 + wp_slot(q0 + q); s_cpk[q] = s_slot[q] + bcast(my_cpk_off, q);
      const uint32_t lo = bcast(my_lut_off, q);
      s_lut[q] = (uint64_t)(uintptr_t)fx.vlut + 2ull * lo; s_rot2[q] = vlut_rot(lo / kDirectHistRange) * 0x10001u;   // the table's rotation, for both 16-bit halves of a dword (no carry: latents < 2^15)
      s_clat[q] = bcast64(my_clat_p, q); s_reg[q] = bcast64(my_reg, q); s_low[q] = reg_bits_of(s_nlat[q]); s_carry[q] = 0;
    }
    auto batch_at = [&](uint32_t q, uint32_t it) { return it < s_nb[q] ? s_nb[q] - 1 - it : 0u; };   // (no batch at this step: batch 0 is read, and never used)
    auto count_at = [&](uint32_t q, uint32_t it) { const uint32_t base = batch_at(q, it) * kBatchN; return it < s_nb[q] ? (s_nlat[q] - base < kBatchN ? s_nlat[q] - base : kBatchN) : 0u; };
    // (branch-free on purpose, as in enc_walkd_kernel.  The 8 bytes of a page's last, partial batch may run into the scratch behind the page: those latents count for nothing)
    auto load_batches = [&](uint32_t it, uint64_t (&w)[kWpH]) {   // the 4 latents a lane owns of every item's batch
#pragma unroll
      for (uint32_t q = 0; q < kWpH; q++) w[q] = __builtin_nontemporal_load((const u64_align2 PCO_GLOBAL*)((const uint8_t PCO_GLOBAL*)(uintptr_t)(s_clat[q] + 2ull * batch_at(q, it) * kBatchN) + 8 * lane));
    };
    auto gather = [&](const uint64_t (&w)[kWpH], uint32_t (&e)[kWpH][4]) {   // value -> bin | offset bits << 8 through enc_vlut_kernel's tables
#pragma unroll
      for (uint32_t q = 0; q < kWpH; q++) {
        const uint16_t PCO_GLOBAL* lut = (const uint16_t PCO_GLOBAL*)(uintptr_t)s_lut[q];
        const uint32_t lo = (uint32_t)w[q] + s_rot2[q], hi = (uint32_t)(w[q] >> 32) + s_rot2[q];
        e[q][0] = lut[lo & (kDirectHistRange - 1)]; e[q][1] = lut[(lo >> 16) & (kDirectHistRange - 1)];
        e[q][2] = lut[hi & (kDirectHistRange - 1)]; e[q][3] = lut[(hi >> 16) & (kDirectHistRange - 1)];
      }
    };
    // pipeline at step s: latents of step s + 2 requested, table entries of step s + 1 gathered; symbols of step s made and its offsets put together
    // (they do not depend on the walk); the tANS fields of step s - 2 put together and the batch written
    uint64_t xl[kWpH], xg[kWpH], xs[kWpH];      // latents: loaded for s + 2 | gathered for s + 1 | this step's
    uint32_t e1[kWpH][4];                        // entries: gathered (during step s - 1) for step s, then (during step s) for s + 1
    uint64_t o0[kWpH], oa[kWpH], ob[kWpH];      // a lane's four offsets, concatenated: steps s, s - 1, s - 2
    uint32_t n0[kWpH], na[kWpH], nb2[kWpH];     // ... and how many bits that is
#pragma unroll
    for (uint32_t q = 0; q < kWpH; q++) { xl[q] = xg[q] = xs[q] = 0; o0[q] = oa[q] = ob[q] = 0; n0[q] = na[q] = nb2[q] = 0; e1[q][0] = e1[q][1] = e1[q][2] = e1[q][3] = 0; }
    load_batches(0, xg);
    gather(xg, e1);
    if (1 < max_nb) load_batches(1, xl);
    // symbols and offsets of one batch of one item (entries in e1[q], latents in xs[q])
    uint32_t dbg_it = 0; (void)dbg_it;
    auto find_item = [&](auto full_tag, uint32_t q, uint32_t hb, uint32_t cnt) {
      constexpr bool kFull = decltype(full_tag)::value;
      uint32_t e[4] = {e1[q][0], e1[q][1], e1[q][2], e1[q][3]};
      if (!kFull) {
#pragma unroll
        for (int k = 0; k < 4; k++) e[k] = 4 * lane + k < cnt ? e[k] : 0u;
      }
      // the walker's chain j reads byte k of the dword at 16 blk + 4 j for its step 4 blk + k: a 4 x 4 byte transpose inside the quad (in registers:
      // the LDS pipe, not the VALU, is what this kernel runs out of), then one conflict-free dword per lane
#ifdef PCO_WP_V3
      { const uint32_t buf = s_slot[q] + kWpSymOff + (hb & 1) * 256 + 16u * (lane >> 2) + (lane & 3u);
        for (int k = 0; k < 4; k++) *(uint8_t PCO_LDS*)(uintptr_t)(buf + 4 * k) = (uint8_t)e[k]; }
      const uint32_t e01 = e[0] | (e[1] << 16), e23 = e[2] | (e[3] << 16);
      if (false)
#else
      const uint32_t e01 = e[0] | (e[1] << 16), e23 = e[2] | (e[3] << 16);
#endif
      *(uint32_t PCO_LDS*)(uintptr_t)(s_slot[q] + kWpSymOff + (hb & 1) * 256 + 4 * lane) = quad_transpose_u8(__builtin_amdgcn_perm(e23, e01, 0x06040200u), lane & 3);
#ifdef PCO_WP_DEBUGSUM
      { uint32_t sm = 0; for (int k = 0; k < 4; k++) sm += (e[k] & 0xffu) * (4 * lane + k + 1); sm = wave_sum(sm); if (blockIdx.x == 0 && lane == 0) g_wp_dbg[((q0 + q) * 1100 + dbg_it) * 4 + 3] = sm; }
#endif
      uint32_t c[4], w[4], dv[4];
#pragma unroll
#ifdef PCO_WP_NOCPK   // (timing experiments: wrong bytes)
      for (int k = 0; k < 4; k++) c[k] = ((e[k] >> 8) << 16) | (e[k] & 0xffu);
#else
      for (int k = 0; k < 4; k++) c[k] = *(const uint32_t PCO_LDS*)(uintptr_t)(s_cpk[q] + 4u * (e[k] & 0xffu));   // lower (16 bits, relative) | offset bits << 16
#endif
#pragma unroll
      for (int k = 0; k < 4; k++) {
        w[k] = (kFull || 4 * lane + k < cnt) ? c[k] >> 16 : 0u;
        const uint32_t xk = (uint32_t)(xs[q] >> (16 * k)) & 0xffffu;
        dv[k] = __builtin_amdgcn_ubfe(xk - c[k], 0u, w[k]);   // (w <= 15; c's high half only disturbs bits 16 and up)
      }
      const uint32_t o01 = w[0] + w[1];
      o0[q] = (uint64_t)(dv[0] | (dv[1] << w[0])) | ((uint64_t)(dv[2] | (dv[3] << w[2])) << o01);   // <= 60 bits, two fields at a time in 32
      n0[q] = o01 + w[2] + w[3];
    };
    // one batch of one item: its tANS fields (left by the walker), then its offsets (ob / nb2), at bits [low - all, low) of the page's body region.
    // In two parts, so that the four items' field reads, scans and stage atomics are scheduled into one another (no branch between them; a
    // batch without bits ORs nothing and flushes nothing) before the four flushes, which are loops.
    uint32_t p_new[kWpH];   // (uniform) where each item's batch starts
#ifdef PCO_WP_ASSERT
    uint64_t dbg_f64[kWpH]; uint32_t dbg_pb[kWpH];
#endif
    auto pack_puts = [&](auto full_tag, uint32_t q, uint32_t pb, uint32_t cnt) {
      constexpr bool kFull = decltype(full_tag)::value;
      const uint32_t stg = s_slot[q] + kWpStgOff;
      uint32_t f[4], nb[4], fv[4];
#ifdef PCO_WP_V2
      { const uint32_t fa = s_slot[q] + kWpAnsOff + (pb & 1) * 512 + 32u * (lane >> 2) + 2u * (lane & 3u);
        for (int k = 0; k < 4; k++) f[k] = *(const uint16_t PCO_LDS*)(uintptr_t)(fa + 8 * k); }
      if (false)
#endif
      {   // the walker left (block, chain) units of four steps: lane 4 blk + j reads unit (blk, j), the quad transposes
        const uint64_t w = *(const uint64_t PCO_LDS*)(uintptr_t)(s_slot[q] + kWpAnsOff + (pb & 1) * 512 + 8 * lane);
        uint32_t a = (uint32_t)w, b = (uint32_t)(w >> 32);
        quad_transpose_u16(a, b, lane & 3);
        f[0] = a & 0xffffu; f[1] = a >> 16; f[2] = b & 0xffffu; f[3] = b >> 16;
      }
#ifdef PCO_WP_DEBUGSUM
      { uint32_t sm = 0; for (int k = 0; k < 4; k++) sm += ((kFull || 4 * lane + k < cnt) ? f[k] : 0u) * (4 * lane + k + 1); sm = wave_sum(sm); if (blockIdx.x == 0 && lane == 0) g_wp_dbg[((q0 + q) * 1100 + dbg_it - 2) * 4 + 2] = sm; }
#endif
#pragma unroll
      for (int k = 0; k < 4; k++) {
        nb[k] = (kFull || 4 * lane + k < cnt) ? f[k] >> 12 : 0u;
        fv[k] = __builtin_amdgcn_ubfe(f[k], 0u, nb[k]);                                                            // <= 12 bits
      }
      const uint32_t a01 = nb[0] + nb[1], abits = a01 + nb[2] + nb[3], obits = nb2[q];
      const uint64_t acc = (uint64_t)(fv[0] | (fv[1] << nb[0])) | ((uint64_t)(fv[2] | (fv[3] << nb[2])) << a01);   // <= 48 bits
      const uint32_t both = abits | (obits << 16);
      const uint32_t incl = wave_incl_scan(both), total = wave_last(incl), excl = incl - both;
      const uint32_t ans_total = total & 0xffffu, all = ans_total + (total >> 16);
      const uint32_t new_low = s_low[q] - all, sh0 = new_low & 31u;
      p_new[q] = new_low;
#ifdef PCO_WP_ASSERT
      {
        const uint32_t prev = __shfl_up(incl, 1, 64);
        if (excl != (lane == 0 ? 0u : prev)) atomicAdd(&g_wp_err[0], 1u);
        const uint64_t z = *(const uint64_t PCO_LDS*)(uintptr_t)(stg + 8 * lane);
        if (z != 0) atomicAdd(&g_wp_err[2], 1u);
        if (lane < 2 && *(const uint32_t PCO_LDS*)(uintptr_t)(stg + 512 + 4 * lane) != 0) atomicAdd(&g_wp_err[2], 1u);
        dbg_f64[q] = (uint64_t)f[0] | ((uint64_t)f[1] << 16) | ((uint64_t)f[2] << 32) | ((uint64_t)f[3] << 48); dbg_pb[q] = pb;
      }
#endif
#ifndef PCO_WP_NOPUT   // (timing experiments: wrong bytes)
      wp_put(stg, sh0 + (excl & 0xffffu), acc);
      wp_put(stg, sh0 + ans_total + (excl >> 16), ob[q]);
#endif
    };
    // the dwords the batch completes go out -- the highest with what the batch above left of it; the lowest stays as the carry unless the batch starts on a dword
    auto pack_flush = [&](uint32_t q) {
      const uint32_t stg = s_slot[q] + kWpStgOff;
#ifdef PCO_WP_ASSERT
      {
        const uint64_t w = *(const uint64_t PCO_LDS*)(uintptr_t)(s_slot[q] + kWpAnsOff + (dbg_pb[q] & 1) * 512 + 8 * lane);
        uint32_t a = (uint32_t)w, b = (uint32_t)(w >> 32);
        quad_transpose_u16(a, b, lane & 3);
        if ((((uint64_t)b << 32) | a) != dbg_f64[q] && s_nlat[q] - dbg_pb[q] * kBatchN >= kBatchN) atomicAdd(&g_wp_err[1], 1u);
      }
#endif
      const uint32_t low = s_low[q], new_low = p_new[q], sh0 = new_low & 31u, lo_dw = new_low >> 5;
      const uint32_t n_dw = ((low - 1u) >> 5) - lo_dw + 1u, i0 = sh0 ? 1u : 0u, top = (low & 31u) ? n_dw - 1u : 0xffffffffu;   // (no bits: n_dw is 0, or 1 with the carry staying where it is)
      uint32_t nc = 0;   // what stays behind of the lowest dword (read before the flush zeroes the stage)
      if (sh0) nc = uni(*(const uint32_t PCO_LDS*)(uintptr_t)stg) | (top == 0u ? s_carry[q] : 0u);   // (a batch inside one dword: the carry stays where it is)
      wp_lds_order();
      uint32_t PCO_GLOBAL* g = (uint32_t PCO_GLOBAL*)(uintptr_t)s_reg[q] + lo_dw;
#ifndef PCO_WP_NOFLUSH
      typedef uint64_t __attribute__((aligned(4))) u64_align4;
      for (uint32_t r = 0; r < n_dw; r += 128) {   // two dwords per lane
        const uint32_t i = r + 2 * lane;
        if (i < n_dw) {
          uint64_t PCO_LDS* sp = (uint64_t PCO_LDS*)((uint32_t PCO_LDS*)(uintptr_t)stg + i);
          uint64_t v = *sp; *sp = 0;
          v |= i == top ? (uint64_t)s_carry[q] : (i + 1 == top ? (uint64_t)s_carry[q] << 32 : 0ull);
          if (i >= i0 && i + 1 < n_dw) *(u64_align4 PCO_GLOBAL*)(g + i) = v;
          else if (i >= i0) g[i] = (uint32_t)v;                       // (the batch's last dword alone)
          else if (i + 1 < n_dw) g[i + 1] = (uint32_t)(v >> 32);      // (dword 0 stays behind as the carry)
        }
      }
#endif
      s_low[q] = new_low; s_carry[q] = nc;
      wp_lds_order();
    };
#ifdef PCO_WP_TIMING
    unsigned long long t_find = 0, t_gath = 0, t_puts = 0, t_flush = 0, t_bar = 0, t0 = WPT_NOW(), t1;
#define WPT_ACC(x) do { __builtin_amdgcn_sched_barrier(0); t1 = WPT_NOW(); x += t1 - t0; t0 = t1; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WPT_ACC(x) do { } while (0)
#endif
    for (uint32_t it = 0; it < max_nb + 2; it++) {
      // ---- rotate the pipeline ----
#pragma unroll
      for (uint32_t q = 0; q < kWpH; q++) { ob[q] = oa[q]; nb2[q] = na[q]; oa[q] = o0[q]; na[q] = n0[q]; xs[q] = xg[q]; xg[q] = xl[q]; }
      // ---- symbols and offsets of step `it` ----
#ifdef PCO_WP_NOHELP   // (timing experiments: wrong bytes)
      if (it > 1000000u) {
#else
      if (it < max_nb) {
#endif
        uint32_t cn[kWpH]; bool all_full = true;
#pragma unroll
        for (uint32_t q = 0; q < kWpH; q++) { cn[q] = count_at(q, it); all_full = all_full && cn[q] == kBatchN; }
        if (all_full) {   // straight-line: the four items' LDS round trips overlap
#pragma unroll
          for (uint32_t q = 0; q < kWpH; q++) find_item(std::true_type{}, q, batch_at(q, it), kBatchN);
        } else {
#pragma unroll
          for (uint32_t q = 0; q < kWpH; q++) { if (cn[q] != 0) find_item(std::false_type{}, q, batch_at(q, it), cn[q]); else { o0[q] = 0; n0[q] = 0; } }
        }
      }
      WPT_ACC(t_find);
      __builtin_amdgcn_sched_barrier(0);
      if (it + 1 < max_nb) gather(xg, e1);
      __builtin_amdgcn_sched_barrier(0);
      if (it + 2 < max_nb) load_batches(it + 2, xl);
      __builtin_amdgcn_sched_barrier(0);
      WPT_ACC(t_gath);
      // ---- pack the batch of step it - 2 ----
#ifndef PCO_WP_NOPACK
      if (it >= 2) {
        uint32_t cn[kWpH]; bool all_full = true;
#pragma unroll
        for (uint32_t q = 0; q < kWpH; q++) { cn[q] = count_at(q, it - 2); all_full = all_full && cn[q] == kBatchN; }
#ifdef PCO_WP_V4
        all_full = false;
#endif
        if (all_full) {
#pragma unroll
          for (uint32_t q = 0; q < kWpH; q++) pack_puts(std::true_type{}, q, batch_at(q, it - 2), kBatchN);
          wp_lds_order();
          WPT_ACC(t_puts);
#pragma unroll
          for (uint32_t q = 0; q < kWpH; q++) pack_flush(q);
          WPT_ACC(t_flush);
        } else {
#pragma unroll
          for (uint32_t q = 0; q < kWpH; q++) if (cn[q] != 0) { pack_puts(std::false_type{}, q, batch_at(q, it - 2), cn[q]); wp_lds_order(); pack_flush(q); }
        }
      }
#endif
      WPT_ACC(t_flush);
      wd_barrier();
      WPT_ACC(t_bar);
    }
#ifdef PCO_WP_TIMING
    if (blockIdx.x == 7 && wave == 1 && lane == 0) { g_wp_timing[0] = t_find; g_wp_timing[1] = t_gath; g_wp_timing[2] = t_puts; g_wp_timing[3] = t_flush; g_wp_timing[4] = t_bar; g_wp_timing[5] = max_nb; }
#endif
    // the body's lowest, partial dword, and the page's record: bits | flag, first bit (counted from fx.answ)
#pragma unroll
    for (uint32_t q = 0; q < kWpH; q++) {
      if (s_nlat[q] == 0) continue;
      const uint32_t pq = bcast(my_p, q), rb = reg_bits_of(s_nlat[q]);
      if (lane == 0) {
        if (s_low[q] & 31u) ((uint32_t PCO_GLOBAL*)(uintptr_t)s_reg[q])[s_low[q] >> 5] = s_carry[q];
        fx.body[2ull * pq] = (uint64_t)(rb - s_low[q]) | kWpBodyFlag;
        fx.body[2ull * pq + 1] = (s_reg[q] - (uint64_t)(uintptr_t)fx.answ) * 8ull + s_low[q];
      }
    }
    return;
  }
  // ================= the walker wave (enc_walkd_kernel's walk; the fields stay in LDS) =================
#ifndef PCO_WP_NOPRIO
  __builtin_amdgcn_s_setprio(3);   // (the walker's chain of dependent steps sets the block's pace; the packing waves of both blocks share its SIMD)
#endif
  const uint32_t j = lane & 3;
  const uint32_t slice = lds0 + wp_slot(my_q);
  const uint32_t info_addr = slice + my_info_off, symbuf = slice + kWpSymOff, ansbuf = slice + kWpAnsOff;
  uint32_t state = my_T;
  wd_barrier();   // (the gathering waves' it = 0)
#ifdef PCO_WP_TIMING
  unsigned long long w_walk = 0, w_bar = 0, w0 = WPT_NOW(), w1;
#endif
  for (uint32_t it = 0; it < max_nb; it++) {
#ifdef PCO_WP_NOWALK   // (timing experiments: wrong bytes)
    if (it > 1000000u) {
#else
    if (it < my_nb) {
#endif
      const uint32_t b = my_nb - 1 - it, base = b * kBatchN, cnt = my_n_lat - base < kBatchN ? my_n_lat - base : kBatchN;
      const uint32_t buf = symbuf + (b & 1) * 256, abuf = ansbuf + (b & 1) * 512 + 8 * j;
      uint32_t bits_acc = 0;
#ifdef PCO_WP_DEBUGSUM
      uint32_t dbg_f = 0, dbg_s = 0;
#endif
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
#ifdef PCO_WP_DEBUGSUM
              dbg_s += ((sd >> (8 * k)) & 0xffu) * (16 * blk + 4 * k + j + 1);
#endif
            }
          }
#ifdef PCO_WP_DEBUGSUM
          for (int k = 0; k < 4; k++) dbg_f += (uint32_t)((out >> (16 * k)) & 0xffffu) * (16 * blk + 4 * k + j + 1);
#endif
          *(uint64_t PCO_LDS*)(uintptr_t)(abuf + 32 * blk) = out;
        }
      } else {               // a full batch, software-pipelined as in enc_walk_kernel
        uint32_t sd = *(const uint32_t PCO_LDS*)(uintptr_t)(buf + 16 * 15 + 4 * j);
        uint32_t nsd = *(const uint32_t PCO_LDS*)(uintptr_t)(buf + 16 * 14 + 4 * j);
#ifdef PCO_WP_DEBUGSUM
        uint32_t nsd_prev = nsd;
#endif
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
          *(uint64_t PCO_LDS*)(uintptr_t)(abuf + 32 * blk) = (uint64_t)(o0 | (o1 << 16)) | ((uint64_t)o23 << 32);
#ifdef PCO_WP_DEBUGSUM
          dbg_f += o0 * (16 * blk + j + 1) + o1 * (16 * blk + 4 + j + 1) + o2 * (16 * blk + 8 + j + 1) + o3 * (16 * blk + 12 + j + 1);
          for (int k = 0; k < 4; k++) dbg_s += ((sd >> (8 * k)) & 0xffu) * (16 * blk + 4 * k + j + 1);
          sd = nsd_prev; nsd_prev = nsd;
#endif
          __builtin_amdgcn_sched_barrier(0);
          i0 = n0; i1 = n1; i2 = n2; i3 = n3;
        }
      }
#ifdef PCO_WP_DEBUGSUM
      dbg_f += quad_dpp<0xB1>(dbg_f); dbg_f += quad_dpp<0x4E>(dbg_f); dbg_s += quad_dpp<0xB1>(dbg_s); dbg_s += quad_dpp<0x4E>(dbg_s);
      if (blockIdx.x == 0 && j == 0) { g_wp_dbg[(my_q * 1100 + it) * 4 + 0] = dbg_f; g_wp_dbg[(my_q * 1100 + it) * 4 + 1] = dbg_s; }
#endif
    }
#ifdef PCO_WP_TIMING
    w1 = WPT_NOW(); w_walk += w1 - w0; w0 = w1;
#endif
    wd_barrier();
#ifdef PCO_WP_TIMING
    w1 = WPT_NOW(); w_bar += w1 - w0; w0 = w1;
#endif
  }
#ifdef PCO_WP_TIMING
  if (blockIdx.x == 7 && lane == 0) { g_wp_timing[8] = w_walk; g_wp_timing[9] = w_bar; }
#endif
  wd_barrier();   // (the packing waves' last step)
  if (my_n_lat > 0) fx.fstate[((uint64_t)my_p * 3 + 1) * 4 + j] = state;
}

// ---------------------------------------------------------------------------------------------------------
// enc_place_kernel: the bodies enc_walkp_kernel left right-aligned in the field scratch, moved behind their pages' heads.  The head ends on
// a byte (enc_scan_kernel: run_start[0]), the body starts at any bit of the scratch: every dst dword is one v_alignbit of two source dwords.
// The first and the last dword of the body in dst are shared (with the head; with the padding) and are OR-ed into what enc_scan_kernel zeroed.
// grid pages * pieces, 256 threads: 16 bytes of dst per thread and step; `pieces` (1 .. 16) from the longest page of the call, so that a block has
// eight steps' worth to move (sixteen blocks per page of 16 384 numbers were mostly launch overhead: 3.4 ms per 131 072 pages, 1.2 with one).
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t kPlaceMaxPieces = 16;
__host__ __device__ constexpr uint32_t place_pieces(uint64_t page_max) { return (uint32_t)(2 * page_max / 32768 < 1 ? 1 : (2 * page_max / 32768 > kPlaceMaxPieces ? kPlaceMaxPieces : 2 * page_max / 32768)); }
__global__ __launch_bounds__(256) void enc_place_kernel(EncWorkspace ws, EncFast fx, uint32_t n_pages, uint32_t pieces) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef u32x4 __attribute__((aligned(4))) u32x4_a4;
  const uint32_t p = blockIdx.x / pieces, piece = blockIdx.x % pieces;
  if (p >= n_pages) return;
  EncPage PCO_GLOBAL* pg = (EncPage PCO_GLOBAL*)ws.pages + p;
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + uni(pg->chunk);
  if (!page_is_fast(ch, pg)) return;
  const uint64_t rec = uni(fx.body[2ull * p]);
  if (!(rec & kWpBodyFlag) || (uni(pg->pad) & 1u) != 0) return;
  const uint64_t bits = rec & ~kWpBodyFlag, sbit = uni(fx.body[2ull * p + 1]);
  if (bits == 0) return;
  const uint64_t head = uni(fx.run_start[(uint64_t)p * fx.run_stride]), end = head + bits;   // dst bits [head, end)
  uint32_t PCO_GLOBAL* dst32 = (uint32_t PCO_GLOBAL*)pg->dst;
  const uint32_t PCO_GLOBAL* src32 = (const uint32_t PCO_GLOBAL*)fx.answ;
  const uint64_t d_first = head >> 5, d_last = (end - 1) >> 5;
  // src bit of dst bit x: sbit + (x - head)
  const uint64_t g0 = d_first >> 2, g1 = d_last >> 2;   // 16-byte groups of dst touched
  for (uint64_t g = g0 + (uint64_t)piece * 256 + threadIdx.x; g <= g1; g += (uint64_t)pieces * 256) {
    const uint64_t d0 = 4 * g;
    if (d0 > d_first && d0 + 3 < d_last) {   // four interior dwords
      const uint64_t s = sbit + (32 * d0 - head);
      const uint32_t PCO_GLOBAL* sp = src32 + (s >> 5);
      const u32x4 a = *(const u32x4_a4 PCO_GLOBAL*)sp; const uint32_t a4 = sp[4];
      const uint32_t r = (uint32_t)s & 31u;
      u32x4 o;
      o.x = __builtin_amdgcn_alignbit(a.y, a.x, r); o.y = __builtin_amdgcn_alignbit(a.z, a.y, r);
      o.z = __builtin_amdgcn_alignbit(a.w, a.z, r); o.w = __builtin_amdgcn_alignbit(a4, a.w, r);
      *(u32x4 PCO_GLOBAL*)(dst32 + d0) = o;
    } else {
      for (uint32_t k = 0; k < 4; k++) {
        const uint64_t d = d0 + k;
        if (d < d_first || d > d_last) continue;
        const uint64_t lo = 32 * d > head ? 32 * d : head, hi = 32 * d + 32 < end ? 32 * d + 32 : end;
        const uint32_t nb = (uint32_t)(hi - lo);
        const uint64_t s = sbit + (lo - head);
        const uint32_t PCO_GLOBAL* sp = src32 + (s >> 5);
        const uint64_t w = (uint64_t)sp[0] | ((uint64_t)sp[1] << 32);
        uint32_t val = (uint32_t)(w >> (s & 31));
        if (nb < 32) val &= (1u << nb) - 1u;
        val <<= (uint32_t)(lo - 32 * d);
        if (nb < 32 || d == d_first) atomicOr((uint32_t*)(dst32 + d), val); else dst32[d] = val;
      }
    }
  }
}

}  // namespace pcogfx
