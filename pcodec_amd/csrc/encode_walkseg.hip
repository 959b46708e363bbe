// encode_walkseg.hip -- the reverse tANS walk of a long page variable, cut into sixteen segments that are walked side by side (round 5).
//
// ans/encoding.rs:65-87: one step maps state -> next_states[symbol][(state >> bits) - weight], bits = min_renorm_bits + (state >= cutoff).
// The chain has no other input than the symbols (no bit cursor: chunk_latent_compressor.rs:96-132 writes the bits afterwards), so a segment
// can be walked without its predecessor as soon as its entry state is known -- and after a few symbols it IS known, whatever it was:
//   * as the state runs over [T, 2T) the row index (state >> bits) - weight runs over [0, weight) upwards from some offset, wrapping once
//     (the states from the cutoff on shift one bit more and land on the low indices), and a symbol's row of next states ascends with the
//     index (encoding.rs pushes table_size + state_idx in table order).  So one step maps the CIRCLE of states onto itself weakly
//     order-preserving, and so does any sequence of steps: the image of an arc [A -> B] of states lies within the arc [f(A) -> f(B)].
//   * A segment therefore starts with the arc of ALL states, A = T and B = 2T - 1, and steps both ends.  While the arc does not pass the
//     2T - 1 -> T cut its ends reach the same row entry only when everything between them does: A == B, and from there on the state is
//     the true one.  An arc that passes the cut and whose ends reach the same row entry covers every entry of the row: it is reset to the
//     arc of all states before the step (a superset).  Exactness never depends on the symbols being kind: a segment whose ends have not met
//     simply has not started yet.
//   * What a segment walked before its ends met is walked again afterwards (whole 256-latent batches, so that the batches' bit totals are
//     written once), from the exit state of the segment above -- which by then is exact, unless that segment never met either: then it is
//     walked again first.  In the worst case (a table that never forgets: four bins of weight T / 4) this is the old chain, a segment at a time.
// Measured full-set meeting times (scripts/ans_merge_sim.py, the reference's spread + encoder on this repo's workloads): the headline's
// 20-bin table median 2 steps, p99 125; float-mult decimals 10 / 161; the lookbacks of configs[3] (weights 1023 + 1) 732 / 4841.
//
// One block of two waves per (page, variable): the tables ONCE in LDS (enc_walkd_kernel keeps sixteen copies, one per item: 72 KB a block, two
// blocks per CU, 512 walking waves on the chip whatever the call's size), sixteen symbol buffers behind them -- 12.8 KB, twelve blocks per CU.
// The walker wave's quad q walks segment q; the gathering wave finds the symbols of the sixteen segments' batches as enc_walkd_kernel's four do
// for its sixteen items, four segments at a time.  Used for calls of up to kWsMaxItems items: on a full chip the gathering is 3.1 ms of
// instruction work by itself, which the unsegmented kernel hides under its 3.4 ms of latency (profiles/r05_walkseg_scaling.txt).
// A second walk that does not arrive at the state the first one had where its ends met fails the chunk (it cannot happen; it is checked).
namespace pcogfx {

constexpr uint32_t kWsSymBase = 4096, kWsExOff = kWsSymBase + kWsSegs * 512, kWsLdsBytes = kWsExOff + 2 * kWsSegs * 4 * 4;   // tables | u8[16][2][256] symbols | u32[16][4] exit states | u32[16][4] exact?
// one step of both ends of the arc; returns the walker's output word for A's step (meaningful once the ends have met)
__device__ __forceinline__ uint32_t ws_step2(uint32_t& a, uint32_t& b, uint32_t& bits_acc, uint64_t info, uint32_t T) {
  const uint32_t d = (uint32_t)info, row = (uint32_t)(info >> 32);
  uint32_t ba = (a + d) >> 16, bb = (b + d) >> 16;
  const bool all = b < a && (a >> ba) == (b >> bb);   // past the cut and on the same row entry: every entry is reached
  a = all ? T : a; b = all ? 2u * T - 1u : b;
  ba = (a + d) >> 16; bb = (b + d) >> 16;
  const uint32_t old = a;
  a = *(const uint16_t PCO_LDS*)(uintptr_t)(row + ((a >> ba) << 1));
  b = *(const uint16_t PCO_LDS*)(uintptr_t)(row + ((b >> bb) << 1));
  bits_acc += ba;
  return (ba << 12) | __builtin_amdgcn_ubfe(old, 0u, ba);
}

#ifdef PCO_WS_TRACE
__device__ unsigned long long g_ws_trace[3 * 16384];   // per block: HW_ID, start, end (s_memrealtime: 100 MHz)
#endif
__global__ __launch_bounds__(128) void enc_walkseg_kernel(EncWorkspace ws, EncFast fx, uint32_t n_pages) {
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  uint8_t PCO_LDS* smem = enc_lds_base();
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  typedef uint64_t __attribute__((aligned(2))) u64_align2;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;
  // ---- the block's item ----
  const uint32_t item = blockIdx.x;
  if (item >= n_pages * ws.n_slots) return;
  const uint32_t p = item / ws.n_slots, sl = item % ws.n_slots;
  const uint32_t v = ws.slot_of_var[0] == sl ? 0u : (ws.slot_of_var[1] == sl ? 1u : 2u);
  EncPage PCO_GLOBAL* pg = (EncPage PCO_GLOBAL*)ws.pages + p;
  const uint32_t t = uni(pg->chunk);
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  if (!page_is_fast(ch, pg)) return;
  const PageVar pv = page_var(ch, v, (uint32_t)uni((uint64_t)pg->n));
  if (!ws_walks(fx.fused, pv)) return;
#ifdef PCO_WS_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 16384) { g_ws_trace[3 * blockIdx.x] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); g_ws_trace[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
  const bool finds = wd_takes(fx.fused, pv);   // its symbols come from the gathering waves (enc_dissect_kernel left it alone); else the walker stages what that kernel wrote
  const uint32_t info_off = ew_info_off(pv.asl), T = 1u << pv.asl;
#ifndef PCO_WS_SKEW
#define PCO_WS_SKEW 1
#endif
  const uint32_t nb = (pv.n_lat + kBatchN - 1) / kBatchN, nbs = (((nb + kWsSegs - 1) / kWsSegs) | 1u) + (PCO_WS_SKEW - 1), n_seg = (nb + nbs - 1) / nbs;   // batches, batches per segment (odd), segments in use
  const PlanRef plan = plan_ref(ws, t, v);
  if (wave == 0) {
    for (uint32_t i = lane; i < T; i += 64) ((uint16_t PCO_LDS*)smem)[i] = plan.next_states()[i];
    for (uint32_t b = lane; b < pv.n_bins; b += 64) {
      const uint32_t si = plan.syminfo()[b];   // cutoff(14) | min_renorm_bits(4) << 14 | (row + 8192)(14) << 18
      const uint32_t cutoff = si & 0x3fffu, minb = (si >> 14) & 15u, row = (si >> 18) - 8192u;
      ((uint64_t PCO_LDS*)(smem + info_off))[b] = (uint64_t)(((minb + 1u) << 16) - cutoff) | ((uint64_t)(lds0 + 2u * row) << 32);
    }
  }
  // the slack between the info words and the symbol buffers: offset bits u8[n_bins] | bins u8[range + 1] (as enc_walkd_kernel's slots)
  const uint32_t ob_off = info_off + 8u * pv.n_bins, vt_off = ob_off + ((pv.n_bins + 3u) & ~3u);
#ifdef PCO_WS_NOLDSLUT
  const bool all_lds = false;
#else
  const bool all_lds = finds && pv.n_bins <= 256 && pv.range < 4096 && vt_off + (uint32_t)pv.range + 1u <= kWsSymBase;
#endif
  if (all_lds && wave == 1) {
    for (uint32_t b = lane; b < pv.n_bins; b += 64) smem[ob_off + b] = (uint8_t)plan.bob()[b];
    const uint16_t PCO_GLOBAL* lut = vlut_ptr(ws, fx, t, v);
    const uint32_t base = (uint32_t)(pv.minv - pv.rel) + vlut_rot(t * ws.n_slots + ws.slot_of_var[v]);
    for (uint32_t i = lane; i <= (uint32_t)pv.range; i += 64) smem[vt_off + i] = (uint8_t)lut[(base + i) & (kDirectHistRange - 1)];
  }
  // a walker lane keeps the segment of its quad (lane >> 2), a gathering lane the segment lane & 15
  const uint32_t my_q = wave == 0 ? lane >> 2 : lane & (kWsSegs - 1);
  const uint32_t my_first = my_q * nbs;                                                              // the segment's first batch
  const uint32_t my_n_lat = my_q < n_seg && (wave == 0 || finds) ? (pv.n_lat - my_first * kBatchN < nbs * kBatchN ? pv.n_lat - my_first * kBatchN : nbs * kBatchN) : 0u;
  const uint32_t my_nb = (my_n_lat + kBatchN - 1) / kBatchN, max_nb = nbs;
  const uint64_t my_at = fast_at(pg, pv.skip) + (uint64_t)my_first * kBatchN, my_clat = uni((uint64_t)pg->start) + pv.skip + (uint64_t)my_first * kBatchN;
  const uint32_t my_m0 = (uint32_t)(pv.minv - pv.rel);
  wd_barrier();
  if (wave != 0) {
    // ================= the gathering wave: the batch of step `it` of all sixteen segments, four segments at a time =================
    // One wave per block (enc_walkd_kernel has four): what bounds the kernel on a full chip is how many WALKERS a CU holds -- their steps are
    // a chain of LDS round trips that the neighbours' traffic stretches from 145 to 400 cycles -- and a block of two waves fits twelve times
    // (its 12.8 KB of LDS), where five waves fitted five times.  The latents of a group of four segments are requested two groups ahead.
    const bool mine = my_n_lat != 0;
    const uint64_t my_clat_p = mine ? (uint64_t)(uintptr_t)(clat_ptr(ws, t, v) + my_clat) : (uint64_t)(uintptr_t)fx.vlut;
    uint8_t PCO_GLOBAL* my_gsym = mine ? fsym_ptr(ws, fx, t, v) + my_at : (uint8_t PCO_GLOBAL*)nullptr;
    uint32_t PCO_GLOBAL* my_gbat = (uint32_t PCO_GLOBAL*)fx.bat + (mine ? ((uint64_t)(p * 3 + v) * fx.bat_stride + my_first) * 2 : 0ull);
    const uint32_t lut_off = (t * ws.n_slots + ws.slot_of_var[v]) * kDirectHistRange;   // (u16 elements)
    const uint32_t rot2 = vlut_rot(lut_off / kDirectHistRange) * 0x10001u;
    const uint16_t PCO_GLOBAL* lut = (const uint16_t PCO_GLOBAL*)fx.vlut + lut_off;
    const uint32_t vt = lds0 + (all_lds ? vt_off : 0u), ot = lds0 + (all_lds ? ob_off : 0u), rg = all_lds ? (uint32_t)pv.range : 0u;
    auto bcast = [](uint32_t x, uint32_t q) { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)q); };
    auto bcast64 = [&](uint64_t x, uint32_t q) { return ((uint64_t)bcast((uint32_t)(x >> 32), q) << 32) | bcast((uint32_t)x, q); };
    auto batch_of = [&](uint32_t it) { return it < my_nb ? my_nb - 1 - it : 0u; };   // (no batch at this step: batch 0 is read, and never used)
    constexpr uint32_t kG = 4, kGroups = kWsSegs / kG;
    auto load_group = [&](uint32_t step, uint64_t (&w)[kG]) {   // step = it * kGroups + group; the 4 latents a lane owns of each of the group's batches
      const uint32_t it = step / kGroups, g = step % kGroups;
      const uint64_t my_src = my_clat_p + 2ull * batch_of(it) * kBatchN;
#pragma unroll
      for (uint32_t r = 0; r < kG; r++) w[r] = __builtin_nontemporal_load((const u64_align2 PCO_GLOBAL*)((const uint16_t PCO_GLOBAL*)(uintptr_t)bcast64(my_src, g * kG + r) + 4 * lane));
    };
    auto gather = [&](const uint64_t (&w)[kG], uint32_t (&e)[kG][4]) {
#pragma unroll
      for (uint32_t r = 0; r < kG; r++) {
        const uint32_t lo = (uint32_t)w[r] + rot2, hi = (uint32_t)(w[r] >> 32) + rot2;
        e[r][0] = lut[lo & (kDirectHistRange - 1)]; e[r][1] = lut[(lo >> 16) & (kDirectHistRange - 1)];
        e[r][2] = lut[hi & (kDirectHistRange - 1)]; e[r][3] = lut[(hi >> 16) & (kDirectHistRange - 1)];
      }
    };
    auto gather_lds = [&](const uint64_t (&w)[kG], uint32_t (&e)[kG][4]) {
#pragma unroll
      for (uint32_t r = 0; r < kG; r++) {
        const uint32_t lo = (uint32_t)w[r], hi = (uint32_t)(w[r] >> 32);
        uint32_t idx[4] = {(lo & 0xffffu) - my_m0, (lo >> 16) - my_m0, (hi & 0xffffu) - my_m0, (hi >> 16) - my_m0};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          idx[k] = idx[k] > rg ? 0u : idx[k];
          const uint32_t bin = *(const uint8_t PCO_LDS*)(uintptr_t)(vt + idx[k]);
          e[r][k] = bin | ((uint32_t)*(const uint8_t PCO_LDS*)(uintptr_t)(ot + bin) << 8);
        }
      }
    };
    const uint32_t n_steps = finds ? max_nb * kGroups : 0u;
    uint64_t w0[kG], w1[kG], w2[kG];   // the groups of steps s, s + 1, s + 2 (rotating)
    if (0 < n_steps) load_group(0, w0);
    if (1 < n_steps) load_group(1, w1);
    for (uint32_t it = 0; it <= max_nb; it++) {   // it == max_nb: nothing left to find, only the barrier
      if (it < max_nb && finds) {
        const bool my_on = it < my_nb;
        const uint32_t my_hb = batch_of(it), my_base = my_hb * kBatchN;
        const uint32_t my_cnt = my_on ? (my_n_lat - my_base < kBatchN ? my_n_lat - my_base : kBatchN) : 0u;
        const uint32_t my_buf = lds0 + kWsSymBase + my_q * 512 + (my_hb & 1) * 256;
        uint32_t my_total = 0;
#pragma unroll
        for (uint32_t g = 0; g < kGroups; g++) {
          const uint32_t step = it * kGroups + g;
          if (step + 2 < n_steps) load_group(step + 2, w2);
          uint32_t e[kG][4];
          if (all_lds) gather_lds(w0, e); else gather(w0, e);
#pragma unroll
          for (uint32_t r = 0; r < kG; r++) {
            const uint32_t q = g * kG + r, cnt = bcast(my_cnt, q);
            if (cnt < kBatchN) {   // (the page's last batch: a latent beyond it is bin 0 with no bits; no batch at this step: nothing is kept)
#pragma unroll
              for (int k = 0; k < 4; k++) e[r][k] = 4 * lane + k < cnt ? e[r][k] : 0u;
            }
            const uint32_t e01 = e[r][0] | (e[r][1] << 16), e23 = e[r][2] | (e[r][3] << 16);
            const uint32_t packed = __builtin_amdgcn_perm(e23, e01, 0x06040200u);               // the four bin bytes
            const uint32_t obs4 = __builtin_amdgcn_perm(e23, e01, 0x07050301u);                 // the four offset-bit counts (<= 64 each)
            const uint32_t total = wave_sum(__builtin_amdgcn_sad_u8(obs4, 0u, 0u));
            my_total = (lane & (kWsSegs - 1)) == q ? total : my_total;
            const uint32_t tr = quad_transpose_u8(packed, lane & 3);
            if (cnt != 0) *(uint32_t PCO_LDS*)(uintptr_t)(bcast(my_buf, q) + 4 * lane) = tr;
          }
#pragma unroll
          for (uint32_t r = 0; r < kG; r++) { w0[r] = w1[r]; w1[r] = w2[r]; }
        }
        // the symbols go on to enc_pack_kernel's scratch from the LDS buffers: a lane copies a quarter (64 bytes) of its own segment's batch,
        // whole 16-latent blocks as enc_dissect_kernel writes them; lanes 0..15 leave the batch's offset-bit total
        if (my_on) {
          constexpr uint32_t kParts = 64 / kWsSegs, kPer = 16 / kParts;   // lanes per segment, 16-byte blocks per lane
          const uint32_t part = lane / kWsSegs, blocks = (my_cnt + 15u) >> 4;
#pragma unroll
          for (uint32_t r = 0; r < kPer; r++) {
            const uint32_t blk = part * kPer + r;
            if (blk < blocks) *(u32x4_unaligned PCO_GLOBAL*)(my_gsym + my_base + 16 * blk) = *(const u32x4 PCO_LDS*)(uintptr_t)(my_buf + 16 * blk);
          }
          if (lane < kWsSegs) my_gbat[(uint64_t)my_hb * 2] = my_total;
        }
      }
      if (it == max_nb) __threadfence();   // (the walker reads the symbols back from the scratch for what it walks twice)
      wd_barrier();
    }
    return;
  }
  // ================= the walker wave: quad q = segment q =================
  __builtin_amdgcn_s_setprio(3);   // (its chain of dependent steps is the block's critical path; the gathering waves of the CU's other blocks share its SIMD)
  const uint32_t j = lane & 3;
  const uint32_t info_addr = lds0 + info_off, symbuf = lds0 + kWsSymBase + my_q * 512;
  uint16_t PCO_GLOBAL* gans = fansw_ptr(ws, fx, t, v) + my_at;
  uint32_t PCO_GLOBAL* gbat = (uint32_t PCO_GLOBAL*)fx.bat + ((uint64_t)(p * 3 + v) * fx.bat_stride + my_first) * 2;
  const uint8_t PCO_GLOBAL* gsym = (const uint8_t PCO_GLOBAL*)fsym_ptr(ws, fx, t, v) + my_at;
  uint32_t PCO_LDS* ex_state = (uint32_t PCO_LDS*)(smem + kWsExOff); uint32_t PCO_LDS* ex_exact = ex_state + kWsSegs * 4;
  const bool top = my_q + 1 == n_seg;                 // the page's last segment starts from the initial state (encoding.rs: table_size)
#ifdef PCO_WS_FORCEMET
  uint32_t st_a = T, st_b = T;   // (timing experiments: wrong bytes)
#else
  uint32_t st_a = T, st_b = top ? T : 2u * T - 1u;
#endif
  uint32_t fix_n = 0;                                  // leading batches (in walk order) that began before the ends had met
  uint32_t st_met = 0; bool met_seen = false;          // the state at the start of the first batch that began with every chain of the quad met
  const bool stages = my_n_lat != 0 && !finds;
  u32x4 pre[4];
  auto fetch_syms = [&](uint32_t b) {   // whole 16-latent blocks of batch b
    const uint32_t cnt = my_n_lat - b * kBatchN < kBatchN ? my_n_lat - b * kBatchN : kBatchN, blocks_bytes = (cnt + 15u) & ~15u;
#pragma unroll
    for (int k = 0; k < 4; k++) { pre[k] = u32x4{0, 0, 0, 0}; if (64 * j + 16 * k < blocks_bytes) pre[k] = *(const u32x4_unaligned PCO_GLOBAL*)(gsym + (uint64_t)b * kBatchN + 64 * j + 16 * k); }
  };
  // one batch of one chain from a known state (the unsegmented walkers' loops)
  auto walk_batch = [&](uint32_t b, uint32_t& state) {
    const uint32_t base = b * kBatchN, cnt = my_n_lat - base < kBatchN ? my_n_lat - base : kBatchN;
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
#ifndef PCO_WS_NOSTORE
        *(u64_align2 PCO_GLOBAL*)(ga + 16 * blk) = (uint64_t)(o0 | (o1 << 16)) | ((uint64_t)o23 << 32);
#else
        if (o0 == 0xdeadbeefu && o1 == 0x12345u && o23 == 77u) *(u64_align2 PCO_GLOBAL*)(ga + 16 * blk) = 1;   // (timing experiments: wrong bytes)
#endif
        __builtin_amdgcn_sched_barrier(0);
        i0 = n0; i1 = n1; i2 = n2; i3 = n3;
      }
    }
    bits_acc += quad_dpp<0xB1>(bits_acc); bits_acc += quad_dpp<0x4E>(bits_acc);
    if (j == 0) gbat[(uint64_t)b * 2 + 1] = bits_acc;
  };
  // the same with both ends of the arc (full batches only: a partial batch is the page's last, whose segment starts from one state)
  auto walk_batch2 = [&](uint32_t b) {
    uint16_t PCO_GLOBAL* ga = gans + (uint64_t)b * kBatchN + 4 * j;
    const uint32_t buf = symbuf + (b & 1) * 256;
    uint32_t bits_acc = 0;
    for (uint32_t blk = 16; blk-- > 0;) {
      const uint32_t sd = *(const uint32_t PCO_LDS*)(uintptr_t)(buf + 16 * blk + 4 * j);
      uint64_t out = 0;
#pragma unroll
      for (int k = 3; k >= 0; k--) {
        const uint64_t info = *(const uint64_t PCO_LDS*)(uintptr_t)(info_addr + 8u * ((sd >> (8 * k)) & 0xffu));
        out |= (uint64_t)ws_step2(st_a, st_b, bits_acc, info, T) << (16 * k);
      }
      *(u64_align2 PCO_GLOBAL*)(ga + 16 * blk) = out;
    }
    bits_acc += quad_dpp<0xB1>(bits_acc); bits_acc += quad_dpp<0x4E>(bits_acc);
    if (j == 0) gbat[(uint64_t)b * 2 + 1] = bits_acc;   // (written again, exactly, when the batch is walked a second time)
  };
  if (stages) fetch_syms(my_nb - 1);
  wd_barrier();   // (the gathering waves' it = 0)
  for (uint32_t it = 0; it < max_nb; it++) {
    if (stages && it < my_nb) {
      const uint32_t b = my_nb - 1 - it;
#pragma unroll
      for (int k = 0; k < 4; k++) *(u32x4 PCO_LDS*)(uintptr_t)(symbuf + (b & 1) * 256 + 64 * j + 16 * k) = pre[k];
      if (b > 0) fetch_syms(b - 1);
    }
    enc_wave_sync();
    const bool open = it < my_nb && st_a != st_b;     // this lane's ends have not met
    const bool any_open = __any(open);
    if (it < my_nb) {
      const uint32_t b = my_nb - 1 - it;
      // a batch that begins with any chain of the quad still open is walked again afterwards
      uint32_t qo = open ? 1u : 0u; qo |= quad_dpp<0xB1>(qo); qo |= quad_dpp<0x4E>(qo);
      fix_n += qo;
      if (qo == 0 && !met_seen) { st_met = st_a; met_seen = true; }
#ifndef PCO_WS_NOWALK
      if (any_open && my_n_lat - b * kBatchN >= kBatchN) walk_batch2(b);
      else { walk_batch(b, st_a); st_b = st_a; }
#endif
    }
    wd_barrier();
  }
  // ---- what was walked before the ends met, again, from the exit state of the segment above ----
  bool exact = my_n_lat == 0 || st_a == st_b;         // this lane's exit state is the true one
  ex_state[my_q * 4 + j] = st_a; ex_exact[my_q * 4 + j] = exact ? 1u : 0u;
  enc_wave_sync();
  bool pending = fix_n != 0;
#ifdef PCO_WS_NOFIX
  pending = false;   // (timing experiments: wrong bytes)
#endif
  for (uint32_t round = 0; round < kWsSegs; round++) {
    if (!__any(pending)) break;
    const uint32_t up = my_q + 1 < kWsSegs ? my_q + 1 : my_q;
    uint32_t up_ok = pending ? ex_exact[up * 4 + j] : 1u; up_ok &= quad_dpp<0xB1>(up_ok); up_ok &= quad_dpp<0x4E>(up_ok);
    const bool go = pending && up_ok != 0;
    uint32_t state = go ? ex_state[up * 4 + j] : T;
    uint32_t n_go = go ? fix_n : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(n_go, d, 64); n_go = n_go > o ? n_go : o; }
    n_go = uni(n_go);
    for (uint32_t it = 0; it < n_go; it++) {
      const bool on = go && it < fix_n;
      if (on) {
        const uint32_t b = my_nb - 1 - it;
        u32x4 s4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) s4[k] = *(const u32x4_unaligned PCO_GLOBAL*)(gsym + (uint64_t)b * kBatchN + 64 * j + 16 * k);   // (full batches: a segment that was ever open is not the page's last)
#pragma unroll
        for (int k = 0; k < 4; k++) *(u32x4 PCO_LDS*)(uintptr_t)(symbuf + (b & 1) * 256 + 64 * j + 16 * k) = s4[k];
      }
      enc_wave_sync();
      if (on) walk_batch(my_nb - 1 - it, state);
      enc_wave_sync();
    }
    enc_wave_sync();
    if (go) {
      if (fix_n >= my_nb) st_a = state;               // the whole segment was open: its exit state is known only now
      else if (state != st_met) ch->status = PCO_GFX_DEVICE_ERROR;   // (cannot happen: the second walk joins the first where its ends had met)
      ex_state[my_q * 4 + j] = st_a; ex_exact[my_q * 4 + j] = 1u;
      pending = false;
    }
    enc_wave_sync();
  }
  if (my_q == 0 && my_n_lat > 0) fx.fstate[((uint64_t)p * 3 + v) * 4 + j] = st_a;
#ifdef PCO_WS_TRACE
  if (lane == 0 && blockIdx.x < 16384) g_ws_trace[3 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace pcogfx
