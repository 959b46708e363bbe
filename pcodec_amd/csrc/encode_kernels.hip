// encode_kernels.hip -- gfx950 chunk encoder.
//
// Replaces, for the device path, the reference's encode stack
//   mode/{classic,int_mult,float_mult,float_quant}.rs split_latents            -> enc_split_kernel
//   delta/consecutive.rs:19-33 encode_in_place (+ moments)                     -> enc_split_kernel
//   histograms.rs + sort_utils.rs (exact equal-count quantile histogram)       -> enc_hist_kernel
//   bin_optimization.rs, ans/encoding.rs:95-175, ans/spec.rs, ans/encoding.rs:28-63,
//   wrapped/chunk_compressor.rs:38-99,502-541 (train_infos, should_fallback)   -> enc_train_kernel
//   compression_table.rs:51-74, chunk_latent_compressor.rs:96-132,163-329,
//   wrapped/chunk_compressor.rs:564-705, metadata/*.rs write_to,
//   standalone/compressor.rs:191-203                                            -> enc_page_kernel
//
// Data layout in HBM (per chunk, strides fixed per launch): latents of each latent variable
// (delta / primary / secondary) as dense arrays in a workspace, an u32 "dissect word" per latent
// (tANS value | tANS bit count | symbol), two sort buffers, and a small per-chunk plan (bins, tANS
// encoder tables).  Everything that is serial per chunk (the tANS chains, the bin DP, the bit cursor)
// runs in one wave per chunk with its tables in LDS; everything elementwise is a coalesced stream.
#include <type_traits>

#include "pco_dev.h"

namespace pcogfx {

// Speculative 16-bit latents (enc_split_kernel<c16>): a variable's stored keys are latent - ref, all below this bound.  The strict histogram's
// 16-bit replay (encode_hist_literal.hip) carries `pivot - ref` = key + 1 in a uint16_t and static_asserts on this constant.
constexpr uint32_t kC16KeyRange = 32768u;

#ifndef PCO_LDS
#define PCO_LDS __attribute__((address_space(3)))
#endif

constexpr uint32_t kMaxUnoptBinsLog = 8;            // device limit this round (compression_level <= 8 at n >= 2^12)
constexpr uint32_t kMaxBins = 1u << kMaxUnoptBinsLog;
constexpr uint32_t kMaxEncTableLog = 12;            // estimated_ans_size_log <= 12 (wrapped/chunk_compressor.rs:73-79)
constexpr uint32_t kDirectHistRange = 4096;         // value-space histogram when max-min < this
constexpr uint32_t kFastEncMaxAsl = 10;             // larger tANS tables take the single-kernel page path

struct EncVar {
  uint32_t present, latent_bits, n_lat, lat_start;
  unsigned long long minv, maxv;
  uint32_t n_hist, n_bins, ans_size_log, max_ob;
  uint32_t is_trivial, needs_ans, hist_path, walk_pending;   // walk_pending: the bin walk was left to enc_hist_walk_kernel
};
static_assert(sizeof(EncVar) == 64, "EncVar");
struct EncChunk {
  uint64_t n;
  uint32_t dtype, status;
  uint32_t mode_kind, mode_k;
  uint64_t mode_base;   // IntMult: base; FloatMult: ordered latent of base
  uint64_t mode_aux;    // FloatMult: bits of inv_base
  uint64_t mode_aux2;   // FloatMult: bits of base
  uint32_t delta_kind, delta_order, window_n_log, state_n_log;
  uint32_t fallback, unopt_bins_log;
  uint32_t n_pages, page_low, page_r, page_first;  // PagingSpec::EqualPagesUpTo layout: the first page_r pages hold page_low+1
  uint32_t fast_ok;         // 1: the chunk's pages go through the dissect / walk / scan / pack kernels (encode_fast.hip)
  uint32_t c16_ok;          // speculative 16-bit latents (enc_split_kernel): 0 = off (full-width latents), 1 = on and holding, 2 = a tile did not fit
  uint32_t exact_paging;    // PagingSpec::Exact: page sizes are arbitrary, positions map to pages through the page list (ws.pages[page_first ...])
  uint32_t big;             // more than 256 histogram bins (compression levels 9..12): the sort histogram, the block-wide bin DP and the page encoder with tables in HBM
  uint64_t c16_ref[2];      // what the 16-bit latents of variables 1 / 2 are relative to
  EncVar v[3];
};
static_assert(sizeof(EncChunk) % 8 == 0, "EncChunk");

// One page = one independent delta + tANS stream (wrapped/chunk_compressor.rs:142-217,659-705).
struct EncPage {
  uint32_t chunk, page_idx;
  uint64_t start, n;        // numbers [start, start+n) of the chunk
  void* dst; uint64_t dst_cap;
  uint32_t flags, pad;      // kPageFlagPreamble: standalone preamble + ChunkMeta first; kPageFlagMetaOnly: ChunkMeta only
  uint64_t moments[8];      // delta state of the page (consecutive moments)
};
constexpr uint32_t kPageFlagPreamble = 1u, kPageFlagMetaOnly = 2u;

__device__ __forceinline__ uint64_t page_start_of(uint64_t i, uint32_t low, uint32_t r) {
  const uint64_t boundary = (uint64_t)r * (low + 1);
  if (i < boundary) return (i / (low + 1)) * (low + 1);
  return boundary + ((i - boundary) / low) * low;
}

// per (chunk, var) plan region: histogram bins, optimized bins, tANS encoder tables.  Its bin capacity is a property of the call:
// 256 (compression levels whose unoptimized_bins_log stays at 8 or below: every level up to 8 at n >= 2^12) or 4096 (levels 9..12:
// wrapped/chunk_compressor.rs:362-371, constants.rs:35).  Layout (cap = capacity): u64 hlower[cap] hupper[cap] blower[cap] |
// u32 hcount[cap] bweight[cap] bcount[cap] syminfo[cap] | u16 next_states[4096] | u8 bob[cap].
constexpr uint32_t kBigBins = 1u << 12;
__host__ __device__ constexpr uint64_t plan_bytes_for(uint32_t cap) { return ((uint64_t)cap * (24 + 16 + 1) + (2ull << kMaxEncTableLog) + 63) & ~63ull; }
struct PlanRef {
  uint8_t PCO_GLOBAL* base; uint32_t cap;
  __device__ __forceinline__ uint64_t PCO_GLOBAL* hlower() const { return (uint64_t PCO_GLOBAL*)base; }
  __device__ __forceinline__ uint64_t PCO_GLOBAL* hupper() const { return (uint64_t PCO_GLOBAL*)base + cap; }
  __device__ __forceinline__ uint64_t PCO_GLOBAL* blower() const { return (uint64_t PCO_GLOBAL*)base + 2 * (uint64_t)cap; }
  __device__ __forceinline__ uint32_t PCO_GLOBAL* hcount() const { return (uint32_t PCO_GLOBAL*)(base + 24ull * cap); }
  __device__ __forceinline__ uint32_t PCO_GLOBAL* bweight() const { return (uint32_t PCO_GLOBAL*)(base + 28ull * cap); }
  __device__ __forceinline__ uint32_t PCO_GLOBAL* bcount() const { return (uint32_t PCO_GLOBAL*)(base + 32ull * cap); }
  __device__ __forceinline__ uint32_t PCO_GLOBAL* syminfo() const { return (uint32_t PCO_GLOBAL*)(base + 36ull * cap); }   // cutoff(14) | min_renorm_bits(4)<<14 | (ns_off - weight + 8192)(14)<<18
  __device__ __forceinline__ uint16_t PCO_GLOBAL* next_states() const { return (uint16_t PCO_GLOBAL*)(base + 40ull * cap); }
  __device__ __forceinline__ uint8_t PCO_GLOBAL* bob() const { return (uint8_t PCO_GLOBAL*)(base + 40ull * cap + (2ull << kMaxEncTableLog)); }
};
struct EncWorkspace {
  EncChunk* chunks;
  uint8_t* plans;           // [task][3] plan regions of plan_bytes_for(plan_cap) bytes each (PlanRef)
  uint32_t plan_cap, pad_cap;   // bin capacity of every plan region of this call: 256 or 4096
  EncPage* pages;           // [n_pages_total]
  uint8_t* lat;             // [lat slot][n_slots][n_stride] elements of lat_esz bytes: slots 0 .. lat_cap1 - 1 (see lat_slot)
  uint8_t* lat2;            // ... and slots lat_cap1 .. (the chunks whose 16-bit speculation failed in the split, allocated once their number is known)
  uint32_t* lat_slot;       // [task] the chunk's slot in lat / lat2, handed out on the device (enc_lat_slots_kernel) to the chunks that write full-width
                            // latents; kNoLatSlot for the others (16-bit latents only: most chunks of most calls -- they cost no full-width scratch)
  uint32_t* lat_used;       // device counter: slots handed out
  uint32_t lat_cap1;
  uint32_t lat_esz;         // bytes per element of lat / sort: 8 when the call has a 64-bit chunk, else 4 (the lookback variable's latents are u32 whatever the type)
  uint8_t* sort;            // [task][2][n_stride] elements of lat_esz bytes
  uint8_t* walk;            // [task][3] kWalkRecBytes: rank records + tables of a deferred bin walk (enc_hist_walk_kernel), or null
  uint32_t* dissect;        // [task][n_slots][n_stride]
  uint64_t n_stride;
  uint32_t n_slots;         // latent slots allocated per task
  uint32_t slot_of_var[3];  // slot index per var (0xffffffff = not allocated)
  uint32_t* need_sort;      // device flag: some variable's value range is >= kWideHistRange (enc_hist_sort_kernel and the sort buffers are needed)
  uint32_t* need_full0;     // device counter: chunks that enc_presample_kernel took out of the 16-bit speculation
};

__device__ __forceinline__ uint8_t PCO_LDS* enc_lds_base() {
  extern __shared__ __attribute__((aligned(16))) uint8_t pco_lds[];
  return (uint8_t PCO_LDS*)pco_lds;
}
__device__ __forceinline__ void enc_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ PlanRef plan_ref(const EncWorkspace& ws, uint32_t task, uint32_t var) {
  return PlanRef{(uint8_t PCO_GLOBAL*)ws.plans + ((uint64_t)task * 3 + var) * plan_bytes_for(ws.plan_cap), ws.plan_cap};
}
constexpr uint32_t kNoLatSlot = 0xffffffffu;
// (a chunk without a slot gets slot 0's address: every dereference is behind the chunk's c16_ok, so nothing reads or writes it, and a
//  pointer that is only computed must not fault)
template <class L> __device__ __forceinline__ L PCO_GLOBAL* lat_ptr(const EncWorkspace& ws, uint32_t task, uint32_t var) {
  uint32_t s = ws.lat_slot[task];
  if (s == kNoLatSlot) s = 0;
  const uint64_t per_var = ws.n_stride * ws.lat_esz, per_slot = (uint64_t)ws.n_slots * per_var;
  uint8_t* base = s < ws.lat_cap1 ? ws.lat + (uint64_t)s * per_slot : ws.lat2 + (uint64_t)(s - ws.lat_cap1) * per_slot;
  return (L PCO_GLOBAL*)(base + (uint64_t)ws.slot_of_var[var] * per_var);
}
__device__ __forceinline__ uint32_t PCO_GLOBAL* dissect_ptr(const EncWorkspace& ws, uint32_t task, uint32_t var) {
  return (uint32_t PCO_GLOBAL*)(ws.dissect + ((uint64_t)task * ws.n_slots + ws.slot_of_var[var]) * ws.n_stride);
}
// Compact latents (x - min as u16) of the variables whose histogram was built by LDS counting (value range below
// kWideHistRange): written by the histogram kernels, read by the dissect / pack kernels instead of the 8-byte latents.
// They live in the (task, slot) region of the dissect buffer, which only enc_page_kernel uses otherwise.
__device__ __forceinline__ uint16_t PCO_GLOBAL* clat_ptr(const EncWorkspace& ws, uint32_t task, uint32_t var) {
  return (uint16_t PCO_GLOBAL*)dissect_ptr(ws, task, var);
}
template <class L> __device__ __forceinline__ L PCO_GLOBAL* sort_ptr(const EncWorkspace& ws, uint32_t task, uint32_t which) {
  return (L PCO_GLOBAL*)(ws.sort + ((uint64_t)task * 2 + which) * ws.n_stride * ws.lat_esz);
}

// wrapped/chunk_compressor.rs:362-371
__host__ __device__ inline uint32_t choose_unoptimized_bins_log(uint32_t level, uint64_t n) {
  const uint32_t log_n = 63u - (uint32_t)__builtin_clzll(n);
  const uint32_t fast = log_n >= 4 ? log_n - 4 : 0;
  if (level <= fast) return level;
  return fast + (level - fast) / 2;
}

// =========================================================================================================
// K0: initialise per-chunk state
// =========================================================================================================
struct EncModePlan {  // host-resolved mode / delta (explicit specs; Auto is resolved by the host driver)
  uint32_t mode_kind, mode_k; uint64_t mode_base, mode_aux, mode_aux2;
  uint32_t delta_kind, delta_order, window_n_log, state_n_log;
  uint32_t n_pages, page_low, page_r, page_first;
  uint32_t ubl_override, exact_paging;   // ubl_override: 0xffffffff = derive from (level, n), trials use the full chunk's value; exact_paging: PagingSpec::Exact
};

__global__ void enc_init_kernel(EncWorkspace ws, const PcoGfxEncodeTask* tasks, const EncModePlan* plans, uint32_t n_tasks, uint32_t level, uint32_t c16_enable) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tasks) return;
  const PcoGfxEncodeTask task = tasks[t];
  const EncModePlan mp = plans[t];
  EncChunk c{};
  c.n = task.n; c.dtype = task.dtype; c.status = PCO_GFX_OK;
  c.mode_kind = mp.mode_kind; c.mode_k = mp.mode_k; c.mode_base = mp.mode_base; c.mode_aux = mp.mode_aux; c.mode_aux2 = mp.mode_aux2;
  c.delta_kind = mp.delta_kind; c.delta_order = mp.delta_order; c.window_n_log = mp.window_n_log; c.state_n_log = mp.state_n_log;
  c.unopt_bins_log = mp.ubl_override != 0xffffffffu ? mp.ubl_override : choose_unoptimized_bins_log(level, task.n);
  const uint32_t lbits = (uint32_t)dtype_bits(task.dtype);
  const uint32_t nlps = mp.delta_kind == kDeltaConsecutive ? mp.delta_order : (mp.delta_kind == kDeltaLookback ? (1u << mp.state_n_log) : 0u);
  const uint64_t n = task.n;
  c.n_pages = mp.n_pages; c.page_low = mp.page_low; c.page_r = mp.page_r; c.page_first = mp.page_first;
  // stored latents of a delta'd variable: every page drops its first nlps (wrapped/chunk_compressor.rs:185-191)
  uint64_t stored = 0;
  for (uint32_t p = 0; p < mp.n_pages; p++) { const uint64_t pn = mp.exact_paging ? ws.pages[mp.page_first + p].n : (uint64_t)(mp.page_low + (p < mp.page_r ? 1u : 0u)); stored += pn > nlps ? pn - nlps : 0; }
  c.exact_paging = mp.exact_paging;
  for (int v = 0; v < 3; v++) { c.v[v].minv = ~0ull; c.v[v].maxv = 0; }
  c.v[0].present = mp.delta_kind == kDeltaLookback; c.v[0].latent_bits = 32;
  c.v[0].lat_start = nlps; c.v[0].n_lat = (uint32_t)stored;  // lookbacks live at the primary's index (page prefix unused)
  c.v[1].present = 1; c.v[1].latent_bits = lbits; c.v[1].lat_start = nlps; c.v[1].n_lat = (uint32_t)stored;
  c.v[2].present = mp.mode_kind == kIntMult || mp.mode_kind == kFloatMult || mp.mode_kind == kFloatQuant;
  c.v[2].latent_bits = lbits; c.v[2].lat_start = 0; c.v[2].n_lat = (uint32_t)n;
  c.big = c.unopt_bins_log > kMaxUnoptBinsLog ? 1u : 0u;
  if (c.unopt_bins_log > 12 || (c.big && ws.plan_cap < kBigBins)) c.status = PCO_GFX_INVALID_ARGUMENT;   // (cannot happen: levels stop at 12)
  c.c16_ok = c16_enable && mp.delta_kind != kDeltaLookback && !c.big ? 1u : 0u;   // (lookback reads the full-width primary back; big chunks are histogrammed by the sort path)
  ws.chunks[t] = c;
}

// gather: out[k] = src[idx[k]]  (Auto sampling: sampling.rs:27-100)
struct GatherTask { const void* src; void* dst; const uint32_t* idx; uint32_t n_idx, elem_bytes; };
__global__ __launch_bounds__(256) void gather_kernel(const GatherTask* tasks, uint32_t blocks_per_task) {   // (linear grid: gridDim.y stops at 65535)
  const uint32_t t = blockIdx.x / blocks_per_task;
  const GatherTask g = tasks[t];
  const uint32_t k = (blockIdx.x - t * blocks_per_task) * 256 + threadIdx.x;
  if (k >= g.n_idx) return;
  const uint32_t i = g.idx[k];
  if (g.elem_bytes == 8) ((uint64_t*)g.dst)[k] = ((const uint64_t*)g.src)[i];
  else if (g.elem_bytes == 4) ((uint32_t*)g.dst)[k] = ((const uint32_t*)g.src)[i];
  else if (g.elem_bytes == 2) ((uint16_t*)g.dst)[k] = ((const uint16_t*)g.src)[i];
  else ((uint8_t*)g.dst)[k] = ((const uint8_t*)g.src)[i];
}

// compact per-task record of a trained plan, for the host-side size estimate of Auto delta trials.  The average bits per latent
// (metadata/chunk_latent_var.rs avg_bits_per_latent: a sum over the bins, in bin order, of f64 terms (ans_size_log - log2(weight) +
// offset_bits) * weight / 2^ans_size_log) is formed here, by one lane per variable in the reference's order, with log2(weight) taken
// from a table the HOST filled with its libm (weights are at most 4096): IEEE add / multiply / divide round the same on both
// sides, so the host gets the reference's f64 bit for bit without 6 bytes per bin crossing PCIe.
struct TrialSummary {
  uint32_t status, fallback;
  uint32_t asl[2], n_bins[2], n_lat[2];
  double avg[2];
};
__global__ __launch_bounds__(64) void enc_trial_summary_kernel(EncWorkspace ws, TrialSummary* out, uint32_t n_tasks, const double* log2_of_weight) {
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  if (t >= n_tasks) return;
  const EncChunk* ch = ws.chunks + t;
  TrialSummary o;
  o.status = ch->status; o.fallback = ch->fallback;
  for (int v = 0; v < 2; v++) {
    const PlanRef plan = plan_ref(ws, t, v);
    const uint32_t nb = ch->v[v].present ? ch->v[v].n_bins : 0, asl = ch->v[v].ans_size_log;
    o.asl[v] = asl; o.n_bins[v] = nb; o.n_lat[v] = ch->v[v].present ? ch->v[v].n_lat : 0;
    double avg = 0.0; const double tw = (double)(1ull << asl);
    for (uint32_t b = 0; b < nb; b++) {
      const uint32_t wi = plan.bweight()[b];
      const double w = (double)wi;
      avg += ((double)asl - log2_of_weight[wi] + (double)plan.bob()[b]) * w / tw;
    }
    o.avg[v] = avg;
  }
  out[t] = o;
}

// =========================================================================================================
// K1: mode split + consecutive delta + min/max (elementwise, coalesced)
// =========================================================================================================
template <class L, int MODE>
__device__ __forceinline__ void split_one(uint32_t num_kind, L mode_base, uint32_t mode_k, uint64_t aux_inv, uint64_t aux_base, L bits, L& p, L& s) {
  s = 0;
  if constexpr (MODE == kClassic) p = to_latent_ordered<L>(bits, num_kind);
  else if constexpr (MODE == kIntMult) { const L u = to_latent_ordered<L>(bits, num_kind); p = (L)(u / mode_base); s = (L)(u % mode_base); }
  else if constexpr (MODE == kFloatQuant) {
    const L num_ = to_latent_ordered<L>(bits, kFloat);
    const L lowmax = (L)(((L)1 << mode_k) - 1);
    p = (L)(num_ >> mode_k);
    const L low = (L)(num_ & lowmax);
    s = (bits & lmid<L>()) ? (L)(lowmax - low) : low;
  } else {  // kFloatMult (mode/float_mult.rs:38-60)
    if constexpr (sizeof(L) >= 4) {
      typedef typename FloatOf<L>::F F;
      const F num = bits_to_float(bits);
      const F inv_base = bits_to_float((L)aux_inv), base = bits_to_float((L)aux_base);
      const F mult = round_half_away(num * inv_base);
      p = int_float_to_latent<L>(mult);
      s = (L)(to_latent_ordered<L>(bits, kFloat) - to_latent_ordered<L>(float_to_bits(mult * base), kFloat) + lmid<L>());
    } else if constexpr (sizeof(L) == 2) {   // f16: every product rounded to f16 (pco_dev.h)
      const uint32_t mult = half_round(half_mul((uint32_t)bits, (uint32_t)aux_inv));
      p = (L)half_int_float_to_latent(mult);
      s = (L)(to_latent_ordered<L>(bits, kFloat) - to_latent_ordered<L>((L)half_mul(mult, (uint32_t)aux_base), kFloat) + lmid<L>());
    } else p = 0;
  }
}

// Primary latents at sampled positions only (the Auto-delta sample, sampling.rs:27-60: 2.5 % of a chunk): what a full split pass
// would have produced there, without touching the other 97.5 %.  grid = (ceil(max n_idx / 256), tasks)
struct SplitGatherTask { const void* src; void* dst; const uint32_t* idx; uint32_t n_idx, dtype, mode_kind, mode_k; uint64_t mode_base, mode_aux, mode_aux2; };
template <class L> __device__ __forceinline__ void split_gather_one(const SplitGatherTask& g, uint32_t k, uint32_t num_kind) {
  const L bits = ((const L PCO_GLOBAL*)g.src)[g.idx[k]];
  L p = 0, s = 0;
  switch (g.mode_kind) {
    case kIntMult: split_one<L, kIntMult>(num_kind, (L)g.mode_base, g.mode_k, g.mode_aux, g.mode_aux2, bits, p, s); break;
    case kFloatMult: split_one<L, kFloatMult>(num_kind, (L)g.mode_base, g.mode_k, g.mode_aux, g.mode_aux2, bits, p, s); break;
    case kFloatQuant: split_one<L, kFloatQuant>(num_kind, (L)g.mode_base, g.mode_k, g.mode_aux, g.mode_aux2, bits, p, s); break;
    default: split_one<L, kClassic>(num_kind, (L)g.mode_base, g.mode_k, g.mode_aux, g.mode_aux2, bits, p, s); break;
  }
  ((L PCO_GLOBAL*)g.dst)[k] = p;
}
__global__ __launch_bounds__(256) void split_gather_kernel(const SplitGatherTask* tasks, uint32_t blocks_per_task) {
  const uint32_t t = blockIdx.x / blocks_per_task;
  const SplitGatherTask g = tasks[t];
  const uint32_t k = (blockIdx.x - t * blocks_per_task) * 256 + threadIdx.x;
  if (k >= g.n_idx) return;
  const uint32_t num_kind = dtype_kind(g.dtype); const int bits = dtype_bits(g.dtype);
  if (bits == 64) split_gather_one<uint64_t>(g, k, num_kind);
  else if (bits == 32) split_gather_one<uint32_t>(g, k, num_kind);
  else if (bits == 16) split_gather_one<uint16_t>(g, k, num_kind);
  else split_gather_one<uint8_t>(g, k, num_kind);
}

// One thread owns kSplitE contiguous numbers (all of its loads are issued before anything is used), a block owns a
// tile of kSplitTile.  The `order` preceding primaries a thread needs for the finite differences come from its left
// neighbour through LDS; the block's halo (the 7 numbers before the tile) is split by threads 249..255.
constexpr uint32_t kSplitE = 8, kSplitTile = 256 * kSplitE;
constexpr uint32_t kSplitLdsPrev = 0;                       // L[257][7]: slot 0 = halo, slot t + 1 = thread t
constexpr uint32_t kSplitLdsRed = 257 * 7 * 8;              // u64[4 waves][4]
constexpr uint32_t kSplitLdsBytes = kSplitLdsRed + 128;

template <class L, int MODE, bool kSpec>
__device__ __forceinline__ void enc_split_body(const EncWorkspace& ws, const PcoGfxEncodeTask& task, uint32_t t, EncPage PCO_GLOBAL* pg, uint32_t tile) {
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  const uint64_t n = uni((uint64_t)pg->n), pstart = uni((uint64_t)pg->start);
  const uint32_t num_kind = dtype_kind(uni(task.dtype));
  const uint32_t mode_k = uni(ch->mode_k);
  const L mode_base = (L)uni((uint64_t)ch->mode_base); const uint64_t aux_inv = uni((uint64_t)ch->mode_aux), aux_base = uni((uint64_t)ch->mode_aux2);
  const uint32_t delta_kind = uni(ch->delta_kind);
  const uint32_t order = delta_kind == kDeltaConsecutive ? uni(ch->delta_order) : 0;
  constexpr bool has_sec = MODE != kClassic;
  const L PCO_GLOBAL* src = (const L PCO_GLOBAL*)task.src + pstart;
  const bool lookback = delta_kind == kDeltaLookback;
  // lookback: stage the un-delta'd primary in sort buffer A; enc_lookback_kernel turns it into lat[0] / lat[1]
  L PCO_GLOBAL* lat1 = kSpec ? nullptr : (lookback ? sort_ptr<L>(ws, t, 0) : lat_ptr<L>(ws, t, 1)) + pstart;
  L PCO_GLOBAL* lat2 = has_sec && !kSpec ? lat_ptr<L>(ws, t, 2) + pstart : nullptr;
  uint8_t PCO_LDS* smem = enc_lds_base();
  const uint32_t tid = threadIdx.x;
  const uint64_t tile0 = (uint64_t)tile * kSplitTile;
  const uint64_t e0 = tile0 + (uint64_t)tid * kSplitE;
  L raw[kSplitE], a[7 + kSplitE], sec[kSplitE];
  const bool full = e0 + kSplitE <= n;
  if (full) {
#pragma unroll
    for (uint32_t k = 0; k < kSplitE; k++) raw[k] = src[e0 + k];
  } else {
#pragma unroll
    for (uint32_t k = 0; k < kSplitE; k++) raw[k] = e0 + k < n ? src[e0 + k] : (L)0;
  }
  L halo_raw = 0;
  const bool halo_on = order > 0 && tid >= 249 && tile0 + tid >= 256 && tile0 + tid - 256 < n;   // element tile0 - 7 + (tid - 249)
  if (halo_on) halo_raw = src[tile0 + tid - 256];
#pragma unroll
  for (uint32_t k = 0; k < kSplitE; k++) split_one<L, MODE>(num_kind, mode_base, mode_k, aux_inv, aux_base, raw[k], a[7 + k], sec[k]);
#pragma unroll
  for (uint32_t jj = 0; jj < 7; jj++) a[jj] = 0;
  if (order > 0) {  // (wrapping) k-th finite difference by repeated adjacent differences (delta/consecutive.rs:3-33)
    L PCO_LDS* prevbuf = (L PCO_LDS*)(smem + kSplitLdsPrev);
#pragma unroll
    for (uint32_t jj = 0; jj < 7; jj++) if (jj + order >= 7) prevbuf[(tid + 1) * 7 + jj] = a[8 + jj];
    if (tid >= 249) { L hp = 0, hs; if (halo_on) split_one<L, MODE>(num_kind, mode_base, mode_k, aux_inv, aux_base, halo_raw, hp, hs); prevbuf[tid - 249] = hp; }
    __syncthreads();
#pragma unroll
    for (uint32_t jj = 0; jj < 7; jj++) if (jj + order >= 7) a[jj] = prevbuf[tid * 7 + jj];
    for (uint32_t r = 0; r < order; r++) {
#pragma unroll
      for (uint32_t idx = 6 + kSplitE; idx >= 1; idx--) a[idx] = (L)(a[idx] - a[idx - 1]);
    }
  }
  L mn1 = (L)~(L)0, mx1 = 0, mn2 = (L)~(L)0, mx2 = 0;
  L d[kSplitE];
  const L toggle = order > 0 ? lmid<L>() : (L)0;
#pragma unroll
  for (uint32_t k = 0; k < kSplitE; k++) d[k] = (L)(a[7 + k] + toggle);
  if (kSpec) {
  } else if (full && e0 >= order) {
#pragma unroll
    for (uint32_t k = 0; k < kSplitE; k++) {
      mn1 = d[k] < mn1 ? d[k] : mn1; mx1 = d[k] > mx1 ? d[k] : mx1;
      if (has_sec) { mn2 = sec[k] < mn2 ? sec[k] : mn2; mx2 = sec[k] > mx2 ? sec[k] : mx2; }
    }
  } else {
#pragma unroll
    for (uint32_t k = 0; k < kSplitE; k++) {
      const uint64_t i = e0 + k;
      if (i < n) {
        if (i >= order) { mn1 = d[k] < mn1 ? d[k] : mn1; mx1 = d[k] > mx1 ? d[k] : mx1; }   // positions < order are junk (not stored)
        if (has_sec) { mn2 = sec[k] < mn2 ? sec[k] : mn2; mx2 = sec[k] > mx2 ? sec[k] : mx2; }
      }
    }
  }
  if (lookback) { mn1 = (L)~(L)0; mx1 = 0; }   // enc_lookback_kernel owns the primary's range
  if (kSpec) {
    // Speculative 16-bit latents: most chunks' latents sit within 2^14 of a value known up front (the centre of a delta'd
    // variable, else the chunk's first latent), and then 2 bytes per latent instead of 8 leave this kernel and enter the
    // histogram.  A tile that does not fit flags the chunk; enc_split_kernel<false> then rewrites the chunk at full width.
    // The reference is kept 2^14 away from both ends of the latent type, so that "d - (ref - 2^14) < 2^15" in the type's
    // modular arithmetic is a statement about numeric distance (the histogram path is chosen from the numeric range);
    // 8-bit latents always fit and are stored as they are.
    constexpr L kBias = sizeof(L) == 1 ? (L)0 : (L)16384, kLmax = (L)~(L)0;
    auto clamp_ref = [&](L x) { return sizeof(L) == 1 ? (L)0 : (x < kBias ? kBias : (x > (L)(kLmax - kBias) ? (L)(kLmax - kBias) : x)); };
    L ref1 = sizeof(L) == 1 ? (L)0 : toggle, ref2 = 0;
    if (order == 0 || has_sec) { L p0, s0; split_one<L, MODE>(num_kind, mode_base, mode_k, aux_inv, aux_base, ((const L PCO_GLOBAL*)task.src)[0], p0, s0); if (order == 0) ref1 = clamp_ref(p0); ref2 = clamp_ref(s0); }
    const L lo1 = (L)(ref1 - kBias), lo2 = (L)(ref2 - kBias);   // what the 16-bit latents are relative to: they lie in [0, 2^15)
    if (pstart == 0 && e0 == 0) { ch->c16_ref[0] = (uint64_t)lo1; ch->c16_ref[1] = (uint64_t)lo2; }
    uint16_t PCO_GLOBAL* c1 = clat_ptr(ws, t, 1) + pstart;
    uint16_t PCO_GLOBAL* c2 = has_sec ? clat_ptr(ws, t, 2) + pstart : nullptr;
    uint32_t bad = 0, rmin1 = 0xffffffffu, rmax1 = 0, rmin2 = 0xffffffffu, rmax2 = 0;
    uint16_t w1[kSplitE], w2[kSplitE];
#pragma unroll
    for (uint32_t k = 0; k < kSplitE; k++) {
      const uint64_t i = e0 + k;
      const L t1 = (L)(d[k] - lo1);
      const uint32_t u1 = (uint32_t)t1;
      w1[k] = (uint16_t)u1;
      if (i < n && i >= order) { if ((uint64_t)t1 >= kC16KeyRange) bad = 1; rmin1 = u1 < rmin1 ? u1 : rmin1; rmax1 = u1 > rmax1 ? u1 : rmax1; }
      if (has_sec) {
        const L t2 = (L)(sec[k] - lo2);
        const uint32_t u2 = (uint32_t)t2;
        w2[k] = (uint16_t)u2;
        if (i < n) { if ((uint64_t)t2 >= kC16KeyRange) bad = 1; rmin2 = u2 < rmin2 ? u2 : rmin2; rmax2 = u2 > rmax2 ? u2 : rmax2; }
      }
    }
    if (full) {   // one 16-byte store per variable (2-byte aligned: a page may start at an odd index)
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      typedef u32x4 __attribute__((aligned(2))) u32x4_a2;
      static_assert(kSplitE == 8, "eight 16-bit latents per thread");
      u32x4 q1; q1.x = w1[0] | ((uint32_t)w1[1] << 16); q1.y = w1[2] | ((uint32_t)w1[3] << 16); q1.z = w1[4] | ((uint32_t)w1[5] << 16); q1.w = w1[6] | ((uint32_t)w1[7] << 16);
      *(u32x4_a2 PCO_GLOBAL*)(c1 + e0) = q1;
      if (has_sec) {
        u32x4 q2; q2.x = w2[0] | ((uint32_t)w2[1] << 16); q2.y = w2[2] | ((uint32_t)w2[3] << 16); q2.z = w2[4] | ((uint32_t)w2[5] << 16); q2.w = w2[6] | ((uint32_t)w2[7] << 16);
        *(u32x4_a2 PCO_GLOBAL*)(c2 + e0) = q2;
      }
    } else {
#pragma unroll
      for (uint32_t k = 0; k < kSplitE; k++) if (e0 + k < n) { c1[e0 + k] = w1[k]; if (has_sec) c2[e0 + k] = w2[k]; }
    }
    // range of the tile, on the 32-bit relative values (a tile that does not fit contributes nothing: its chunk is redone)
    {
      auto umin = [](uint32_t p, uint32_t q) { return p < q ? p : q; }; auto umax = [](uint32_t p, uint32_t q) { return p > q ? p : q; };
      rmin1 = wave_reduce_u32(rmin1, umin); rmax1 = wave_reduce_u32(rmax1, umax);
      if (has_sec) { rmin2 = wave_reduce_u32(rmin2, umin); rmax2 = wave_reduce_u32(rmax2, umax); }
    }
    const uint32_t wbad = __any(bad != 0) ? 1u : 0u;
    uint32_t PCO_LDS* red32 = (uint32_t PCO_LDS*)(smem + kSplitLdsRed);
    if (lane_id() == 0) { const uint32_t w = tid >> 6; red32[w * 5 + 0] = rmin1; red32[w * 5 + 1] = rmax1; red32[w * 5 + 2] = rmin2; red32[w * 5 + 3] = rmax2; red32[w * 5 + 4] = wbad; }
    __syncthreads();
    if (tid == 0) {
      uint32_t m1 = 0xffffffffu, x1 = 0, m2 = 0xffffffffu, x2 = 0, anybad = 0;
      for (uint32_t w = 0; w < 4; w++) {
        m1 = red32[w * 5] < m1 ? red32[w * 5] : m1; x1 = red32[w * 5 + 1] > x1 ? red32[w * 5 + 1] : x1;
        m2 = red32[w * 5 + 2] < m2 ? red32[w * 5 + 2] : m2; x2 = red32[w * 5 + 3] > x2 ? red32[w * 5 + 3] : x2;
        anybad |= red32[w * 5 + 4];
      }
      if (anybad) atomicCAS(&ws.chunks[t].c16_ok, 1u, 2u);
      else {
        if (m1 <= x1) { atomicMin((unsigned long long*)&ws.chunks[t].v[1].minv, (unsigned long long)(L)(lo1 + (L)m1)); atomicMax((unsigned long long*)&ws.chunks[t].v[1].maxv, (unsigned long long)(L)(lo1 + (L)x1)); }
        if (has_sec && m2 <= x2) { atomicMin((unsigned long long*)&ws.chunks[t].v[2].minv, (unsigned long long)(L)(lo2 + (L)m2)); atomicMax((unsigned long long*)&ws.chunks[t].v[2].maxv, (unsigned long long)(L)(lo2 + (L)x2)); }
      }
    }
  } else if (full) {
#pragma unroll
    for (uint32_t k = 0; k < kSplitE; k++) lat1[e0 + k] = d[k];
    if (has_sec) {
#pragma unroll
      for (uint32_t k = 0; k < kSplitE; k++) lat2[e0 + k] = sec[k];
    }
  } else {
#pragma unroll
    for (uint32_t k = 0; k < kSplitE; k++) if (e0 + k < n) { lat1[e0 + k] = d[k]; if (has_sec) lat2[e0 + k] = sec[k]; }
  }
  if (e0 == 0 && order > 0) {
    // moments[o] = (delta^o p)[o]
    L q[8];
    for (uint32_t jj = 0; jj < 8; jj++) {
      if (jj < order && jj < n) { L pp, ss; split_one<L, MODE>(num_kind, mode_base, mode_k, aux_inv, aux_base, src[jj], pp, ss); q[jj] = pp; } else q[jj] = 0;
    }
    for (uint32_t oo = 0; oo < order; oo++) {
      pg->moments[oo] = oo < n ? (uint64_t)q[0] : 0ull;  // an exhausted page yields L::ZERO moments
      for (uint32_t jj = 0; jj + 1 < 8; jj++) q[jj] = (L)(q[jj + 1] - q[jj]);
    }
  }
  if (kSpec) return;
  // min / max: wave reduce, block reduce through LDS, one atomic pair per block
  for (int dlt = 32; dlt >= 1; dlt >>= 1) {
    L o1 = shfl_idx(mn1, (int)(lane_id() ^ dlt)); mn1 = o1 < mn1 ? o1 : mn1;
    L o2 = shfl_idx(mx1, (int)(lane_id() ^ dlt)); mx1 = o2 > mx1 ? o2 : mx1;
    if (has_sec) {
      L o3 = shfl_idx(mn2, (int)(lane_id() ^ dlt)); mn2 = o3 < mn2 ? o3 : mn2;
      L o4 = shfl_idx(mx2, (int)(lane_id() ^ dlt)); mx2 = o4 > mx2 ? o4 : mx2;
    }
  }
  uint64_t PCO_LDS* red = (uint64_t PCO_LDS*)(smem + kSplitLdsRed);
  if (lane_id() == 0) { const uint32_t w = tid >> 6; red[w * 4 + 0] = (uint64_t)mn1; red[w * 4 + 1] = (uint64_t)mx1; red[w * 4 + 2] = (uint64_t)mn2; red[w * 4 + 3] = (uint64_t)mx2; }
  __syncthreads();
  if (tid == 0) {
    uint64_t m1 = ~0ull, x1 = 0, m2 = ~0ull, x2 = 0;
    for (uint32_t w = 0; w < 4; w++) {
      m1 = red[w * 4] < m1 ? red[w * 4] : m1; x1 = red[w * 4 + 1] > x1 ? red[w * 4 + 1] : x1;
      m2 = red[w * 4 + 2] < m2 ? red[w * 4 + 2] : m2; x2 = red[w * 4 + 3] > x2 ? red[w * 4 + 3] : x2;
    }
    if (m1 <= x1) { atomicMin((unsigned long long*)&ws.chunks[t].v[1].minv, (unsigned long long)m1); atomicMax((unsigned long long*)&ws.chunks[t].v[1].maxv, (unsigned long long)x1); }
    if (has_sec && m2 <= x2) { atomicMin((unsigned long long*)&ws.chunks[t].v[2].minv, (unsigned long long)m2); atomicMax((unsigned long long*)&ws.chunks[t].v[2].maxv, (unsigned long long)x2); }
  }
}

template <class L, bool kSpec>
__device__ __forceinline__ void enc_split_mode(const EncWorkspace& ws, const PcoGfxEncodeTask& task, uint32_t t, EncPage PCO_GLOBAL* pg, uint32_t tile, uint32_t mode_kind) {
  if (mode_kind == kClassic) enc_split_body<L, kClassic, kSpec>(ws, task, t, pg, tile);
  else if (mode_kind == kIntMult) enc_split_body<L, kIntMult, kSpec>(ws, task, t, pg, tile);
  else if (mode_kind == kFloatQuant) enc_split_body<L, kFloatQuant, kSpec>(ws, task, t, pg, tile);
  else enc_split_body<L, kFloatMult, kSpec>(ws, task, t, pg, tile);
}

// grid pages * ceil(tiles_per_page / tiles_per_block), 256 threads.  enc_split_kernel<true>: the chunks that speculate on
// 16-bit latents (EncChunk::c16_ok == 1).  enc_split_kernel<false>: full-width latents for the chunks whose c16_ok == want --
// 0: never speculated (launched when the call has such chunks), 2: the speculation failed (launched after the <true> kernel;
// the blocks of all other chunks leave at once, which several tiles per block keep cheap).
// (Eight waves per SIMD -- 64 VGPRs -- for the one-tile kernels: a block issues its loads, computes, stores and leaves, so what is in flight per
//  CU is what its resident blocks hold, and at 100 VGPRs (five blocks) the kernel sat at 3.9 TB/s of its own traffic; at 64: 5.7 TB/s,
//  5.3 -> 3.8 ms per 8192 chunks.  The looping form spills at 64 and is launched for stragglers only.)
template <bool kSpec, bool kLoop>
__global__ __launch_bounds__(256, kLoop ? 1 : 8) void enc_split_kernel(EncWorkspace ws, const PcoGfxEncodeTask* tasks, uint32_t tiles_per_page, uint32_t tiles_per_block, uint32_t want) {
  const uint32_t blocks_per_page = kLoop ? (tiles_per_page + tiles_per_block - 1) / tiles_per_block : tiles_per_page;
  const uint32_t page = blockIdx.x / blocks_per_page, tile0 = (blockIdx.x % blocks_per_page) * (kLoop ? tiles_per_block : 1u);
  EncPage PCO_GLOBAL* pg = (EncPage PCO_GLOBAL*)ws.pages + page;
  if (uni(pg->flags) & kPageFlagMetaOnly) return;
  const uint32_t t = uni(pg->chunk);
  if (uni(ws.chunks[t].status) != PCO_GFX_OK) return;
  if (uni(__hip_atomic_load(&ws.chunks[t].c16_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != want) return;   // (a failed chunk's remaining <true> tiles stop here too)
  const PcoGfxEncodeTask task = tasks[t];
  const int bits = dtype_bits(uni(task.dtype));
  const uint32_t mode_kind = uni(ws.chunks[t].mode_kind);
  if (!kLoop) {
    if ((uint64_t)tile0 * kSplitTile >= uni((uint64_t)pg->n)) return;
    if (bits == 64) enc_split_mode<uint64_t, kSpec>(ws, task, t, pg, tile0, mode_kind);
    else if (bits == 32) enc_split_mode<uint32_t, kSpec>(ws, task, t, pg, tile0, mode_kind);
    else if (bits == 16) enc_split_mode<uint16_t, kSpec>(ws, task, t, pg, tile0, mode_kind);
    else if (bits == 8) enc_split_mode<uint8_t, kSpec>(ws, task, t, pg, tile0, mode_kind);
    return;
  }
  for (uint32_t tile = tile0; tile < tile0 + tiles_per_block && (uint64_t)tile * kSplitTile < uni((uint64_t)pg->n); tile++) {
    if (tile != tile0) __syncthreads();   // the tile body reuses its LDS
    if (bits == 64) enc_split_mode<uint64_t, kSpec>(ws, task, t, pg, tile, mode_kind);
    else if (bits == 32) enc_split_mode<uint32_t, kSpec>(ws, task, t, pg, tile, mode_kind);
    else if (bits == 16) enc_split_mode<uint16_t, kSpec>(ws, task, t, pg, tile, mode_kind);
    else if (bits == 8) enc_split_mode<uint8_t, kSpec>(ws, task, t, pg, tile, mode_kind);
  }
}

// One wave per chunk: would the chunk's latents fit 16 bits?  64 sampled positions are split and differenced the way the
// tiles do it; if one of them is out of range the chunk is taken out of the speculation up front (c16_ok = 0: it goes
// through the full-width kernel once) instead of being split twice.  A sample that fits proves nothing -- the tiles check
// every latent.  Single-page chunks of at least 4096 numbers only (pages break the difference at their starts).
template <class L, int MODE>
__device__ __forceinline__ bool presample_bad(const PcoGfxEncodeTask& task, const EncChunk PCO_GLOBAL* ch) {
  const uint32_t lane = lane_id();
  const uint32_t num_kind = dtype_kind(uni(task.dtype));
  const uint32_t mode_k = uni(ch->mode_k);
  const L mode_base = (L)uni((uint64_t)ch->mode_base); const uint64_t aux_inv = uni((uint64_t)ch->mode_aux), aux_base = uni((uint64_t)ch->mode_aux2);
  const uint32_t order = uni(ch->delta_kind) == kDeltaConsecutive ? uni(ch->delta_order) : 0;
  constexpr bool has_sec = MODE != kClassic;
  const uint64_t n = uni((uint64_t)ch->n);
  const L PCO_GLOBAL* src = (const L PCO_GLOBAL*)task.src;
  constexpr L kBias = sizeof(L) == 1 ? (L)0 : (L)16384, kLmax = (L)~(L)0;
  if (sizeof(L) == 1) return false;
  auto clamp_ref = [&](L x) { return x < kBias ? kBias : (x > (L)(kLmax - kBias) ? (L)(kLmax - kBias) : x); };
  L ref1 = order > 0 ? lmid<L>() : (L)0, ref2 = 0;
  if (order == 0 || has_sec) { L p0, s0; split_one<L, MODE>(num_kind, mode_base, mode_k, aux_inv, aux_base, src[0], p0, s0); if (order == 0) ref1 = clamp_ref(p0); ref2 = clamp_ref(s0); }
  const uint64_t pos = (uint64_t)lane * ((n - 8) / 64);   // order <= 7: positions pos .. pos + order are inside the chunk
  L a[8];
  L sec_last = 0;
#pragma unroll
  for (uint32_t k = 0; k < 8; k++) { L pp = 0, ss = 0; if (k <= order) split_one<L, MODE>(num_kind, mode_base, mode_k, aux_inv, aux_base, src[pos + k], pp, ss); a[k] = pp; if (k == order) sec_last = ss; }
  for (uint32_t r = 0; r < order; r++) {
#pragma unroll
    for (uint32_t idx = 7; idx >= 1; idx--) a[idx] = (L)(a[idx] - a[idx - 1]);
  }
  L d = 0;
#pragma unroll
  for (uint32_t k = 0; k < 8; k++) if (k == order) d = a[k];
  d = (L)(d + (order > 0 ? lmid<L>() : (L)0));
  bool bad = (uint64_t)(L)(d - (L)(ref1 - kBias)) >= kC16KeyRange;
  if (has_sec) bad = bad || (uint64_t)(L)(sec_last - (L)(ref2 - kBias)) >= kC16KeyRange;
  return __any(bad) != 0;
}
template <class L>
__device__ __forceinline__ bool presample_mode(const PcoGfxEncodeTask& task, const EncChunk PCO_GLOBAL* ch, uint32_t mode_kind) {
  if (mode_kind == kClassic) return presample_bad<L, kClassic>(task, ch);
  if (mode_kind == kIntMult) return presample_bad<L, kIntMult>(task, ch);
  if (mode_kind == kFloatQuant) return presample_bad<L, kFloatQuant>(task, ch);
  return presample_bad<L, kFloatMult>(task, ch);
}
__global__ __launch_bounds__(64) void enc_presample_kernel(EncWorkspace ws, const PcoGfxEncodeTask* tasks, uint32_t n_tasks) {
  const uint32_t t = blockIdx.x;
  if (t >= n_tasks) return;
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  if (uni(ch->status) != PCO_GFX_OK || uni(ch->c16_ok) != 1 || uni(ch->n_pages) != 1 || uni((uint64_t)ch->n) < 4096) return;
  const PcoGfxEncodeTask task = tasks[t];
  const int bits = dtype_bits(uni(task.dtype));
  const uint32_t mode_kind = uni(ch->mode_kind);
  bool bad = false;
  if (bits == 64) bad = presample_mode<uint64_t>(task, ch, mode_kind);
  else if (bits == 32) bad = presample_mode<uint32_t>(task, ch, mode_kind);
  else if (bits == 16) bad = presample_mode<uint16_t>(task, ch, mode_kind);
  if (bad && lane_id() == 0) { ch->c16_ok = 0; atomicAdd(ws.need_full0, 1u); }
}
// Full-width latent scratch goes to the chunks that write it, and only to them: every chunk whose c16_ok is `want` and that has no slot yet
// takes the next one.  Run twice: want = 0 behind enc_presample_kernel (chunks that never speculate: lookback, more than 256 bins, taken out by
// the sample, calls without the speculation), want = 2 behind the 16-bit split (chunks in which a tile did not fit).  A synchronous call
// reads the counter back after each and allocates exactly that many slots; an asynchronous one has a slot per chunk up front.
__global__ __launch_bounds__(64) void enc_lat_slots_kernel(EncWorkspace ws, uint32_t n_tasks, uint32_t want) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tasks) return;
  if (ws.chunks[t].c16_ok != want || ws.lat_slot[t] != kNoLatSlot) return;
  ws.lat_slot[t] = atomicAdd(ws.lat_used, 1u);
}

// =========================================================================================================
// K1b: lookback delta (delta/lookback.rs:22-185).  choose_lookbacks is a greedy, strictly element-ordered search
// over 16 proposals (6 brute, 4 "repeating", 6 hashed), so one wave owns one page: each 64-element tile first
// resolves all hash-table proposals in parallel (the table reads of a tile are independent of the choices; the
// in-tile read-after-write hazards are patched with a 64-step broadcast), then lanes 0..15 score the 16
// proposals element by element.  The element loop is a serial chain (the counts and the "repeating" proposals depend
// on the previous choice), so its latency is what matters: the latents of the last kLbRing positions live in an LDS
// ring and the counts of lookbacks <= kLbCountsLds in LDS (both cover the usual short / seasonal lookbacks; longer
// ones fall back to HBM), the 16-way arg-max runs on the DPP network, and the chosen lookbacks of a tile are stored
// once per tile.  Other state: last-index hash tables in HBM (2 x 2^(w+1) u32).
// =========================================================================================================
// LDS layout, in two sizes: the full one for real pages, and a small one for the 6.6 k-number sample pages of the Auto-delta
// trials (thousands of them per call: at 9 KB instead of 18 KB per wave, 16 instead of 8 of them share a CU)
template <uint32_t kCounts, uint32_t kRingN> struct LbCfg {
  static constexpr uint32_t kLbCountsLds = kCounts;
  static constexpr uint32_t kLbRing = kRingN;                        // must be >= 64 + the largest lookback served from LDS
  static constexpr uint32_t kLbLdsCounts = 0;                        // u32[kCounts]
  static constexpr uint32_t kLbLdsHp = kCounts * 4;                  // u32[64][6] hash proposals of the tile
  static constexpr uint32_t kLbLdsRing = kLbLdsHp + 64 * 6 * 4;      // u64[kRingN] latents of the last kRingN positions (by position mod kRingN)
  static constexpr uint32_t kLbLdsHpOther = kLbLdsRing + kRingN * 8; // u64[64][6] latent at the far hashed proposals of the tile (prefetched)
  static constexpr uint32_t kLbLdsHpCnt = kLbLdsHpOther + 64 * 6 * 8;   // u32[64][6] their lookback counts as of the tile start
  static constexpr uint32_t kLbLdsBig = kLbLdsHpCnt + 64 * 6 * 4;       // u32[64] lookbacks > kCounts chosen inside the tile (their global counts are bumped at the tile end)
  static constexpr uint32_t kLbLdsBytes = kLbLdsBig + 64 * 4;
};
typedef LbCfg<1024, 1024> LbFull;   // 18 KB of LDS per page: eight pages per CU (lookbacks beyond ~960 read their latent and count from HBM)
typedef LbCfg<256, 256> LbSmall;   // 9 KB: sixteen pages per CU (the register budget allows no more)
constexpr uint32_t kLbSmallMaxPage = 8192;   // pages up to this size take the small layout

struct LookbackScratch { uint32_t* hash; uint32_t* counts; };  // per page: hash[2 << (wlog+1)], counts[1 << wlog]

#ifdef PCO_LB_TIMING
__device__ unsigned long long g_lb_timing[16];
#define LB_STAMP(idx) do { __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long _n = __builtin_readcyclecounter(); lb_acc[idx] += _n - lb_t0; lb_t0 = _n; } while (0)
#else
#define LB_STAMP(idx) do { } while (0)
#endif
// One wave owns a page, and the LDS traffic of a wave is processed in order: between the steps of the search only the compiler has
// to be kept from reordering.  (A workgroup-scope fence here also waited for every outstanding HBM read -- the next tile's latents and
// table entries, the apply step's read -- and so undid the overlap they were sent early for.  HBM ordering that matters is between an
// atomic and a later read of the same address by the same wave, which the memory pipeline keeps.)
// The last-index tables and far counts of a page are private to the wave that owns the page for the whole kernel: workgroup scope is all
// their atomics and reads need.  (At agent scope every access went past the XCD's L2 to the memory side for cross-XCD coherence nobody
// asked for.)
constexpr int kLbScope = __HIP_MEMORY_SCOPE_WORKGROUP;
__device__ __forceinline__ void lb_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
template <class L, class Cfg>
__device__ void lookback_page(const EncWorkspace& ws, uint32_t t, EncPage PCO_GLOBAL* pg, uint32_t PCO_GLOBAL* hash_tbl, uint32_t PCO_GLOBAL* gcounts) {
  constexpr uint32_t kLbCountsLds = Cfg::kLbCountsLds, kLbRing = Cfg::kLbRing, kLbLdsCounts = Cfg::kLbLdsCounts, kLbLdsHp = Cfg::kLbLdsHp, kLbLdsRing = Cfg::kLbLdsRing;
  constexpr uint32_t kLbLdsHpOther = Cfg::kLbLdsHpOther, kLbLdsHpCnt = Cfg::kLbLdsHpCnt, kLbLdsBig = Cfg::kLbLdsBig;
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  const uint32_t lane = lane_id();
  const uint32_t wlog = uni(ch->window_n_log), state_n = 1u << uni(ch->state_n_log);
  const uint32_t window_n = 1u << wlog, hash_table_n = 2u << wlog, hash_mask = hash_table_n - 1;
  const uint32_t n = (uint32_t)uni((uint64_t)pg->n); const uint64_t pstart = uni((uint64_t)pg->start);
  const L PCO_GLOBAL* pre = sort_ptr<L>(ws, t, 0) + pstart;
  uint32_t PCO_GLOBAL* lbs = lat_ptr<uint32_t>(ws, t, 0) + pstart;
  L PCO_GLOBAL* out = lat_ptr<L>(ws, t, 1) + pstart;
  uint32_t PCO_LDS* lcounts = (uint32_t PCO_LDS*)(enc_lds_base() + kLbLdsCounts);
  uint32_t PCO_LDS* hp = (uint32_t PCO_LDS*)(enc_lds_base() + kLbLdsHp);
  uint64_t PCO_LDS* ring = (uint64_t PCO_LDS*)(enc_lds_base() + kLbLdsRing);
  uint64_t PCO_LDS* hp_other = (uint64_t PCO_LDS*)(enc_lds_base() + kLbLdsHpOther);
  uint32_t PCO_LDS* hp_cnt = (uint32_t PCO_LDS*)(enc_lds_base() + kLbLdsHpCnt);
  uint32_t PCO_LDS* big = (uint32_t PCO_LDS*)(enc_lds_base() + kLbLdsBig);
  // delta state = the first state_n latents, right aligned (lookback.rs:179-181); state_n == 1 from this encoder
  if (lane == 0) for (uint32_t i = 0; i < state_n && i < 8; i++) pg->moments[i] = i < n ? (uint64_t)pre[i] : 0ull;
  if (n <= state_n) return;
  for (uint32_t i = lane; i < state_n; i += 64) ring[i & (kLbRing - 1)] = (uint64_t)pre[i];   // the positions before the first tile
  const uint32_t n_counts = window_n < n ? window_n : n;
  for (uint32_t i = lane; i < kLbCountsLds; i += 64) lcounts[i] = 1;
  for (uint32_t i = kLbCountsLds + lane; i < n_counts; i += 64) gcounts[i] = 1;
  for (uint32_t i = lane; i < 2 * hash_table_n; i += 64) hash_tbl[i] = 0;
  __threadfence_block();
  enc_wave_sync();
  auto hash_fn = [&](uint64_t x) { x = (x ^ (x >> 32)) * 11400714819323197441ull; x = x ^ (x >> 32); return (uint32_t)x & hash_mask; };
  uint32_t proposed = 1;            // (first tile) lanes 0..15: proposed_lookbacks[lane] = min(lane+1, state_n) with state_n == 1
  if (lane < 16) proposed = (lane + 1) < state_n ? (lane + 1) : state_n;
  uint32_t best_lookback = 1, repeating_idx = 0;
  // (later tiles) the "repeating" proposals (slots 6..9) and their counts, and the count of the current best lookback: wave-uniform
  uint32_t ring_lb0 = 1, ring_lb1 = 1, ring_lb2 = 1, ring_lb3 = 1, ring_c0 = 1, ring_c1 = 1, ring_c2 = 1, ring_c3 = 1, cnt_best = 1;
  L mn1 = (L)~(L)0, mx1 = 0; uint32_t mn0 = 0xffffffffu, mx0 = 0;
  // the latent `lb` positions before position i: the LDS ring serves the recent ones
  auto latent_back = [&](uint32_t i, uint32_t lb) { return lb < kLbRing - 64 ? (L)ring[(i - lb) & (kLbRing - 1)] : pre[i - lb]; };
  auto lz_of = [&](L l, L other) { const L d1 = (L)(l - other), d2 = (L)(other - l); const L dlt = d1 < d2 ? d1 : d2; return LBits<L>::v - bitlen<L>(dlt); };
#ifdef PCO_LB_TIMING
  unsigned long long lb_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lb_t0 = __builtin_readcyclecounter(), lb_rounds = 0;
#endif
  // A tile's latents and the last-index table entries of its six hash proposals per element: two dependent HBM round trips.  They are
  // fetched one tile ahead -- right after the previous tile has sent its own table updates, which the reads must observe (same wave,
  // same addresses, in order at L2) -- so that they travel while that tile is being decided.
  auto tile_slots = [&](uint64_t lv_, uint32_t (&slot_)[6]) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const uint64_t bucket = lv_ >> (c == 0 ? 0 : 8);
      slot_[3 * c + 0] = c * hash_table_n + hash_fn(bucket - 1);
      slot_[3 * c + 1] = c * hash_table_n + hash_fn(bucket);
      slot_[3 * c + 2] = c * hash_table_n + hash_fn(bucket + 1);
    }
  };
  // (the latents do not depend on anything the search does, so they are read two tiles ahead: when the table reads of the next tile are
  //  due, their addresses can be formed without waiting)
  auto tile_latent = [&](uint32_t i0t) { return i0t < n && lane < n - i0t ? (uint64_t)pre[i0t + lane] : 0ull; };
  auto tile_fetch = [&](uint32_t i0t, uint64_t lv_, uint32_t (&val_)[6]) {
    const bool a = i0t < n && lane < n - i0t;
    uint32_t sl[6]; tile_slots(lv_, sl);
#pragma unroll
    for (int r = 0; r < 6; r++) val_[r] = a ? __hip_atomic_load(&hash_tbl[sl[r]], __ATOMIC_RELAXED, kLbScope) : 0u;  // L2-served: earlier tiles updated it with atomics
  };
  uint64_t pf_lv = tile_latent(state_n), pf2_lv = tile_latent(state_n + 64); uint32_t pf_val[6];
  tile_fetch(state_n, pf_lv, pf_val);
  bool pend_act = false; uint32_t pend_ie = 0; L pend_lv = 0, pend_other = 0;   // the previous tile's apply step, its read in flight
  auto apply_pending = [&]() {
    if (pend_act) { const L d = (L)(pend_lv - pend_other + lmid<L>()); out[pend_ie] = d; mn1 = d < mn1 ? d : mn1; mx1 = d > mx1 ? d : mx1; }
    pend_act = false;
  };
  for (uint32_t i0 = state_n; i0 < n; i0 += 64) {
    const uint32_t tile_n = n - i0 < 64 ? n - i0 : 64;
    const bool first_tile = i0 == state_n;
    // ---- phase 1: hash proposals of the whole tile ----
    const uint32_t ie = i0 + lane;
    const bool act = lane < tile_n;
    const uint64_t lv = pf_lv;
    if (act) ring[ie & (kLbRing - 1)] = lv;
    uint32_t slot[6], val[6], plb[6];
    tile_slots(lv, slot);
#pragma unroll
    for (int r = 0; r < 6; r++) val[r] = pf_val[r];
    LB_STAMP(0);
    // in-tile hazards: an earlier element of the tile wrote its centre bucket (slot[1] / slot[4]) before we read: each of my six
    // slots needs the LAST earlier lane whose centre slot (same table) equals it.  Eight wave votes per table give every lane the set
    // of lanes whose centre slot agrees with a given slot in its low 8 bits (usually nobody); the few candidates are checked newest
    // first.  (Round 1 broadcast every lane's two centre slots in a 64-step loop: 14 k cycles per tile, most of the kernel once
    // the element loop was gone.)
    {
      uint64_t vote[2][8];
#pragma unroll
      for (int c = 0; c < 2; c++) {
#pragma unroll
        for (int b = 0; b < 8; b++) vote[c][b] = __ballot(act && ((slot[3 * c + 1] >> b) & 1u));
      }
      const uint64_t earlier = __ballot(act) & (((uint64_t)1 << lane) - 1);
      uint64_t cand[6];
#pragma unroll
      for (int r = 0; r < 6; r++) {
        uint64_t m = earlier;
#pragma unroll
        for (int b = 0; b < 8; b++) m &= ((slot[r] >> b) & 1u) ? vote[r / 3][b] : ~vote[r / 3][b];
        cand[r] = act ? m : 0ull;
      }
      for (;;) {
        bool any = false;
#pragma unroll
        for (int r = 0; r < 6; r++) any = any || cand[r] != 0;
        if (!__any(any)) break;
#pragma unroll
        for (int r = 0; r < 6; r++) {
          const uint32_t j = cand[r] ? 63u - (uint32_t)__builtin_clzll(cand[r]) : 0u;
          const uint32_t theirs = (uint32_t)__shfl((int)slot[3 * (r / 3) + 1], (int)j, 64);   // (every lane takes part in the exchange)
          if (cand[r]) {
            if (theirs == slot[r]) { val[r] = i0 + j; cand[r] = 0; }
            else cand[r] &= ~((uint64_t)1 << j);
          }
        }
      }
    }
    LB_STAMP(1);
    if (act) { (void)__hip_atomic_fetch_max(&hash_tbl[slot[1]], ie, __ATOMIC_RELAXED, kLbScope); (void)__hip_atomic_fetch_max(&hash_tbl[slot[4]], ie, __ATOMIC_RELAXED, kLbScope); }
    pf_lv = pf2_lv; pf2_lv = tile_latent(i0 + 128);
    tile_fetch(i0 + 64, pf_lv, pf_val);   // the next tile's (nothing past the page's end)
#pragma unroll
    for (int r = 0; r < 6; r++) {
      const uint32_t lb = ie - val[r];
      const uint32_t pidx = 10 + r;
      plb[r] = lb <= window_n ? lb : (pidx < ie ? pidx : ie);
    }
    uint32_t my_lb = 1;   // lane e keeps the lookback chosen for element e of the tile
    if (first_tile) {
      // ---- the page's first tile, element by element: lanes 0..15 = the 16 proposals.  (The brute-force slots fill up over the first 16
      //      positions and overwrite the "repeating" slots on the way: lookback.rs:129-130.  From the second tile on none of that happens.) ----
#pragma unroll
      for (int r = 0; r < 6; r++) {
        hp[lane * 6 + r] = plb[r];
        if (act && plb[r] >= kLbRing - 64) hp_other[lane * 6 + r] = (uint64_t)pre[ie - plb[r]];
        if (act && plb[r] - 1 >= kLbCountsLds) hp_cnt[lane * 6 + r] = __hip_atomic_load(&gcounts[plb[r] - 1], __ATOMIC_RELAXED, kLbScope);
      }
      uint32_t n_big = 0;   // uniform
      lb_sync();
      for (uint32_t e = 0; e < tile_n; e++) {
        const uint32_t i = i0 + e;
        const L l = (L)ring[i & (kLbRing - 1)];   // uniform
        if (i <= 16) { const uint32_t new_brute = i < 16 ? i : 16; if (lane == new_brute - 1) proposed = new_brute; }
        else if (lane == 15) proposed = 16;       // (slot 15 is a hash slot: rewritten below)
        if (lane >= 10 && lane < 16) proposed = hp[e * 6 + (lane - 10)];
        uint32_t key = 0;
        if (lane < 16) {
          const uint32_t lb = proposed;
          uint32_t cnt; L other;
          const bool hashed = lane >= 10;
          if (lb < kLbRing - 64) other = (L)ring[(i - lb) & (kLbRing - 1)];
          else if (hashed) other = (L)hp_other[e * 6 + (lane - 10)];
          else other = pre[i - lb];
          if (lb - 1 < kLbCountsLds) cnt = lcounts[lb - 1];
          else {  // count at the tile start + the times this lookback was chosen earlier in the tile
            cnt = hashed ? hp_cnt[e * 6 + (lane - 10)] : __hip_atomic_load(&gcounts[lb - 1], __ATOMIC_RELAXED, kLbScope);
            for (uint32_t k = 0; k < n_big; k++) cnt += big[k] == lb ? 1u : 0u;
          }
          const uint32_t goodness = (32u - clz_u32(cnt)) + lz_of(l, other);
          key = (goodness << 4) | (15u - lane);  // max key = max goodness, first proposal on ties (lookback.rs:88-96)
        }
        // arg-max over lanes 0..15 on the DPP network (row 0): after row_shr 1, 2, 4, 8 lane 15 holds the maximum
        { uint32_t o = dpp0<0x111, 0xf>(key); key = o > key ? o : key; o = dpp0<0x112, 0xf>(key); key = o > key ? o : key;
          o = dpp0<0x114, 0xf>(key); key = o > key ? o : key; o = dpp0<0x118, 0xf>(key); key = o > key ? o : key; }
        const uint32_t best_p = 15u - ((uint32_t)__builtin_amdgcn_readlane((int)key, 15) & 15u);
        const uint32_t new_best = (uint32_t)__builtin_amdgcn_readlane((int)proposed, (int)best_p);
        if (new_best != best_lookback) repeating_idx++;
        if (lane == 6 + (repeating_idx & 3u)) proposed = new_best;
        best_lookback = new_best;
        if (lane == e) my_lb = new_best;
        if (new_best - 1 < kLbCountsLds) { if (lane == 0) lcounts[new_best - 1] += 1; }
        else { if (lane == 0) big[n_big] = new_best; n_big++; }
        lb_sync();
      }
      if (lane < n_big) (void)__hip_atomic_fetch_add(&gcounts[big[lane] - 1], 1u, __ATOMIC_RELAXED, kLbScope);   // publish before the next tile prefetches counts
      
      lb_sync();
      // hand the state over to the tile-parallel path
      ring_lb0 = (uint32_t)__builtin_amdgcn_readlane((int)proposed, 6); ring_lb1 = (uint32_t)__builtin_amdgcn_readlane((int)proposed, 7);
      ring_lb2 = (uint32_t)__builtin_amdgcn_readlane((int)proposed, 8); ring_lb3 = (uint32_t)__builtin_amdgcn_readlane((int)proposed, 9);
      auto count_now = [&](uint32_t lb) { return uni(lb - 1 < kLbCountsLds ? lcounts[lb - 1] : __hip_atomic_load(&gcounts[lb - 1], __ATOMIC_RELAXED, kLbScope)); };
      ring_c0 = count_now(ring_lb0); ring_c1 = count_now(ring_lb1); ring_c2 = count_now(ring_lb2); ring_c3 = count_now(ring_lb3);
      cnt_best = count_now(best_lookback);
    } else {
      // ---- later tiles: one lane per element, all 64 decided together under the guess that every element repeats the lookback
      //      of the one before it (B).  Under that guess the "repeating" slots never change and only B's count moves, so an element's
      //      16 candidates and their counts are known without waiting for its predecessors.  The first element that decides
      //      otherwise ends the round: everything before it, and its own decision, were made on the true state; the state is
      //      brought up to date and the rest of the tile is decided again.  Exactly choose_lookbacks' sequence (lookback.rs:101-159),
      //      one round per tile on periodic data instead of 64 dependent steps. ----
      const L l = (L)lv;
      uint32_t s_lb[12], s_lz[12], c_far[6], r_lz0, r_lz1, r_lz2, r_lz3;
#pragma unroll
      for (int k = 0; k < 6; k++) s_lb[k] = (uint32_t)k + 1;           // brute force: 1..6 (the page is past position 16)
#pragma unroll
      for (int r = 0; r < 6; r++) s_lb[6 + r] = plb[r];
      // Every candidate's latent, and the far candidates' counts: all the reads are issued before any is used.  Both the LDS ring and HBM
      // are read for every candidate, unconditionally -- the HBM read of a near candidate goes to the element itself, a cache hit -- because a
      // per-lane "near or far" branch around each read made the 22 reads of a tile wait for one another (12 k cycles per tile at four
      // pages per CU: scripts/lb_timing.py).
      {
        const uint32_t ie_s = act ? ie : i0;   // (lanes past the page's end read a valid position)
        uint32_t all_lb[16];
#pragma unroll
        for (int k = 0; k < 12; k++) all_lb[k] = s_lb[k];
        all_lb[12] = ring_lb0; all_lb[13] = ring_lb1; all_lb[14] = ring_lb2; all_lb[15] = ring_lb3;
        L c_near[16], c_hbm[16]; uint32_t far_cnt[6];
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const bool far = act && all_lb[k] >= kLbRing - 64;
          c_near[k] = (L)ring[(ie_s - all_lb[k]) & (kLbRing - 1)];
          c_hbm[k] = pre[ie_s - (far ? all_lb[k] : 0u)];
        }
#pragma unroll
        for (int r = 0; r < 6; r++) far_cnt[r] = __hip_atomic_load(&gcounts[act && plb[r] - 1 >= kLbCountsLds ? plb[r] - 1 : 0u], __ATOMIC_RELAXED, kLbScope);
        uint32_t lz_all[16];
#pragma unroll
        for (int k = 0; k < 16; k++) lz_all[k] = act ? lz_of(l, all_lb[k] >= kLbRing - 64 ? c_hbm[k] : c_near[k]) : 0u;
#pragma unroll
        for (int k = 0; k < 12; k++) s_lz[k] = lz_all[k];
        r_lz0 = lz_all[12]; r_lz1 = lz_all[13]; r_lz2 = lz_all[14]; r_lz3 = lz_all[15];
#pragma unroll
        for (int r = 0; r < 6; r++) c_far[r] = act && plb[r] - 1 >= kLbCountsLds ? far_cnt[r] : 0u;
      }
      LB_STAMP(2);
      uint32_t e_start = 0;
      // count `lb` += k for everything that mirrors it
      auto add_count = [&](uint32_t lb, uint32_t k) -> uint32_t {   // returns the new count
        if (k == 0) return 0u;
        uint32_t now;
        if (lb - 1 < kLbCountsLds) { now = uni(lcounts[lb - 1]) + k; if (lane == 0) lcounts[lb - 1] = now; }
        else {
          uint32_t old = 0; if (lane == 0) old = __hip_atomic_fetch_add(&gcounts[lb - 1], k, __ATOMIC_RELAXED, kLbScope);
          now = uni(old) + k;
#pragma unroll
          for (int r = 0; r < 6; r++) c_far[r] += plb[r] == lb ? k : 0u;
        }
        if (ring_lb0 == lb) ring_c0 = now; if (ring_lb1 == lb) ring_c1 = now; if (ring_lb2 == lb) ring_c2 = now; if (ring_lb3 == lb) ring_c3 = now;
        return now;
      };
      for (;;) {
#ifdef PCO_LB_TIMING
        lb_rounds++;
#endif
        const uint32_t B = best_lookback;
        const uint32_t cb = cnt_best + (lane - e_start);   // B's count as this element sees it
        uint32_t best_g = 0, best = 0;
        auto consider = [&](uint32_t lb, uint32_t lz, uint32_t cnt) { const uint32_t g = (32u - clz_u32(lb == B ? cb : cnt)) + lz; if (g > best_g) { best_g = g; best = lb; } };
#pragma unroll
        for (int k = 0; k < 6; k++) consider(s_lb[k], s_lz[k], lcounts[k]);
        consider(ring_lb0, r_lz0, ring_c0); consider(ring_lb1, r_lz1, ring_c1); consider(ring_lb2, r_lz2, ring_c2); consider(ring_lb3, r_lz3, ring_c3);
#pragma unroll
        for (int r = 0; r < 6; r++) {
          const uint32_t lb = plb[r];
          const uint32_t idx = lb - 1 < kLbCountsLds ? lb - 1 : 0u;
          const uint32_t near = lcounts[idx];
          consider(lb, s_lz[6 + r], lb - 1 < kLbCountsLds ? near : c_far[r]);
        }
        uint64_t mism = __ballot(act && lane >= e_start && best != B);
        const uint32_t e_star = mism ? (uint32_t)__builtin_ctzll(mism) : tile_n;
        if (lane >= e_start && lane < e_star) my_lb = B;
        lb_sync();   // (the counts were read by every lane before lane 0 changes them)
        const uint32_t nb = add_count(B, e_star - e_start);
        if (nb) cnt_best = nb;
        if (e_star >= tile_n) break;
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)best, (int)e_star);
        if (lane == e_star) my_lb = c;
        repeating_idx++;
        // the slot takes the new lookback BEFORE its count is bumped; add_count then mirrors the bump into the slot's count
        const uint32_t rs = repeating_idx & 3u;
        const uint32_t c_lz = act && lane > e_star ? lz_of(l, latent_back(ie, c)) : 0u;   // (only the elements still to be decided; c may reach before the page for earlier ones)
        if (rs == 0) { ring_lb0 = c; r_lz0 = c_lz; } else if (rs == 1) { ring_lb1 = c; r_lz1 = c_lz; } else if (rs == 2) { ring_lb2 = c; r_lz2 = c_lz; } else { ring_lb3 = c; r_lz3 = c_lz; }
        cnt_best = add_count(c, 1u);
        best_lookback = c;
        e_start = e_star + 1;
        lb_sync();
        if (e_start >= tile_n) break;
      }
      
      lb_sync();
      LB_STAMP(3);
    }
    if (act) { lbs[ie] = my_lb; mn0 = my_lb < mn0 ? my_lb : mn0; mx0 = my_lb > mx0 ? my_lb : mx0; }
    // ---- apply (lookback.rs:166-185): l[i] -= l[i - lb], + MID; reads the un-delta'd copy so it is parallel.  The read of l[i - lb] is
    //      sent now and used one tile later, after that tile's own reads have gone out: nothing waits for it on its own ----
    apply_pending();
    pend_act = act; pend_ie = ie; pend_lv = (L)lv; pend_other = pre[act ? ie - my_lb : i0];
    LB_STAMP(4);
  }
  apply_pending();
#ifdef PCO_LB_TIMING
  if (lane == 0) { for (int k = 0; k < 5; k++) atomicAdd(&g_lb_timing[k], lb_acc[k]); atomicAdd(&g_lb_timing[5], lb_rounds); atomicAdd(&g_lb_timing[6], (unsigned long long)((n - state_n + 63) / 64)); atomicAdd(&g_lb_timing[7], 1ull); }
#endif
  for (int dlt = 32; dlt >= 1; dlt >>= 1) {
    L o1 = shfl_idx(mn1, (int)(lane ^ dlt)); mn1 = o1 < mn1 ? o1 : mn1;
    L o2 = shfl_idx(mx1, (int)(lane ^ dlt)); mx1 = o2 > mx1 ? o2 : mx1;
    uint32_t o3 = __shfl_xor(mn0, dlt, 64); mn0 = o3 < mn0 ? o3 : mn0;
    uint32_t o4 = __shfl_xor(mx0, dlt, 64); mx0 = o4 > mx0 ? o4 : mx0;
  }
  if (lane == 0) {
    atomicMin((unsigned long long*)&ws.chunks[t].v[1].minv, (unsigned long long)mn1); atomicMax((unsigned long long*)&ws.chunks[t].v[1].maxv, (unsigned long long)mx1);
    atomicMin((unsigned long long*)&ws.chunks[t].v[0].minv, (unsigned long long)mn0); atomicMax((unsigned long long*)&ws.chunks[t].v[0].maxv, (unsigned long long)mx0);
  }
}

template <class Cfg>
// grid = the lookback pages only (page_ids lists them): a launch over all pages left the blocks of the other pages to exit at once, and
// with the lookback trial at every 4th page of an Auto-delta wave all the work landed on the two XCDs that blocks 1 and 5 (mod 8) go to.
__global__ __launch_bounds__(64) void enc_lookback_kernel(EncWorkspace ws, const uint32_t* page_ids, uint32_t n_lb_pages, uint32_t* lb_scratch, uint64_t scratch_stride_u32, const uint32_t* only_if) {
  if (blockIdx.x >= n_lb_pages) return;
  if (only_if && uni(only_if[blockIdx.x]) == 0) return;   // (behind enc_lookback_pipe_kernel: only the pages it handed back)
  const uint32_t p = page_ids[blockIdx.x];
  EncPage PCO_GLOBAL* pg = (EncPage PCO_GLOBAL*)ws.pages + p;
  if (uni(pg->flags) & kPageFlagMetaOnly) return;
  const uint32_t t = uni(pg->chunk);
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  if (uni(ch->status) != PCO_GFX_OK || uni(ch->delta_kind) != kDeltaLookback) return;
  uint32_t PCO_GLOBAL* base = (uint32_t PCO_GLOBAL*)lb_scratch + (uint64_t)blockIdx.x * scratch_stride_u32;
  const uint32_t wlog = uni(ch->window_n_log);
  uint32_t PCO_GLOBAL* hash_tbl = base; uint32_t PCO_GLOBAL* gcounts = base + (4ull << wlog);
  const int bits = dtype_bits(uni(ch->dtype));
  if (bits == 64) lookback_page<uint64_t, Cfg>(ws, t, pg, hash_tbl, gcounts);
  else if (bits == 32) lookback_page<uint32_t, Cfg>(ws, t, pg, hash_tbl, gcounts);
  else if (bits == 16) lookback_page<uint16_t, Cfg>(ws, t, pg, hash_tbl, gcounts);
  else lookback_page<uint8_t, Cfg>(ws, t, pg, hash_tbl, gcounts);
}

// =========================================================================================================
// K2: exact equal-count quantile histogram (histograms.rs).  The reference's quickselect output is a pure
// function of the sorted multiset (property-tested against the literal algorithm by the CPU test suite):
// walk bins b with end ranks c_count(b) = ceil((b+1)n/B); a run of equal values that straddles the end of
// the bin containing its first rank is a "constant run" (histograms.rs:142-161), everything else merges
// into the pending bin.  We therefore only need rank -> (value, run start, run end) queries:
//   * direct path  (max-min < 4096): LDS counting histogram over value space + prefix sums;
//   * sorted path  (otherwise): stable LSD radix sort (8-bit digits, significant digits only).
// =========================================================================================================
struct HistRec { uint32_t st, en; };
constexpr uint32_t kWalkRecBytes = 16384;   // u32[6][256] tables | u32[256] run starts | u32[256] run ends | u64[4][256] value, next, predecessor, successor
#ifdef PCO_HIST_TIMING
__device__ unsigned long long g_hist_timing[16];   // [0..4]: narrow kernel (count, prefix, lookups, emit, vars); [8..12]: the wide kernels
#endif

// The walk over the bins, resumable: with more than 256 bins (compression levels 9..12) the rank records are produced a window of 256
// bins at a time and the walk stops when its next bin lies beyond the window.  Records are indexed by (bin - win_base).
template <class L> struct HistWalk {
  uint32_t pos = 0; L pos_value = 0; bool pending = false; uint32_t pending_start = 0; L pending_lower = 0; uint32_t next_avail = 0, n_hist = 0;
};
template <class L> __device__ __forceinline__ void walk_store(uint64_t PCO_LDS* a, const HistWalk<L>& w) {
  a[0] = w.pos; a[1] = (uint64_t)w.pos_value; a[2] = w.pending ? 1u : 0u; a[3] = w.pending_start; a[4] = (uint64_t)w.pending_lower; a[5] = w.next_avail; a[6] = w.n_hist;
}
template <class L> __device__ __forceinline__ HistWalk<L> walk_load(const uint64_t PCO_LDS* a) {
  HistWalk<L> w; w.pos = (uint32_t)a[0]; w.pos_value = (L)a[1]; w.pending = a[2] != 0; w.pending_start = (uint32_t)a[3]; w.pending_lower = (L)a[4]; w.next_avail = (uint32_t)a[5]; w.n_hist = (uint32_t)a[6];
  return w;
}
template <class L>
__device__ __forceinline__ void hist_walk(HistWalk<L>& w, uint32_t win_base, uint32_t win_end, uint32_t n_lat, uint32_t bins_log,
                                          const L PCO_LDS* rv, const uint32_t PCO_LDS* rst, const uint32_t PCO_LDS* ren,
                                          const L PCO_LDS* rnext, const L PCO_LDS* rpred, const L PCO_LDS* rsucc, const PlanRef& plan) {
  // sequential (one lane); at most 2 * 2^bins_log iterations over all windows
  const uint64_t n = n_lat, B = (uint64_t)1 << bins_log;
  // floor((pos << bins_log) / n) by one multiplication: M = ceil(2^64 / n) is exact for dividends below 2^64 / n, and ours stay below
  // 2^36 with n <= 2^24 (the 64-bit divisions were most of this serial loop)
  const uint64_t magic = n > 1 ? (~0ull / n) + 1 : 0ull;
  auto bin_idx = [&](uint64_t pos) { return n > 1 ? (uint32_t)__umul64hi(pos << bins_log, magic) : (uint32_t)(pos << bins_log); };
  auto c_count = [&](uint32_t b) { return (uint32_t)((((uint64_t)b + 1) * n + B - 1) >> bins_log); };
  uint32_t pos = w.pos; L pos_value = w.pos_value;
  bool pending = w.pending; uint32_t pending_start = w.pending_start; L pending_lower = w.pending_lower;
  uint32_t next_avail = w.next_avail, n_hist = w.n_hist;
  auto emit = [&](uint32_t start, uint32_t end, L lower, L upper) {
    plan.hcount()[n_hist] = end - start; plan.hlower()[n_hist] = (uint64_t)lower; plan.hupper()[n_hist] = (uint64_t)upper; n_hist++;
  };
  while (pos < n_lat) {
    const uint32_t target_abs = bin_idx(pos);
    if (target_abs >= win_end) break;
    const uint32_t target = target_abs - win_base;
    const uint32_t c = c_count(target_abs);
    const L v = rv[target]; const uint32_t st = rst[target], en = ren[target];
    if (en <= c) {  // every run in [pos, c) fits: absorb and complete at c
      if (!pending) { pending_start = pos; pending_lower = pos_value; }
      emit(pending_start, c, pending_lower, v);
      pending = false; next_avail = target_abs + 1;
      pos = c; pos_value = rnext[target];
    } else {        // the run [st, en) of value v straddles c: constant run
      if (st > pos && !pending) { pending = true; pending_start = pos; pending_lower = pos_value; }
      const uint32_t mid = st + (en - st) / 2;
      uint32_t b = bin_idx(mid);
      if (b > next_avail) {
        const uint32_t spare = b - 1;
        if (pending) { emit(pending_start, st, pending_lower, rpred[target]); pending = false; next_avail = spare + 1; }
        else b = spare;
      }
      if (!pending) { pending = true; pending_start = st; pending_lower = v; }
      if (en >= c_count(b)) { emit(pending_start, en, pending_lower, v); pending = false; next_avail = b + 1; }
      pos = en; pos_value = rsucc[target];
    }
  }
  w.pos = pos; w.pos_value = pos_value; w.pending = pending; w.pending_start = pending_start; w.pending_lower = pending_lower; w.next_avail = next_avail; w.n_hist = n_hist;
}
template <class L>
__device__ __forceinline__ void hist_state_machine(uint32_t n_lat, uint32_t bins_log, L first_value,
                                                   const L PCO_LDS* rv, const uint32_t PCO_LDS* rst, const uint32_t PCO_LDS* ren,
                                                   const L PCO_LDS* rnext, const L PCO_LDS* rpred, const L PCO_LDS* rsucc,
                                                   const PlanRef& plan, uint32_t& n_hist_out) {
  HistWalk<L> w; w.pos_value = first_value;
  hist_walk<L>(w, 0u, 1u << bins_log, n_lat, bins_log, rv, rst, ren, rnext, rpred, rsucc, plan);
  n_hist_out = w.n_hist;
}

// The same walk for one window covering every bin, with everything that costs a 64-bit multiplication taken out of the serial loop
// (bin_idx and c_count were 100 instructions of a 130-instruction step: 280 k cycles per variable, 44 % of enc_hist_kernel on the
// benchmark's data, 70 % of the 16 k-counter kernel's on float-mult primaries -- scripts/hist_timing.py).  One thread per bin fills,
// before the walk:  pre[0][b] = c_count(b);  pre[1][b] = bin_idx(c_count(b)), the bin the walk stands in after completing bin b;
// pre[2][b] = bin_idx(en_b), ... after a constant run;  pre[3][b] = bin_idx(middle of the run);  pre[4][b] = c_count(pre[3][b]);
// pre[5][b] = c_count(pre[3][b] - 1).
// pre[0] = c_count(b); pre[1] = bin_idx(c_count(b)); pre[2] = bin_idx(en_b); pre[3] = bin_idx(middle of the run); pre[4] = c_count(pre[3]); pre[5] = c_count(pre[3] - 1)
__device__ __forceinline__ void hist_precompute(uint32_t b, uint32_t n_lat, uint32_t bins_log, uint32_t st, uint32_t en, uint32_t (&pre)[6]) {
  const uint64_t n = n_lat, B = (uint64_t)1 << bins_log;
  const uint64_t magic = n > 1 ? (~0ull / n) + 1 : 0ull;
  auto bin_idx = [&](uint64_t pos) { return n > 1 ? (uint32_t)__umul64hi(pos << bins_log, magic) : (uint32_t)(pos << bins_log); };
  auto c_count = [&](uint32_t bb) { return (uint32_t)((((uint64_t)bb + 1) * n + B - 1) >> bins_log); };
  const uint32_t c = c_count(b);
  const uint32_t bm = bin_idx(st + (en - st) / 2);
  pre[0] = c; pre[1] = bin_idx(c); pre[2] = bin_idx(en); pre[3] = bm; pre[4] = c_count(bm); pre[5] = bm > 0 ? c_count(bm - 1) : 0u;
}
// Everything a step of the walk may need about bin b, as ONE 64-byte record (round 6): the six tables above, the run [st, en) and the four
// values (at the bin's last rank, the one after it, the run's predecessor and successor; 64-bit whatever the latent type).  A step used to
// read twelve LDS words from twelve arrays -- twelve DS instructions issued by a lone lane, some 16 cycles each that nothing hides, before
// the round trip itself; now it is four 16-byte reads.
constexpr uint32_t kWalkRecDwords = 16;
constexpr uint32_t kWalkScratchBytes = 256 * kWalkRecDwords * 4;   // what hist_emit needs of its scratch area when it walks itself
__device__ __forceinline__ void hist_pack_record(uint32_t PCO_LDS* rec, uint32_t b, const uint32_t (&pre)[6], uint32_t st, uint32_t en, uint64_t v, uint64_t vnext, uint64_t vpred, uint64_t vsucc) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 PCO_LDS* r4 = (u32x4 PCO_LDS*)(rec + b * kWalkRecDwords);
  u32x4 a, c, d, e;
  a[0] = pre[0]; a[1] = pre[1]; a[2] = pre[2]; a[3] = pre[3];
  c[0] = pre[4]; c[1] = pre[5]; c[2] = st; c[3] = en;
  d[0] = (uint32_t)v; d[1] = (uint32_t)(v >> 32); d[2] = (uint32_t)vnext; d[3] = (uint32_t)(vnext >> 32);
  e[0] = (uint32_t)vpred; e[1] = (uint32_t)(vpred >> 32); e[2] = (uint32_t)vsucc; e[3] = (uint32_t)(vsucc >> 32);
  r4[0] = a; r4[1] = c; r4[2] = d; r4[3] = e;
}
// The walk of hist_walk above for one window covering every bin, by one lane, over the records; bins go straight to the plan (stores nobody waits for)
template <class L>
__device__ __forceinline__ uint32_t hist_walk_rec(uint32_t n_lat, L first_value, const uint32_t PCO_LDS* rec, const PlanRef& plan) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  uint32_t pos = 0, target = 0; uint64_t pos_value = (uint64_t)first_value;
  bool pending = false; uint32_t pending_start = 0; uint64_t pending_lower = 0;
  uint32_t next_avail = 0, n_hist = 0;
  auto emit = [&](uint32_t start, uint32_t end, uint64_t lower, uint64_t upper) { plan.hcount()[n_hist] = end - start; plan.hlower()[n_hist] = lower; plan.hupper()[n_hist] = upper; n_hist++; };
  while (pos < n_lat) {
    const u32x4 PCO_LDS* r4 = (const u32x4 PCO_LDS*)(rec + target * kWalkRecDwords);
    const u32x4 ra = r4[0], rb = r4[1], rc = r4[2], rd = r4[3];
    const uint32_t c = ra[0], nb_c = ra[1], nb_en = ra[2], bm = ra[3], c_bm = rb[0], c_bm1 = rb[1], st = rb[2], en = rb[3];
    const uint64_t v = (uint64_t)rc[0] | ((uint64_t)rc[1] << 32), vnext = (uint64_t)rc[2] | ((uint64_t)rc[3] << 32);
    const uint64_t vpred = (uint64_t)rd[0] | ((uint64_t)rd[1] << 32), vsucc = (uint64_t)rd[2] | ((uint64_t)rd[3] << 32);
    if (en <= c) {  // every run in [pos, c) fits: absorb and complete at c
      if (!pending) { pending_start = pos; pending_lower = pos_value; }
      emit(pending_start, c, pending_lower, v);
      pending = false; next_avail = target + 1;
      pos = c; pos_value = vnext;
      target = nb_c;
    } else {        // the run [st, en) of value v straddles c: constant run
      if (st > pos && !pending) { pending = true; pending_start = pos; pending_lower = pos_value; }
      uint32_t b = bm, cb = c_bm;
      if (b > next_avail) {
        const uint32_t spare = b - 1;
        if (pending) { emit(pending_start, st, pending_lower, vpred); pending = false; next_avail = spare + 1; }
        else { b = spare; cb = c_bm1; }
      }
      if (!pending) { pending = true; pending_start = st; pending_lower = v; }
      if (en >= cb) { emit(pending_start, en, pending_lower, v); pending = false; next_avail = b + 1; }
      pos = en; pos_value = vsucc;
      target = nb_en;
    }
  }
  return n_hist;
}

// The histogram of one variable from its rank records, by the whole block (every thread calls this after the records are in LDS and a
// barrier).  When no run of equal values straddles a bin end (ren[b] <= c_count(b) for every b: data without heavy ties) and there
// are at least as many latents as bins, the state machine above visits the bins in order and emits bin b = ranks
// [c_count(b-1), c_count(b)) with the bounds (value at its first rank, value at its last rank): that is done by one thread per
// bin.  Otherwise one thread walks the state machine (a serial loop of up to 2 x bins steps, ~100 k cycles: most of a small chunk's
// histogram time before this shortcut).
template <class L>
__device__ __forceinline__ void hist_emit(uint32_t n_lat, uint32_t bins_log, L first_value, const L PCO_LDS* rv, const uint32_t PCO_LDS* rst, const uint32_t PCO_LDS* ren,
                                          const L PCO_LDS* rnext, const L PCO_LDS* rpred, const L PCO_LDS* rsucc, const PlanRef& plan, EncVar PCO_GLOBAL* ev, uint32_t path,
                                          uint32_t PCO_LDS* pre /* kWalkScratchBytes of 16-byte aligned scratch (the counters / sort area: done with by now) */,
                                          uint8_t PCO_GLOBAL* defer = nullptr /* where to leave records + tables for enc_hist_walk_kernel instead of walking here */) {
  const uint32_t tid = threadIdx.x, B = 1u << bins_log;
  const uint64_t n64 = n_lat;
  auto c_count = [&](uint32_t b) { return (uint32_t)((((uint64_t)b + 1) * n64 + B - 1) >> bins_log); };
  const bool mine_ok = tid >= B || ren[tid] <= c_count(tid);
  const bool simple = __syncthreads_and(mine_ok && n_lat >= B) != 0;
  if (simple) {
    if (tid < B) {
      const uint32_t c0 = tid == 0 ? 0u : c_count(tid - 1), c1 = c_count(tid);
      plan.hcount()[tid] = c1 - c0; plan.hlower()[tid] = (uint64_t)(tid == 0 ? first_value : rnext[tid - 1]); plan.hupper()[tid] = (uint64_t)rv[tid];
    }
    if (tid == 0) { ev->n_hist = B; ev->hist_path = path; }
  } else {
    uint32_t p6[6] = {0, 0, 0, 0, 0, 0};
    if (tid < B) hist_precompute(tid, n_lat, bins_log, rst[tid], ren[tid], p6);
    if (defer != nullptr) {   // the walk is one thread's work: with two 1024-thread blocks per CU it was most of the kernel's time; it gets a wave of its own later
      if (tid < B) {
        uint32_t PCO_GLOBAL* g32 = (uint32_t PCO_GLOBAL*)defer;
#pragma unroll
        for (uint32_t k = 0; k < 6; k++) g32[k * 256 + tid] = p6[k];
        g32[6 * 256 + tid] = rst[tid]; g32[7 * 256 + tid] = ren[tid];
        uint64_t PCO_GLOBAL* g64 = (uint64_t PCO_GLOBAL*)(defer + 8192);
        g64[tid] = (uint64_t)rv[tid]; g64[256 + tid] = (uint64_t)rnext[tid]; g64[512 + tid] = (uint64_t)rpred[tid]; g64[768 + tid] = (uint64_t)rsucc[tid];
      }
      if (tid == 0) { ev->hist_path = path; ev->walk_pending = 1u; }
      return;
    }
    if (tid < B) hist_pack_record(pre, tid, p6, rst[tid], ren[tid], (uint64_t)rv[tid], (uint64_t)rnext[tid], (uint64_t)rpred[tid], (uint64_t)rsucc[tid]);
    __syncthreads();
    if (tid == 0) { const uint32_t nh = hist_walk_rec<L>(n_lat, first_value, pre, plan); ev->n_hist = nh; ev->hist_path = path; }
  }
}

// LDS layout of enc_hist_kernel
// records first (u64[256] x4, u32[256] x2, block-scan scratch u32[512]), then the counters: u32[R + 8] counts / prefix of
// the direct path (R = 4096 in enc_hist_kernel, 32768 in enc_hist_wide_kernel) or the radix counters of the sorted path
constexpr uint32_t kHistLdsRecV = 0;
constexpr uint32_t kHistLdsCounts = 4 * 2048 + 2 * 1024 + 2048;
constexpr uint32_t kWideHistRange = 32768, kMidHistRange = 16384;   // the two LDS-counting tiers above kDirectHistRange (2 resp. 1 block per CU)
constexpr uint32_t kSelBucketsLog = 13, kSelBuckets = 1u << kSelBucketsLog, kSelMaxNeeded = 1600;
constexpr uint32_t kSmallHistCap = 8192;   // wide-range variables of at most this many latents are ordered whole in LDS (enc_hist_small_kernel)
__host__ __device__ constexpr uint32_t hist_lds_bytes(uint32_t range) { return kHistLdsCounts + (range + 8) * 4; }
constexpr uint32_t kHistLdsBytes = hist_lds_bytes(kDirectHistRange);
// enc_hist_sort_kernel: radix counters u32[2048] | bucket prefix u32[8200] | marks u32[256] | marked-bucket list u32[2][1600 + 1]
constexpr uint32_t kHistSortLdsBytes = kHistLdsCounts + (2048 + kSelBuckets + 8 + kSelBuckets / 32 + 2 * (kSelMaxNeeded + 1)) * 4;

template <bool B> struct BoolC { static constexpr bool value = B; };   // (a compile-time flag handed to a generic lambda)
// counts[c] += 1 for the lanes with `on`; the lanes that share the first active lane's value are added with one atomic
// (secondary latents and residuals are often dominated by one value: 64 same-address LDS atomics would serialise)
__device__ __forceinline__ void hist_count(uint32_t PCO_LDS* counts, uint32_t c, bool on, bool aggregate) {
  if (!aggregate) { if (on) atomicAdd((uint32_t*)&counts[c], 1u); return; }
  const uint64_t act = __ballot(on);
  if (act == 0) return;
  const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)__builtin_ctzll(act));
  const uint64_t same = __ballot(on && c == c0);
  if (on) {
    if (c != c0) atomicAdd((uint32_t*)&counts[c], 1u);
    else if ((same & (((uint64_t)1 << lane_id()) - 1)) == 0) atomicAdd((uint32_t*)&counts[c0], (uint32_t)__popcll(same));
  }
}

// T threads per block; R = value range handled by LDS counting.  kWide: only variables whose range lies in
// [kDirectHistRange, R) are processed (enc_hist_wide_kernel); otherwise those are left to that kernel.
template <class L, uint32_t T, uint32_t R, bool kWide, bool kSort>
__device__ void hist_var(const EncWorkspace& ws, uint32_t t, uint32_t var, uint32_t bins_log) {
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  EncVar PCO_GLOBAL* ev = &ch->v[var];
  const PlanRef plan = plan_ref(ws, t, var);
  const uint32_t tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
  const uint32_t n_lat = ev->n_lat;
  if (n_lat == 0) { if (tid == 0) ev->n_hist = 0; return; }
  const L minv = (L)ev->minv, maxv = (L)ev->maxv;
  const L range = (L)(maxv - minv);
  // enc_hist_kernel: range < 4096; enc_hist_wide_kernel<16384>: [4096, 16384), <32768>: [16384, 32768); enc_hist_sort_kernel: the rest
  // (what the launcher has to run after this kernel: bit 0 the select / sort kernels, bit 1 the 16384-counter kernel, bit 2 the 32768-counter one,
  //  bit 3 the whole-variable-in-LDS kernel)
  if (!kSort && !kWide && ch->big != 0 && tid == 0) atomicOr(ws.need_sort, 1u);
  if (!kSort && !kWide && ch->big == 0 && (uint64_t)range >= kDirectHistRange && tid == 0) atomicOr(ws.need_sort, n_lat <= kSmallHistCap ? 8u : ((uint64_t)range >= kWideHistRange ? 1u : ((uint64_t)range >= kMidHistRange ? 4u : 2u)));
  const bool big = ch->big != 0;   // more than 256 bins: every variable of the chunk takes the sort path, whatever its range
  if (!kSort && big) return;
  // short variables beyond the first counting tier (the Auto-delta trial samples, mostly) are cheaper to order whole in LDS than to count in
  // 16 k - 32 k counters: enc_hist_small_kernel
  if (!big && (kWide || kSort) && n_lat <= kSmallHistCap) return;
  if (!big && (kSort ? (uint64_t)range < kWideHistRange : (kWide ? ((uint64_t)range < (R == kWideHistRange ? kMidHistRange : kDirectHistRange) || (uint64_t)range >= R) : (uint64_t)range >= kDirectHistRange))) return;
  if (kSort && !big && ev->hist_path != 2) return;   // the radix-sort path is the fallback of enc_hist_select_kernel (encode_hist_select.hip), which flags what it gave up on
  // Stored latents = every position that is not among the first `skip` of its page (wrapped/chunk_compressor.rs:129-140).
  const L PCO_GLOBAL* lat = lat_ptr<L>(ws, t, var);
  const uint32_t n_all = (uint32_t)ch->n, skip = ev->lat_start, plow = ch->page_low, pr = ch->page_r;
  const bool single_page = ch->n_pages == 1;
  const bool exact_paging = ch->exact_paging != 0; const uint32_t n_pg = ch->n_pages;
  const EncPage PCO_GLOBAL* pgl = (const EncPage PCO_GLOBAL*)ws.pages + ch->page_first;
  auto exact_start = [&](uint32_t i) {   // PagingSpec::Exact: start of the page holding position i (last page whose start <= i)
    uint32_t lo = 0, hi = n_pg;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)pgl[mid].start <= i) lo = mid; else hi = mid; }
    return (uint64_t)pgl[lo].start;
  };
  auto stored = [&](uint32_t i) { return skip == 0 || (single_page ? i >= skip : (uint64_t)i - (exact_paging ? exact_start(i) : page_start_of(i, plow, pr)) >= skip); };
  uint8_t PCO_LDS* smem = enc_lds_base();
  uint32_t PCO_LDS* counts = (uint32_t PCO_LDS*)(smem + kHistLdsCounts);
  L PCO_LDS* rv = (L PCO_LDS*)(smem + kHistLdsRecV);
  L PCO_LDS* rnext = (L PCO_LDS*)(smem + kHistLdsRecV + 2048);
  L PCO_LDS* rpred = (L PCO_LDS*)(smem + kHistLdsRecV + 4096);
  L PCO_LDS* rsucc = (L PCO_LDS*)(smem + kHistLdsRecV + 6144);
  uint32_t PCO_LDS* rst = (uint32_t PCO_LDS*)(smem + kHistLdsRecV + 8192);
  uint32_t PCO_LDS* ren = (uint32_t PCO_LDS*)(smem + kHistLdsRecV + 9216);
  uint32_t PCO_LDS* scan = (uint32_t PCO_LDS*)(smem + kHistLdsRecV + 10240);
  const uint32_t B = 1u << bins_log;
  const uint64_t n64 = n_lat;
  auto c_count = [&](uint32_t b) { return (uint32_t)((((uint64_t)b + 1) * n64 + B - 1) >> bins_log); };
  __syncthreads();
  if ((uint64_t)range < R && !big) {
    // ---------------- direct path ----------------
#ifdef PCO_HIST_TIMING
    unsigned long long ht0 = __builtin_readcyclecounter();
#define HIST_STAMP(idx) do { __syncthreads(); if (tid == 0) { const unsigned long long _n = __builtin_readcyclecounter(); atomicAdd(&g_hist_timing[(kWide ? 8 : 0) + idx], _n - ht0); ht0 = _n; } } while (0)
#else
#define HIST_STAMP(idx) do { } while (0)
#endif
    constexpr uint32_t PER = R / T;   // counters per thread in the prefix pass
    for (uint32_t i = tid; i < R + 8; i += T) counts[i] = 0;
    __syncthreads();
    {  // counting; the compact copy (x - min, u16) is written on the way
      uint16_t PCO_GLOBAL* clat = clat_ptr(ws, t, var);
      // (wave-uniform) few distinct values: same-address atomics would serialise, aggregate per wave.  The lookbacks of a lookback delta
      // (variable 0) are thousands of distances of which nearly all decisions take ONE, the period: aggregated whatever their range
      // (configs[3]: this kernel 3.5 -> 1.5 ms per 4096 chunks; which way a wave counts changes no count)
      const bool agg = uni((uint32_t)(((uint64_t)range < 256 || var == 0) ? 1u : 0u)) != 0;
      const bool c16 = uni(ch->c16_ok) == 1 && var != 0;   // the split left 16-bit latents relative to c16_ref (in the compact copy's place)
      const uint16_t c16_off = (uint16_t)((uint64_t)minv - (uint64_t)ch->c16_ref[var == 2 ? 1 : 0]);
      // A block's time IS the kernel's time (a block takes what it takes alone, five of them on a CU or two: scripts/hist_timing.py), and the count
      // was 70 % of it.  Round 6: (1) EVERY position is counted and the unstored ones -- the first `skip` of each page -- are taken out again
      // afterwards (a few atomics per page): the per-latent "is it stored" (page arithmetic, a binary search under PagingSpec::Exact) and the
      // aggregate-or-not branch sat INSIDE the loop, some forty instructions and six branches per latent; an unstored position may hold anything,
      // hence the range check.  (2) 16-bit latents: a thread takes eight CONSECUTIVE ones with one 16-byte load, two rounds ahead of the one it
      // counts (the eight 2-byte loads of a round were waited for before its first atomic).  Which thread counts a latent changes no count.
      auto count_all = [&](auto agg_c) {
        constexpr bool kAgg = decltype(agg_c)::value;
        uint32_t base = 0;
        if (c16) {   // (the 16-bit latents stay as the split wrote them, relative to c16_ref: the page kernels take that as their reference)
          if (((uint64_t)(uintptr_t)clat & 15u) == 0 && n_all >= 8 * T) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 PCO_GLOBAL* c4 = (const u32x4 PCO_GLOBAL*)clat;
            const uint32_t n_it = n_all / (8 * T);
            auto fetch = [&](uint32_t it) { return c4[(uint64_t)(it < n_it ? it : n_it - 1) * T + tid]; };   // (unconditional: past the end the last round again, unused)
            u32x4 cur = fetch(0), nx1 = fetch(1);
            for (uint32_t it = 0; it < n_it; it++) {
              const u32x4 nx2 = fetch(it + 2);
              const uint32_t w[4] = {cur[0], cur[1], cur[2], cur[3]};
#pragma unroll
              for (uint32_t k = 0; k < 8; k++) {
                const uint32_t c = (uint32_t)(uint16_t)(((k & 1u) ? w[k >> 1] >> 16 : w[k >> 1]) - c16_off);
                hist_count(counts, c, c < R, kAgg);
              }
              cur = nx1; nx1 = nx2;
            }
            base = n_it * 8 * T;
          }
          for (; base + 8 * T <= n_all; base += 8 * T) {
            uint16_t x[8];
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = clat[base + k * T + tid];
#pragma unroll
            for (int k = 0; k < 8; k++) { const uint32_t c = (uint32_t)(uint16_t)(x[k] - c16_off); hist_count(counts, c, c < R, kAgg); }
          }
        } else {
          for (; base + 8 * T <= n_all; base += 8 * T) {
            L x[8];
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = lat[base + k * T + tid];
#pragma unroll
            for (int k = 0; k < 8; k++) {
              const L d = (L)(x[k] - minv);
              clat[base + k * T + tid] = (uint16_t)d;
              hist_count(counts, (uint32_t)d, (uint64_t)d < R, kAgg);
            }
          }
        }
        for (uint32_t i0 = base; i0 < n_all; i0 += T) {   // whole waves enter hist_count
          const uint32_t i = i0 + tid;
          const uint64_t d = i < n_all ? (c16 ? (uint64_t)(uint16_t)(clat[i] - c16_off) : (uint64_t)(L)(lat[i] - minv)) : 0ull;
          if (i < n_all && !c16) clat[i] = (uint16_t)d;
          hist_count(counts, (uint32_t)d, i < n_all && d < R, kAgg);
        }
      };
      // A CONSTANT variable whose 16-bit latents the split already wrote (the adjustments of exact decimals under float-mult, BASELINE configs[2]:
      // every secondary; constant columns) has nothing to count: all n_lat stored latents are the minimum (1.39 -> 0.1 ms per 8192 such variables)
      const bool constant16 = (uint64_t)range == 0 && c16;
      if (constant16) { if (tid == 0) counts[0] = n_lat; }
      else if (agg) count_all(BoolC<true>{}); else count_all(BoolC<false>{});
      if (skip != 0 && !constant16) {   // (delta state positions: wrapped/chunk_compressor.rs:129-140)
        __syncthreads();
        for (uint32_t q = tid; q < n_pg * skip; q += T) {
          const uint32_t pi = q / skip, j = q - pi * skip;
          const uint64_t pst = pgl[pi].start;
          if (j < (uint64_t)pgl[pi].n) {
            const uint64_t d = c16 ? (uint64_t)(uint16_t)(clat[pst + j] - c16_off) : (uint64_t)(L)(lat[pst + j] - minv);
            if (d < R) atomicSub((uint32_t*)&counts[(uint32_t)d], 1u);
          }
        }
      }
    }
    __syncthreads();
    HIST_STAMP(0);
    // exclusive prefix over R counters: PER per thread + block scan
    uint32_t s = 0;
    for (uint32_t k = 0; k < PER; k++) s += counts[tid * PER + k];
    uint32_t incl = wave_incl_scan(s);
    if (lane == 63) scan[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0; for (uint32_t w = 0; w < wave; w++) wbase += scan[w];
    uint32_t run = wbase + incl - s;
    for (uint32_t k = 0; k < PER; k++) { const uint32_t c = counts[tid * PER + k]; counts[tid * PER + k] = run; run += c; }
    if (tid == T - 1) counts[R] = run;  // == n_lat
    __syncthreads();
    HIST_STAMP(1);
    auto lookup = [&](uint32_t r, L& value, uint32_t& st, uint32_t& en) {
      uint32_t lo = 0, hi = R;  // last v with P[v] <= r
      while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (counts[mid] <= r) lo = mid; else hi = mid; }
      value = (L)(minv + (L)lo); st = counts[lo]; en = counts[lo + 1];
    };
    if (tid < B) {
      const uint32_t c = c_count(tid);
      L v; uint32_t st, en; lookup(c - 1, v, st, en);
      rv[tid] = v; rst[tid] = st; ren[tid] = en;
      L x; uint32_t a, b2;
      if (c < n_lat) { lookup(c, x, a, b2); rnext[tid] = x; } else rnext[tid] = 0;
      if (st > 0) { lookup(st - 1, x, a, b2); rpred[tid] = x; } else rpred[tid] = 0;
      if (en < n_lat) { lookup(en, x, a, b2); rsucc[tid] = x; } else rsucc[tid] = 0;
    }
    __syncthreads();
    HIST_STAMP(2);
    hist_emit<L>(n_lat, bins_log, minv, rv, rst, ren, rnext, rpred, rsucc, plan, ev, 0u, (uint32_t PCO_LDS*)(smem + kHistLdsCounts),
                 ws.walk != nullptr ? (uint8_t PCO_GLOBAL*)ws.walk + ((uint64_t)t * 3 + var) * kWalkRecBytes : (uint8_t PCO_GLOBAL*)nullptr);   // (enc_hist_kernel is launched without the record buffer unless PCO_GFX_HIST_DEFER=1: its five blocks per CU hide their own walks)
    __syncthreads();
    HIST_STAMP(3);
#ifdef PCO_HIST_TIMING
    if (tid == 0) atomicAdd(&g_hist_timing[(kWide ? 8 : 0) + 4], 1ull);
#endif
    return;
  }
  if constexpr (!kSort) return;
  // ---------------- sorted path (256 threads) ----------------
  // The histogram only asks for the values (and their runs) at <= 2^bins_log ranks, so most of a wide-range variable
  // never needs ordering.  (1) count the latents into 8192 buckets by the top bits of key = x - min; (2) mark the
  // buckets that hold a queried rank, plus the nearest non-empty bucket on either side (predecessor / successor of
  // a run); (3) stable LSD radix sort (8-bit digits) of the latents of the marked buckets only -- typically 5-20 %
  // of the chunk; (4) answer the rank queries in the sorted subset, translating ranks through the bucket prefix.
  if constexpr (kSort) {
  constexpr uint32_t NB = kSelBuckets;
  const uint32_t sig_bits = bitlen<L>(range);
  const uint32_t bshift = sig_bits > kSelBucketsLog ? sig_bits - kSelBucketsLog : 0u;
  const uint32_t npass = sig_bits == 0 ? 1u : (sig_bits + 7) / 8;   // (a constant variable of a big chunk still goes through one pass: S must hold it)
  L PCO_GLOBAL* bufA = sort_ptr<L>(ws, t, 0);
  L PCO_GLOBAL* bufB = sort_ptr<L>(ws, t, 1);
  uint32_t PCO_LDS* cnt = counts;            // [4][256]
  uint32_t PCO_LDS* cursor = counts + 1024;  // [4][256]
  uint32_t PCO_LDS* P = counts + 2048;                 // u32[NB + 8]: bucket counts, then exclusive prefix
  uint32_t PCO_LDS* need = P + NB + 8;                 // u32[NB / 32] bitmap
  uint32_t PCO_LDS* nl_k = need + NB / 32;             // u32[kSelMaxNeeded] marked buckets in order
  uint32_t PCO_LDS* nl_oc = nl_k + kSelMaxNeeded;      // u32[kSelMaxNeeded + 1] sorted-subset offset of each marked bucket
  auto bucket_of = [&](L x) { return (uint32_t)((L)(x - minv) >> bshift); };
  // (1) bucket counts + prefix
  for (uint32_t i = tid; i < NB + 8; i += 256) P[i] = 0;
  for (uint32_t i = tid; i < NB / 32; i += 256) need[i] = 0;
  __syncthreads();
  {
    uint32_t base = 0;
    for (; base + 8 * 256 <= n_all; base += 8 * 256) {
      L x[8];
#pragma unroll
      for (int k = 0; k < 8; k++) x[k] = lat[base + k * 256 + tid];
#pragma unroll
      for (int k = 0; k < 8; k++) if (stored(base + k * 256 + tid)) atomicAdd((uint32_t*)&P[bucket_of(x[k])], 1u);
    }
    for (uint32_t i = base + tid; i < n_all; i += 256) if (stored(i)) atomicAdd((uint32_t*)&P[bucket_of(lat[i])], 1u);
  }
  __syncthreads();
  {
    constexpr uint32_t PER = NB / 256;
    uint32_t s0 = 0;
    for (uint32_t k = 0; k < PER; k++) s0 += P[tid * PER + k];
    const uint32_t incl = wave_incl_scan(s0);
    if (lane == 63) scan[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0; for (uint32_t w = 0; w < wave; w++) wbase += scan[w];
    uint32_t run = wbase + incl - s0;
    for (uint32_t k = 0; k < PER; k++) { const uint32_t c = P[tid * PER + k]; P[tid * PER + k] = run; run += c; }
    if (tid == 255) P[NB] = run;  // == n_lat
  }
  __syncthreads();
  auto bucket_of_rank = [&](uint32_t r) {   // the (non-empty) bucket holding rank r: last k with P[k] <= r
    uint32_t lo = 0, hi = NB;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (P[mid] <= r) lo = mid; else hi = mid; }
    return lo;
  };
  // (2) mark
  for (uint32_t bq = tid; bq < B; bq += 256) {
    const uint32_t c = c_count(bq);
    for (uint32_t which = 0; which < 2; which++) {
      const uint32_t r = c - 1 + which;
      if (r >= n_lat) continue;
      const uint32_t k = bucket_of_rank(r);
      atomicOr((uint32_t*)&need[k >> 5], 1u << (k & 31));
      if (P[k] > 0) { const uint32_t kp = bucket_of_rank(P[k] - 1); atomicOr((uint32_t*)&need[kp >> 5], 1u << (kp & 31)); }
      if (P[k + 1] < n_lat) { const uint32_t kn = bucket_of_rank(P[k + 1]); atomicOr((uint32_t*)&need[kn >> 5], 1u << (kn & 31)); }
    }
  }
  __syncthreads();
  // list of marked buckets (ascending) with the offset of each one's latents in the sorted subset
  uint32_t n_need = 0, n_sub = 0;
  {
    const uint32_t word = need[tid];        // NB / 32 == 256 words: one per thread
    uint32_t nb_here = __popc(word), el_here = 0;
    for (uint32_t w = word; w; w &= w - 1) { const uint32_t k = tid * 32 + (uint32_t)__builtin_ctz(w); el_here += P[k + 1] - P[k]; }
    const uint32_t i_nb = wave_incl_scan(nb_here), i_el = wave_incl_scan(el_here);
    if (lane == 63) { scan[wave] = i_nb; scan[16 + wave] = i_el; }
    __syncthreads();
    uint32_t b_nb = 0, b_el = 0; for (uint32_t w = 0; w < wave; w++) { b_nb += scan[w]; b_el += scan[16 + w]; }
    uint32_t slot = b_nb + i_nb - nb_here, off = b_el + i_el - el_here;
    for (uint32_t w = word; w; w &= w - 1) {
      const uint32_t k = tid * 32 + (uint32_t)__builtin_ctz(w);
      if (slot < kSelMaxNeeded) { nl_k[slot] = k; nl_oc[slot] = off; }
      slot++; off += P[k + 1] - P[k];
    }
    n_need = scan[0] + scan[1] + scan[2] + scan[3]; n_sub = scan[16] + scan[17] + scan[18] + scan[19];
  }
  __syncthreads();
  if (tid == 0) nl_oc[n_need < kSelMaxNeeded ? n_need : kSelMaxNeeded] = n_sub;
  const bool subset = n_need <= kSelMaxNeeded;   // always true (<= 6 marks per queried rank); otherwise sort everything
  auto needed = [&](L x) { const uint32_t k = bucket_of(x); return !subset || ((need[k >> 5] >> (k & 31)) & 1u) != 0; };
  const uint32_t n_sort = subset ? n_sub : n_lat;
  __syncthreads();
  // (3a) gather the marked buckets' latents into sort buffer B, in any order (the keys carry no payload, so only the
  //      radix passes themselves have to be stable): one more read of the variable instead of two by the first pass
  if (subset) {
    uint32_t PCO_LDS* fill = scan + 32;
    if (tid == 0) *fill = 0;
    __syncthreads();
    auto gather_group = [&](uint32_t i, L x, bool in_range) {
      const bool act = in_range && stored(i) && needed(x);
      const uint64_t m = __ballot(act);
      if (m == 0) return;
      uint32_t base0 = 0;
      if (lane == 0) base0 = atomicAdd((uint32_t*)fill, (uint32_t)__popcll(m));
      base0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)base0);
      if (act) bufB[base0 + __popcll(m & (((uint64_t)1 << lane) - 1))] = x;
    };
    uint32_t i0 = 0;
    for (; i0 + 8 * 256 <= n_all; i0 += 8 * 256) {
      L x[8];
#pragma unroll
      for (int k = 0; k < 8; k++) x[k] = lat[i0 + 256 * k + tid];
#pragma unroll
      for (int k = 0; k < 8; k++) gather_group(i0 + 256 * k + tid, x[k], true);
    }
    for (; i0 < n_all; i0 += 256) { const uint32_t i = i0 + tid; gather_group(i, i < n_all ? lat[i] : (L)0, i < n_all); }
    __threadfence_block();
    __syncthreads();
  }
  // (3) stable LSD radix sort of those latents
  for (uint32_t p = 0; p < npass; p++) {
    const L PCO_GLOBAL* in = p == 0 ? (subset ? bufB : lat) : ((p & 1) ? bufA : bufB);
    L PCO_GLOBAL* out = (p & 1) ? bufB : bufA;
    const uint32_t shift = 8 * p;
    // without the gather (never in practice) pass 0 reads the page-structured latent array, skipping each page's junk prefix
    const bool raw0 = p == 0 && !subset;
    const uint32_t n_in = raw0 ? n_all : n_sort;
    const uint32_t q = (n_in + 3) / 4;       // per-wave contiguous quarter (keeps the scatter stable)
    const uint32_t w_begin = wave * q < n_in ? wave * q : n_in;
    const uint32_t w_end = (wave + 1) * q < n_in ? (wave + 1) * q : n_in;
    for (uint32_t i = tid; i < 1024; i += 256) cnt[i] = 0;
    __syncthreads();
    auto count_one = [&](uint32_t i, L x) {
      if (raw0 && !stored(i)) return;
      const uint32_t d = (uint32_t)(((L)(x - minv)) >> shift) & 255u;
      atomicAdd((uint32_t*)&cnt[wave * 256 + d], 1u);
    };
    {
      uint32_t i0 = w_begin;
      for (; i0 + 8 * 64 <= w_end; i0 += 8 * 64) {   // 8 loads in flight per lane (a wave has only three others to hide behind)
        L x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = in[i0 + 64 * k + lane];
#pragma unroll
        for (int k = 0; k < 8; k++) count_one(i0 + 64 * k + lane, x[k]);
      }
      for (uint32_t i = i0 + lane; i < w_end; i += 64) count_one(i, in[i]);
    }
    __syncthreads();
    {  // thread = digit: exclusive scan over digits of the per-digit totals, then per-wave bases
      const uint32_t c0 = cnt[tid], c1 = cnt[256 + tid], c2 = cnt[512 + tid], c3 = cnt[768 + tid];
      const uint32_t tot = c0 + c1 + c2 + c3;
      const uint32_t incl = wave_incl_scan(tot);
      if (lane == 63) scan[wave] = incl;
      __syncthreads();
      uint32_t wbase = 0; for (uint32_t w = 0; w < wave; w++) wbase += scan[w];
      const uint32_t excl = wbase + incl - tot;
      cursor[tid] = excl; cursor[256 + tid] = excl + c0; cursor[512 + tid] = excl + c0 + c1; cursor[768 + tid] = excl + c0 + c1 + c2;
    }
    __syncthreads();
    uint32_t PCO_LDS* mycur = cursor + wave * 256;
    auto scatter_group = [&](uint32_t i0, L x) {   // 64 consecutive elements, in order: match-any ranking within the wave
      const uint32_t i = i0 + lane;
      const bool act = i < w_end && (!raw0 || stored(i));
      const uint32_t d = act ? ((uint32_t)(((L)(x - minv)) >> shift) & 255u) : 0xffffffffu;
      uint64_t m = __ballot(act);
      if (m == 0) return;
#pragma unroll
      for (uint32_t bit = 0; bit < 8; bit++) { const uint64_t bm = __ballot((d >> bit) & 1); m &= ((d >> bit) & 1) ? bm : ~bm; }
      const uint64_t lt = ((uint64_t)1 << lane) - 1;
      const uint32_t rank = __popcll(m & lt), gcount = __popcll(m);
      const uint32_t basec = act ? mycur[d] : 0;
      enc_wave_sync();
      if (act && rank == 0) mycur[d] = basec + gcount;
      enc_wave_sync();
      if (act) out[basec + rank] = x;
    };
    {
      uint32_t i0 = w_begin;
      for (; i0 + 8 * 64 <= w_end; i0 += 8 * 64) {   // the loads of eight groups are issued together, the groups still scatter in order
        L x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = in[i0 + 64 * k + lane];
#pragma unroll
        for (int k = 0; k < 8; k++) scatter_group(i0 + 64 * k, x[k]);
      }
      for (; i0 < w_end; i0 += 64) scatter_group(i0, i0 + lane < w_end ? in[i0 + lane] : (L)0);
    }
    __threadfence_block();
    __syncthreads();
  }
  const L PCO_GLOBAL* S = (npass & 1) ? bufA : bufB;
  // (4) rank queries.  A rank r lives in bucket k = bucket_of_rank(r), whose latents occupy S[oc .. oc + count) -- runs of
  // equal values never leave their bucket, so run bounds are found inside that window and translated back.
  auto window_of = [&](uint32_t k, uint32_t& oc) {   // position of marked bucket k in the list
    if (!subset) { oc = P[k]; return; }
    uint32_t lo = 0, hi = n_need;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (nl_k[mid] <= k) lo = mid; else hi = mid; }
    oc = nl_oc[lo];
  };
  auto value_at = [&](uint32_t r) { const uint32_t k = bucket_of_rank(r); uint32_t oc; window_of(k, oc); return S[oc + (r - P[k])]; };
  auto lookup_sorted = [&](uint32_t r, L& value, uint32_t& st, uint32_t& en) {
    const uint32_t k = bucket_of_rank(r); uint32_t oc; window_of(k, oc);
    const uint32_t at = oc + (r - P[k]), wend = oc + (P[k + 1] - P[k]);
    value = S[at];
    uint32_t lo = oc, hi = at;   // first index with S[idx] >= value (S[at] == value)
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (S[mid] < value) lo = mid + 1; else hi = mid; }
    st = P[k] + (lo - oc);
    lo = at + 1; hi = wend;      // first index with S[idx] > value
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (S[mid] <= value) lo = mid + 1; else hi = mid; }
    en = P[k] + (lo - oc);
  };
  if (B <= 256) {
    if (tid < B) {
      const uint32_t c = c_count(tid);
      L v; uint32_t st, en; lookup_sorted(c - 1, v, st, en);
      rv[tid] = v; rst[tid] = st; ren[tid] = en;
      rnext[tid] = c < n_lat ? value_at(c) : (L)0;
      rpred[tid] = st > 0 ? value_at(st - 1) : (L)0;
      rsucc[tid] = en < n_lat ? value_at(en) : (L)0;
    }
    __syncthreads();
    hist_emit<L>(n_lat, bins_log, minv, rv, rst, ren, rnext, rpred, rsucc, plan, ev, 1u, (uint32_t PCO_LDS*)(smem + kHistLdsCounts),
                 ws.walk != nullptr ? (uint8_t PCO_GLOBAL*)ws.walk + ((uint64_t)t * 3 + var) * kWalkRecBytes : (uint8_t PCO_GLOBAL*)nullptr);
    __syncthreads();
  } else {
    // more than 256 bins: rank records a window of 256 bins at a time.  A window none of whose bins is straddled by a run of equal
    // values, entered with nothing pending at its first bin's start rank, is emitted by one thread per bin; otherwise one thread
    // walks it (hist_walk).  The walk's state lives in LDS between windows.
    uint64_t PCO_LDS* wst = (uint64_t PCO_LDS*)(smem + kHistLdsRecV + 10240 + 256);   // (behind the block-scan scratch)
    if (tid == 0) { HistWalk<L> w0; w0.pos_value = minv; walk_store<L>(wst, w0); }
    __syncthreads();
    for (uint32_t wb = 0; wb < B; wb += 256) {
      const uint32_t bq = wb + tid;
      if (tid < 256 && bq < B) {
        const uint32_t c = c_count(bq);
        L v; uint32_t st, en; lookup_sorted(c - 1, v, st, en);
        rv[tid] = v; rst[tid] = st; ren[tid] = en;
        rnext[tid] = c < n_lat ? value_at(c) : (L)0;
        rpred[tid] = st > 0 ? value_at(st - 1) : (L)0;
        rsucc[tid] = en < n_lat ? value_at(en) : (L)0;
      }
      __syncthreads();
      const uint32_t wn = B - wb < 256 ? B - wb : 256;
      const HistWalk<L> w = walk_load<L>(wst);
      const bool clean = !w.pending && n_lat >= B && w.pos == (wb == 0 ? 0u : c_count(wb - 1));
      const bool mine_ok = tid >= wn || ren[tid] <= c_count(wb + tid);
      const bool simple = __syncthreads_and(mine_ok && clean) != 0;
      if (simple) {
        if (tid < wn) {
          const uint32_t c0 = bq == 0 ? 0u : c_count(bq - 1), c1 = c_count(bq);
          const uint32_t at = w.n_hist + tid;
          plan.hcount()[at] = c1 - c0; plan.hlower()[at] = (uint64_t)(tid == 0 ? w.pos_value : rnext[tid - 1]); plan.hupper()[at] = (uint64_t)rv[tid];
        }
        __syncthreads();
        if (tid == 0) { HistWalk<L> w2 = w; w2.pos = c_count(wb + wn - 1); w2.pos_value = rnext[wn - 1]; w2.next_avail = wb + wn; w2.n_hist = w.n_hist + wn; walk_store<L>(wst, w2); }
      } else if (tid == 0) {
        HistWalk<L> w2 = w;
        hist_walk<L>(w2, wb, wb + wn, n_lat, bins_log, rv, rst, ren, rnext, rpred, rsucc, plan);
        walk_store<L>(wst, w2);
      }
      __syncthreads();
    }
    if (tid == 0) { ev->n_hist = (uint32_t)wst[6]; ev->hist_path = 1; }
    __syncthreads();
  }
  }
}

template <uint32_t T, uint32_t R, bool kWide, bool kSort>
__device__ __forceinline__ void hist_chunk(const EncWorkspace& ws, uint32_t t) {
  const EncChunk PCO_GLOBAL* ch = (const EncChunk PCO_GLOBAL*)ws.chunks + t;
  if (uni(ch->status) != PCO_GFX_OK) return;
  const int bits = dtype_bits(uni(ch->dtype));
  const uint32_t ubl = uni(ch->unopt_bins_log);
  for (uint32_t var = 0; var < 3; var++) {
    if (!uni(ch->v[var].present)) continue;
    // secondary latents get fewer bins (wrapped/chunk_compressor.rs:238-248)
    const uint32_t bl = var == 2 ? (ubl < 6 ? ubl : 6) : ubl;
    if (var == 0) hist_var<uint32_t, T, R, kWide, kSort>(ws, t, var, bl);
    else if (bits == 64) hist_var<uint64_t, T, R, kWide, kSort>(ws, t, var, bl);
    else if (bits == 32) hist_var<uint32_t, T, R, kWide, kSort>(ws, t, var, bl);
    else if (bits == 16) hist_var<uint16_t, T, R, kWide, kSort>(ws, t, var, bl);
    else hist_var<uint8_t, T, R, kWide, kSort>(ws, t, var, bl);
  }
}
__global__ __launch_bounds__(256) void enc_hist_kernel(EncWorkspace ws, uint32_t n_tasks) {
  if (blockIdx.x < n_tasks) hist_chunk<256, kDirectHistRange, false, false>(ws, blockIdx.x);
}
// value ranges >= 32768: bucket pre-selection + radix sort of the needed part
__global__ __launch_bounds__(256) void enc_hist_sort_kernel(EncWorkspace ws, uint32_t n_tasks) {
  if (blockIdx.x < n_tasks) hist_chunk<256, kDirectHistRange, false, true>(ws, blockIdx.x);
}
// value ranges in [4096, 16384) / [16384, 32768): LDS counting with 1024-thread blocks (66 KB / 131 KB of counters: 2 / 1 per CU)
template <uint32_t R>
__global__ __launch_bounds__(1024) void enc_hist_wide_kernel(EncWorkspace ws, uint32_t n_tasks) {
  if (blockIdx.x < n_tasks) hist_chunk<1024, R, true, false>(ws, blockIdx.x);
}

// The bin walks the 1024-thread histogram kernels left behind (EncVar::walk_pending): one wave per chunk, records and tables from HBM
// into LDS, the walk by one lane, the bins copied out by the wave.  Thousands of walks at once instead of one per resident block.
constexpr uint32_t kWalkLdsBytes = kWalkScratchBytes;   // the records (round 6: 21.5 -> 16 KB, nine walks on a CU instead of seven; the bins go straight to the plan)
__global__ __launch_bounds__(64) void enc_hist_walk_kernel(EncWorkspace ws, uint32_t n_tasks) {
  const uint32_t t = blockIdx.x;
  if (t >= n_tasks) return;
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  if (uni(ch->status) != PCO_GFX_OK) return;
  uint32_t PCO_LDS* rec = (uint32_t PCO_LDS*)enc_lds_base();
  const uint32_t lane = lane_id(), ubl = uni(ch->unopt_bins_log);
  for (uint32_t var = 0; var < 3; var++) {
    EncVar PCO_GLOBAL* ev = &ch->v[var];
    if (!uni(ev->present) || !uni(ev->walk_pending)) continue;
    const uint32_t bins_log = var == 2 ? (ubl < 6 ? ubl : 6) : ubl, B = 1u << bins_log;
    const uint8_t PCO_GLOBAL* src = (const uint8_t PCO_GLOBAL*)ws.walk + ((uint64_t)t * 3 + var) * kWalkRecBytes;
    const uint32_t PCO_GLOBAL* g32 = (const uint32_t PCO_GLOBAL*)src; const uint64_t PCO_GLOBAL* g64 = (const uint64_t PCO_GLOBAL*)(src + 8192);
    for (uint32_t i = lane; i < B; i += 64) {
      uint32_t p6[6];
#pragma unroll
      for (uint32_t k = 0; k < 6; k++) p6[k] = g32[k * 256 + i];
      hist_pack_record(rec, i, p6, g32[6 * 256 + i], g32[7 * 256 + i], g64[i], g64[256 + i], g64[512 + i], g64[768 + i]);
    }
    enc_wave_sync();
    const PlanRef plan = plan_ref(ws, t, var);
    if (lane == 0) { const uint32_t nh = hist_walk_rec<uint64_t>(uni(ev->n_lat), (uint64_t)ev->minv, rec, plan); ev->n_hist = nh; ev->walk_pending = 0u; }
    enc_wave_sync();
  }
}

// =========================================================================================================
// K3: bin optimisation DP, weight quantisation, tANS encoder tables, fallback decision (one wave per chunk)
// =========================================================================================================
__device__ __forceinline__ float log2_approx_dev(float x) {  // bin_optimization.rs:19-43
  constexpr float Z = 0.674f;
  constexpr uint32_t SIGNIF_MASK = 0x7FFFFF;
  const uint32_t Z_SIGNIF = __float_as_uint(Z) & SIGNIF_MASK;
  constexpr float Bc = 2.0f / Z;
  constexpr float Cc = -Bc / (6.0f * Z);
  constexpr float Ac = -Bc - Cc;
  const uint32_t bits = __float_as_uint(x);
  const uint32_t exp = bits >> 23, signif = bits & SIGNIF_MASK;
  const uint32_t high_bit = signif > Z_SIGNIF ? 1u : 0u;
  const uint32_t log_int = exp + high_bit - 127u;
  const float normalized = __uint_as_float(((0x7Fu ^ high_bit) << 23) | signif);
  const float t0 = __fmul_rn(Cc, normalized);
  const float t1 = __fadd_rn(Bc, t0);
  const float t2 = __fmul_rn(normalized, t1);
  const float t3 = __fadd_rn((float)log_int, Ac);
  return __fadd_rn(t3, t2);
}
template <class L>
__device__ __forceinline__ float bin_cost_dev(float meta, L lower, L upper, uint32_t count, float total_log2) {  // :46-57
  const float countf = (float)count;
  const float ans_cost = __fsub_rn(total_log2, log2_approx_dev(countf));
  const float offset_cost = (float)bitlen<L>((L)(upper - lower));
  return __fadd_rn(meta, __fmul_rn(__fadd_rn(ans_cost, offset_cost), countf));
}

// LDS layout of the training kernels for a bin capacity CAP: 256 (one wave per chunk, enc_train_kernel) or 4096 (levels 9..12, one block
// of 1024 threads per chunk, enc_train_big_kernel; there the float weights reuse the DP's cost array, the cumulative weights the
// cumulative counts and the state symbols the back pointers, each dead by the time its tenant arrives: 138 KB instead of 178).
template <uint32_t CAP> struct TrainLds {
  static constexpr bool kAlias = CAP > 256;
  static constexpr uint32_t cc = 0;                                   // u32[CAP + 1] cumulative counts
  static constexpr uint32_t best = cc + (CAP + 4) * 4;                // f32[CAP + 1]
  static constexpr uint32_t low = (best + (CAP + 4) * 4 + 7) & ~7u;   // u64[CAP]
  static constexpr uint32_t up = low + CAP * 8;                       // u64[CAP]
  static constexpr uint32_t bj = up + CAP * 8;                        // u16[CAP]
  static constexpr uint32_t part = bj + CAP * 2;                      // u16[CAP][2] partition
  static constexpr uint32_t fw = kAlias ? best : part + CAP * 4;      // f32[CAP] float weights
  static constexpr uint32_t w = kAlias ? part + CAP * 4 : fw + CAP * 4;   // u32[CAP] weights
  // u16[4096] state symbols.  CAP 256 (round 6): over everything in front of the weights -- the histogram, the DP's arrays and the float weights are
  // dead once thread 0 has quantised the weights (a block_sync away from the table build, which reads w, cum and sym only): 19.2 -> 11 KB a wave,
  // fourteen waves on a CU instead of eight for a kernel that is one latency chain per wave
  static constexpr uint32_t sym = kAlias ? bj : 0;
  static constexpr uint32_t cum = kAlias ? cc : w + CAP * 4;          // u32[CAP + 1]
  static constexpr uint32_t red = kAlias ? w + CAP * 4 : cum + (CAP + 4) * 4;   // block reduction scratch: f32[16] u32[16]
  static constexpr uint32_t bytes = red + 256;
  static_assert(kAlias || w >= 8192, "the state symbols end in front of the weights");
};
constexpr uint32_t kTrainLdsBytes = TrainLds<256>::bytes;
constexpr uint32_t kTrainBigLdsBytes = TrainLds<kBigBins>::bytes;
static_assert(kTrainBigLdsBytes <= 160 * 1024, "one block per CU");

// NW waves per block: 1 (CAP 256) or 16 (CAP 4096).  Everything but the DP's inner loop is the work of wave 0.
template <class L, uint32_t CAP, uint32_t NW>
__device__ void train_var(const EncWorkspace& ws, uint32_t t, uint32_t var) {
  typedef TrainLds<CAP> TL;
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  EncVar PCO_GLOBAL* ev = &ch->v[var];
  const PlanRef plan = plan_ref(ws, t, var);
  const uint32_t lane = lane_id(), tid = threadIdx.x, wave = tid >> 6;
  auto block_sync = [&]() { if (NW > 1) __syncthreads(); else enc_wave_sync(); };
  uint8_t PCO_LDS* smem = enc_lds_base();
  uint32_t PCO_LDS* cc = (uint32_t PCO_LDS*)(smem + TL::cc);
  float PCO_LDS* best = (float PCO_LDS*)(smem + TL::best);
  L PCO_LDS* lows = (L PCO_LDS*)(smem + TL::low);
  L PCO_LDS* ups = (L PCO_LDS*)(smem + TL::up);
  uint16_t PCO_LDS* bj = (uint16_t PCO_LDS*)(smem + TL::bj);
  uint16_t PCO_LDS* part = (uint16_t PCO_LDS*)(smem + TL::part);
  uint32_t PCO_LDS* wts = (uint32_t PCO_LDS*)(smem + TL::w);
  float PCO_LDS* fw = (float PCO_LDS*)(smem + TL::fw);
  uint16_t PCO_LDS* ssym = (uint16_t PCO_LDS*)(smem + TL::sym);
  uint32_t PCO_LDS* cum = (uint32_t PCO_LDS*)(smem + TL::cum);
  float PCO_LDS* red_c = (float PCO_LDS*)(smem + TL::red);
  uint32_t PCO_LDS* red_j = (uint32_t PCO_LDS*)(smem + TL::red + 64);

  const uint32_t n_lat = uni(ev->n_lat);
  const uint32_t nb = uni(ev->n_hist);
  if (n_lat == 0 || nb == 0) {  // train_infos: empty latents -> TrainedBins::default()
    if (tid == 0) { ev->n_bins = 0; ev->ans_size_log = 0; ev->max_ob = 0; ev->is_trivial = 1; ev->needs_ans = 1; plan.next_states()[0] = 1; }
    return;
  }
  const uint32_t ubl = uni(ch->unopt_bins_log);
  const uint32_t bins_log = var == 2 ? (ubl < 6 ? ubl : 6) : ubl;
  const uint32_t n_log_ceil = n_lat <= 1 ? 0 : (32 - clz_u32(n_lat - 1));
  uint32_t est = bins_log + 2; if (est > 12) est = 12; if (est > n_log_ceil) est = n_log_ceil;  // estimated_ans_size_log
  // load histogram bins
  block_sync();
  for (uint32_t b = tid; b < nb; b += 64 * NW) { lows[b] = (L)plan.hlower()[b]; ups[b] = (L)plan.hupper()[b]; cc[b + 1] = plan.hcount()[b]; }
  if (tid == 0) { cc[0] = 0; best[0] = 0.0f; }
  block_sync();
  if (tid == 0) { uint32_t c = 0; for (uint32_t b = 0; b < nb; b++) { c += cc[b + 1]; cc[b + 1] = c; } }
  block_sync();
  const uint32_t total_count = cc[nb];
  const float total_log2 = log2_approx_dev((float)total_count);
  const float meta = (float)(est + LBits<L>::v + offset_bits_bits(LBits<L>::v));
  // ---- DP (bin_optimization.rs:104-178): best[i+1] = min_j best[j] + cost(j..i); ties -> largest j ----
  if constexpr (NW == 1 && CAP == 256) {
    // One wave, at most 256 bins: lane t OWNS the candidates j = t, t + 64, t + 128, t + 192 -- their lower bound, cumulative count and
    // best[j] live in its registers (best[j] arrives when step j - 1 ends: every lane knows the step's minimum, the owner keeps it), so a
    // step reads nothing from LDS but its own upper bound and count, which were fetched during the step before, and waits for no store.
    L low_j[4]; uint32_t cc_j[4]; float best_j[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) { const uint32_t j = tid + 64 * q; low_j[q] = j < nb ? lows[j] : (L)0; cc_j[q] = j < nb ? cc[j] : 0u; best_j[q] = 0.0f; }   // (best[0] = 0; the others are filled in below)
    L upper = ups[0]; uint32_t cci = cc[1];
    for (uint32_t i = 0; i < nb; i++) {
      const L upper_i = upper; const uint32_t cci_i = cci;
      if (i + 1 < nb) { upper = ups[i + 1]; cci = cc[i + 2]; }
      float bc = 3.402823466e+38f; uint32_t bjv = 0xffffffffu;
      const uint32_t nq = (i >> 6) + 1;   // (uniform) slots that hold a candidate j <= i in some lane
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
        if (q >= nq) break;
        const uint32_t j = tid + 64 * q;
        const float cost = __fadd_rn(best_j[q], bin_cost_dev<L>(meta, low_j[q], upper_i, cci_i - cc_j[q], total_log2));
        const bool take = j <= i && cost <= bc;   // (j ascends with q: on equal cost the larger j wins, as the reference's descending scan with '<' keeps it)
        bc = take ? cost : bc; bjv = take ? j : bjv;
      }
      {
        const uint32_t fb = __float_as_uint(bc);
        const uint32_t cbits = fb ^ (((uint32_t)((int32_t)fb >> 31)) | 0x80000000u);
        const uint32_t cmin = wave_reduce_u32(cbits, [](uint32_t p, uint32_t q) { return p < q ? p : q; });
        const uint32_t jc = bjv != 0xffffffffu && cbits == cmin ? bjv + 1u : 0u;
        const uint32_t jmax = wave_reduce_u32(jc, [](uint32_t p, uint32_t q) { return p > q ? p : q; });
        bc = __uint_as_float(cmin ^ ((cmin >> 31) - 1u | 0x80000000u)); bjv = jmax - 1u;
      }
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) best_j[q] = tid + 64 * q == i + 1 ? bc : best_j[q];
      if (lane == 0) { best[i + 1] = bc; bj[i] = (uint16_t)bjv; }   // (for the rewind below, which a wave sync precedes)
    }
    enc_wave_sync();
  } else
  for (uint32_t i = 0; i < nb; i++) {
    const L upper = ups[i]; const uint32_t cci = cc[i + 1];
    float bc = 3.402823466e+38f; uint32_t bjv = 0xffffffffu;
    for (int32_t j = (int32_t)i - (int32_t)tid; j >= 0; j -= 64 * NW) {
      const float cost = __fadd_rn(best[j], bin_cost_dev<L>(meta, lows[j], upper, cci - cc[j], total_log2));
      if (cost < bc) { bc = cost; bjv = (uint32_t)j; }
    }
    {   // argmin over the wave, ties -> largest j: the smallest cost by a DPP reduction over the floats' order-preserving bit patterns
        // (lanes without a candidate hold FLT_MAX; costs are positive, never NaN), then the largest j among the lanes that hold it
      const uint32_t fb = __float_as_uint(bc);
      const uint32_t cbits = fb ^ (((uint32_t)((int32_t)fb >> 31)) | 0x80000000u);
      const uint32_t cmin = wave_reduce_u32(cbits, [](uint32_t p, uint32_t q) { return p < q ? p : q; });
      const uint32_t jc = bjv != 0xffffffffu && cbits == cmin ? bjv + 1u : 0u;
      const uint32_t jmax = wave_reduce_u32(jc, [](uint32_t p, uint32_t q) { return p > q ? p : q; });
      bc = __uint_as_float(cmin ^ ((cmin >> 31) - 1u | 0x80000000u)); bjv = jmax - 1u;   // (jmax == 0: no lane had a candidate -- bjv = 0xffffffff as before)
    }
    if (NW > 1) {   // across the block's waves (only those that had a j to look at)
      const uint32_t n_act = (i >> 6) + 1 < NW ? (i >> 6) + 1 : NW;
      if (lane == 0 && wave < n_act) { red_c[wave] = bc; red_j[wave] = bjv; }
      __syncthreads();
      if (tid == 0) {
        for (uint32_t w = 1; w < n_act; w++) {
          const float oc = red_c[w]; const uint32_t oj = red_j[w];
          const bool take = oj != 0xffffffffu && (bjv == 0xffffffffu || oc < bc || (oc == bc && oj > bjv));
          if (take) { bc = oc; bjv = oj; }
        }
        best[i + 1] = bc; bj[i] = (uint16_t)bjv;
      }
      __syncthreads();
    } else {
      enc_wave_sync();
      if (lane == 0) { best[i + 1] = bc; bj[i] = (uint16_t)bjv; }
      enc_wave_sync();
    }
  }
  // ---- partition choice + optimized bins + quantisation: sequential (thread 0) ----
  uint32_t n_opt = 0, ans_size_log = 0;
  if (tid == 0) {
    const float best_cost = best[nb];
    const float bias = __fmul_rn(0.1f, (float)total_count);
    const float thr = __fadd_rn(best_cost, bias);
    const float single = bin_cost_dev<L>(meta, lows[0], ups[nb - 1], total_count, total_log2);
    bool done = false;
    if (single < thr) { part[0] = 0; part[1] = (uint16_t)(nb - 1); n_opt = 1; done = true; }
    if (!done) {
      bool all_trivial = true;
      for (uint32_t b = 0; b < nb; b++) if (lows[b] != ups[b]) { all_trivial = false; break; }
      if (all_trivial) {
        float cost = 0.0f;
        for (uint32_t b = 0; b < nb; b++) cost = __fadd_rn(cost, bin_cost_dev<L>(meta, lows[b], ups[b], cc[b + 1] - cc[b], total_log2));
        if (cost < thr) { for (uint32_t b = 0; b < nb; b++) { part[2 * b] = (uint16_t)b; part[2 * b + 1] = (uint16_t)b; } n_opt = nb; done = true; }
      }
    }
    if (!done) {  // rewind_best_partitioning (built back to front, then reversed in place)
      uint32_t cnt = 0; uint32_t i = nb - 1;
      for (;;) { const uint32_t j = bj[i]; part[2 * cnt] = (uint16_t)j; part[2 * cnt + 1] = (uint16_t)i; cnt++; if (j > 0) i = j - 1; else break; }
      for (uint32_t a = 0; a < cnt / 2; a++) {
        const uint16_t j0 = part[2 * a], i0 = part[2 * a + 1];
        part[2 * a] = part[2 * (cnt - 1 - a)]; part[2 * a + 1] = part[2 * (cnt - 1 - a) + 1];
        part[2 * (cnt - 1 - a)] = j0; part[2 * (cnt - 1 - a) + 1] = i0;
      }
      n_opt = cnt;
    }
    // optimized bins (bin_optimization.rs:180-198)
    uint32_t max_ob = 0;
    for (uint32_t s = 0; s < n_opt; s++) {
      const uint32_t j = part[2 * s], i = part[2 * s + 1];
      const uint32_t count = cc[i + 1] - cc[j];
      const L lower = lows[j], upper = ups[i];
      const uint32_t ob = bitlen<L>((L)(upper - lower));
      plan.bcount()[s] = count; plan.blower()[s] = (uint64_t)lower; plan.bob()[s] = (uint8_t)ob;
      wts[s] = count; max_ob = max_ob > ob ? max_ob : ob;
    }
    // quantize_weights (ans/encoding.rs:95-175)
    if (n_opt == 1) { ans_size_log = 0; wts[0] = 1; }
    else {
      const uint32_t min_size_log = 32 - clz_u32(n_opt - 1);
      uint32_t size_log = min_size_log > est ? min_size_log : est;
      const uint32_t required = 1u << size_log;
      const float multiplier = __fdiv_rn((float)required, (float)n_lat);
      float desired_surplus = 0.0f;
      for (uint32_t s = 0; s < n_opt; s++) {
        float v = __fsub_rn(__fmul_rn((float)wts[s], multiplier), 1.0f);
        v = v > 0.0f ? v : 0.0f;
        fw[s] = v; desired_surplus = __fadd_rn(desired_surplus, v);
      }
      const uint32_t required_surplus = required - n_opt;
      const float surplus_mult = desired_surplus == 0.0f ? 0.0f : __fdiv_rn((float)required_surplus, desired_surplus);
      uint32_t weight_sum = 0;
      for (uint32_t s = 0; s < n_opt; s++) {
        const float f = __fadd_rn(1.0f, __fmul_rn(fw[s], surplus_mult));
        fw[s] = f;
        const float r = roundf(f);
        uint32_t w = r <= 0.0f ? 0u : (r >= 4294967296.0f ? 0xffffffffu : (uint32_t)r);
        wts[s] = w; weight_sum += w;
      }
      uint32_t i = 0;
      while (weight_sum > required && i < n_opt) { if (wts[i] > 1 && (float)wts[i] > fw[i]) { wts[i]--; weight_sum--; } i++; }
      i = 0;
      while (weight_sum < required && i < n_opt) { if ((float)wts[i] < fw[i]) { wts[i]++; weight_sum++; } i++; }
      if (weight_sum != required) ch->status = PCO_GFX_INVALID_ARGUMENT;  // the reference would panic (index out of bounds)
      uint32_t pow2 = 32;
      for (uint32_t s = 0; s < n_opt; s++) { const uint32_t tz = wts[s] == 0 ? 32u : (uint32_t)__builtin_ctz(wts[s]); pow2 = pow2 < tz ? pow2 : tz; }
      size_log -= pow2;
      for (uint32_t s = 0; s < n_opt; s++) wts[s] >>= pow2;
      ans_size_log = size_log;
    }
    uint32_t c = 0;
    for (uint32_t s = 0; s < n_opt; s++) { plan.bweight()[s] = wts[s]; cum[s] = c; c += wts[s]; }
    cum[n_opt] = c;
    ev->n_bins = n_opt; ev->ans_size_log = ans_size_log; ev->max_ob = max_ob;
    ev->is_trivial = (n_opt == 1 && plan.bob()[0] == 0) ? 1u : 0u;
    ev->needs_ans = n_opt != 1 ? 1u : 0u;
  }
  __threadfence_block();
  block_sync();
  n_opt = uni(ev->n_bins); ans_size_log = uni(ev->ans_size_log);
  // ---- tANS encoder tables (ans/spec.rs:37-59, ans/encoding.rs:28-63) ----
  const uint32_t T = 1u << ans_size_log;
  uint32_t stride = (3 * T) / 5; if ((stride & 1) == 0) stride += 1;
  for (uint32_t tt = tid; tt < T; tt += 64 * NW) {
    uint32_t lo = 0, hi = n_opt;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (cum[mid] <= tt) lo = mid; else hi = mid; }
    ssym[(stride * tt) & (T - 1)] = (uint16_t)lo;
  }
  block_sync();
  for (uint32_t s = tid; s < n_opt; s += 64 * NW) {
    const uint32_t w = wts[s];
    const uint32_t max_x_s = 2 * w - 1;
    const uint32_t min_renorm_bits = ans_size_log - (31 - clz_u32(max_x_s));
    const uint32_t cutoff = (2 * w) << min_renorm_bits;
    const uint32_t adj = cum[s] - w + 8192u;  // next_states index = adj - 8192 + (state >> bits)
    plan.syminfo()[s] = cutoff | (min_renorm_bits << 14) | (adj << 18);
  }
  // next_states: states of symbol s in ascending state order -> T + state_idx; fill counters reuse wts[] (set to cum)
  block_sync();
  for (uint32_t s = tid; s < n_opt; s += 64 * NW) wts[s] = cum[s];
  block_sync();
  if (wave == 0) {   // (state order matters: one wave walks the table)
    const uint32_t sym_bits = 32 - clz_u32(n_opt - 1 > 0 ? n_opt - 1 : 1);
    for (uint32_t i0 = 0; i0 < T; i0 += 64) {
      const uint32_t i = i0 + lane;
      const bool act = i < T;
      const uint32_t s = act ? (uint32_t)ssym[i] : 0xffffffffu;
      uint64_t m = __ballot(act);
      for (uint32_t bit = 0; bit < sym_bits; bit++) { const uint64_t bm = __ballot((s >> bit) & 1); m &= ((s >> bit) & 1) ? bm : ~bm; }
      const uint64_t lt = ((uint64_t)1 << lane) - 1;
      const uint32_t rank = __popcll(m & lt), gcount = __popcll(m);
      const uint32_t basec = act ? wts[s] : 0;
      enc_wave_sync();
      if (act && rank == 0) wts[s] = basec + gcount;
      enc_wave_sync();
      if (act) plan.next_states()[basec + rank] = (uint16_t)(T + i);
    }
  }
  block_sync();
}

// should_fallback (wrapped/chunk_compressor.rs:502-541) and the choice of page encoder, by one thread once every variable is trained
__device__ void train_finish(const EncWorkspace& ws, uint32_t t) {
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  const int bits = dtype_bits(ch->dtype);
  const uint32_t mode_kind = ch->mode_kind, delta_kind = ch->delta_kind;
  uint32_t fallback = 0;
  if (!(delta_kind == kDeltaNone && mode_kind == kClassic)) {
    const uint64_t n = ch->n;
    const uint64_t n_pages = ch->n_pages;
    uint64_t worst_bits = 7 * n_pages;
    uint64_t meta_bits = kBitsModeVariant + (mode_kind == kIntMult || mode_kind == kFloatMult ? (uint64_t)bits : (mode_kind == kFloatQuant ? kBitsQuantK : 0));
    meta_bits += 4 + 5 + 5 + 64 + 32 * 32;  // DeltaEncoding::MAX_BIT_SIZE
    uint64_t page_meta_bits = 0;
    for (uint32_t var = 0; var < 3; var++) {
      if (!ch->v[var].present) continue;
      const PlanRef plan = plan_ref(ws, t, var);
      const uint32_t lb = ch->v[var].latent_bits, asl = ch->v[var].ans_size_log, nbv = ch->v[var].n_bins;
      for (uint32_t s = 0; s < nbv; s++)
        worst_bits += (uint64_t)plan.bcount()[s] * (uint64_t)(plan.bob()[s] + asl - (31 - clz_u32(plan.bweight()[s])));
      meta_bits += kBitsAnsSizeLog + kBitsNBins + (uint64_t)nbv * (asl + lb + offset_bits_bits(lb));
      uint32_t nlps = 0;
      if (var == 1) nlps = delta_kind == kDeltaConsecutive ? ch->delta_order : (delta_kind == kDeltaLookback ? (1u << ch->state_n_log) : 0u);
      page_meta_bits += (uint64_t)asl * 4 + (uint64_t)lb * nlps;
    }
    const uint64_t worst = (meta_bits + 7) / 8 + n_pages * ((page_meta_bits + 7) / 8) + (worst_bits + 7) / 8;
    const uint64_t base_meta_bits = kBitsModeVariant + (4 + 5 + 5 + 64 + 32 * 32) + kBitsAnsSizeLog + kBitsNBins + (uint64_t)bits + offset_bits_bits(bits);
    const uint64_t baseline = (base_meta_bits + 7) / 8 + (n * (uint64_t)bits + 7) / 8;
    fallback = worst > baseline ? 1u : 0u;
  }
  ch->fallback = fallback;
  uint32_t fast_ok = fallback ? 0u : 1u;
  for (uint32_t var = 0; var < 3; var++) if (ch->v[var].present && (ch->v[var].ans_size_log > kFastEncMaxAsl || ch->v[var].n_bins > kMaxBins)) fast_ok = 0;
  ch->fast_ok = fast_ok;
}

__global__ __launch_bounds__(64) void enc_train_kernel(EncWorkspace ws, uint32_t n_tasks) {
  const uint32_t t = blockIdx.x;
  if (t >= n_tasks) return;
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  if (uni(ch->status) != PCO_GFX_OK || uni(ch->big) != 0) return;   // (more than 256 bins: enc_train_big_kernel)
  const int bits = dtype_bits(uni(ch->dtype));
  for (uint32_t var = 0; var < 3; var++) {
    if (!uni(ch->v[var].present)) continue;
    if (var == 0) train_var<uint32_t, kMaxBins, 1>(ws, t, var);
    else if (bits == 64) train_var<uint64_t, kMaxBins, 1>(ws, t, var);
    else if (bits == 32) train_var<uint32_t, kMaxBins, 1>(ws, t, var);
    else if (bits == 16) train_var<uint16_t, kMaxBins, 1>(ws, t, var);
    else train_var<uint8_t, kMaxBins, 1>(ws, t, var);
    __threadfence_block();
    enc_wave_sync();
  }
  if (lane_id() == 0) train_finish(ws, t);
}
// chunks with more than 256 histogram bins (compression levels 9..12): the O(bins^2) DP over up to 4096 bins by a block of 1024 threads
__global__ __launch_bounds__(1024) void enc_train_big_kernel(EncWorkspace ws, uint32_t n_tasks) {
  const uint32_t t = blockIdx.x;
  if (t >= n_tasks) return;
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  if (uni(ch->status) != PCO_GFX_OK || uni(ch->big) == 0) return;
  const int bits = dtype_bits(uni(ch->dtype));
  for (uint32_t var = 0; var < 3; var++) {
    if (!uni(ch->v[var].present)) continue;
    if (var == 0) train_var<uint32_t, kBigBins, 16>(ws, t, var);
    else if (bits == 64) train_var<uint64_t, kBigBins, 16>(ws, t, var);
    else if (bits == 32) train_var<uint32_t, kBigBins, 16>(ws, t, var);
    else if (bits == 16) train_var<uint16_t, kBigBins, 16>(ws, t, var);
    else train_var<uint8_t, kBigBins, 16>(ws, t, var);
    __threadfence_block();
    __syncthreads();
  }
  if (threadIdx.x == 0) train_finish(ws, t);
}

// =========================================================================================================
// K4: dissect + reverse tANS + bit packing + metadata, one wave per chunk
// =========================================================================================================
constexpr uint32_t kStgDwords = 704;                       // staging for one section (<= 2048 B offsets + slack)
constexpr uint32_t kPageLdsStg = 0;                        // u32[704]
constexpr uint32_t kPageLdsSym = 2816;                     // u32[256] dissect words of the batch
constexpr uint32_t kPageLdsVar = kPageLdsSym + 1024;       // per-var tables follow
// per-variable table area for a bin capacity `cap` (256, or the variable's bin count rounded up to a power of two when it has more:
// levels 9..12): search lowers (8-byte stride) [cap] | offset bits u8[cap] | syminfo u32[cap] | next states u16[2^ans_size_log]
constexpr uint32_t kPageVarLow = 0;
__host__ __device__ constexpr uint32_t page_var_ob(uint32_t cap) { return cap * 8; }
__host__ __device__ constexpr uint32_t page_var_info(uint32_t cap) { return cap * 9; }
__host__ __device__ constexpr uint32_t page_var_ns(uint32_t cap) { return cap * 13; }
__host__ __device__ constexpr uint32_t page_var_bytes(uint32_t table_log, uint32_t cap = 256) { return cap * 13 + (2u << table_log); }
__device__ __forceinline__ uint32_t page_var_cap(uint32_t n_bins) { uint32_t c = 256; while (c < n_bins) c <<= 1; return c; }

__device__ __forceinline__ void store_result(PcoGfxTaskResult PCO_GLOBAL* p, uint64_t n_out, uint32_t status, uint32_t aux) {
  p->n_out = n_out; p->consumed = 0; p->status = status; p->aux = aux;
}

// Streaming bit sink: LSB-first fields into dst (bit_writer.rs:22-42 semantics), staged in LDS, flushed as dwords.
struct BitSink {
  uint32_t PCO_LDS* stg; uint32_t PCO_GLOBAL* dst; uint64_t dst_cap_bits; uint64_t outbit; uint32_t overflow;
  __device__ __forceinline__ void init(uint32_t PCO_LDS* s, uint32_t PCO_GLOBAL* d, uint64_t cap_bytes) {
    stg = s; dst = d; dst_cap_bits = cap_bytes * 8; outbit = 0; overflow = 0;
    for (uint32_t i = lane_id(); i < kStgDwords; i += 64) stg[i] = 0;
    enc_wave_sync();
  }
  // OR `nbits` (<= 64) of `val` at bit offset `rel` (relative to outbit)
  __device__ __forceinline__ void put(uint32_t rel, uint64_t val, uint32_t nbits) {
    if (nbits == 0) return;
    if (nbits < 64) val &= ((uint64_t)1 << nbits) - 1;
    const uint32_t pos = (uint32_t)(outbit & 31) + rel;
    const uint32_t dw = pos >> 5, sh = pos & 31;
    atomicOr((uint32_t*)&stg[dw], (uint32_t)(val << sh));
    if (sh + nbits > 32) {
      const uint64_t rest = sh ? (val >> (32 - sh)) : (val >> 32);
      atomicOr((uint32_t*)&stg[dw + 1], (uint32_t)rest);
      if (sh + nbits > 64) atomicOr((uint32_t*)&stg[dw + 2], (uint32_t)(rest >> 32));
    }
  }
  // all lanes: finish a section of `total` bits
  __device__ __forceinline__ void advance(uint32_t total) {
    enc_wave_sync();
    const uint64_t newbit = outbit + total;
    if (newbit + 64 > dst_cap_bits) { overflow = 1; }
    const uint64_t base_dw = outbit >> 5;
    const uint32_t ncomplete = (uint32_t)((newbit >> 5) - base_dw);
    const uint32_t lane = lane_id();
    uint32_t last = 0;
    if (!overflow) for (uint32_t i = lane; i < ncomplete; i += 64) dst[base_dw + i] = stg[i];
    last = stg[ncomplete];
    enc_wave_sync();
    const uint32_t used = ncomplete + 1;
    for (uint32_t i = lane; i < used && i < kStgDwords; i += 64) stg[i] = 0;
    enc_wave_sync();
    if (lane == 0) stg[0] = last;
    enc_wave_sync();
    outbit = newbit;
  }
  __device__ __forceinline__ void finish_byte() { const uint32_t pad = (uint32_t)((8 - (outbit & 7)) & 7); advance(pad); }
  // write the trailing partial dword; returns total bytes
  __device__ __forceinline__ uint64_t close() {
    enc_wave_sync();
    if (!overflow && lane_id() == 0 && (outbit & 31)) dst[outbit >> 5] = stg[0];
    return (outbit + 7) >> 3;
  }
  // uniform single field written by lane 0
  __device__ __forceinline__ void put_uniform(uint64_t val, uint32_t nbits) { if (lane_id() == 0) put(0, val, nbits); advance(nbits); }
};

template <class LV>
__device__ __forceinline__ void page_load_var_tables(uint8_t PCO_LDS* vt, const PlanRef& plan, uint32_t n_bins, uint32_t asl) {
  const uint32_t lane = lane_id(), cap = page_var_cap(n_bins);
  LV PCO_LDS* low = (LV PCO_LDS*)(vt + kPageVarLow);
  uint8_t PCO_LDS* ob = vt + page_var_ob(cap);
  uint32_t PCO_LDS* info = (uint32_t PCO_LDS*)(vt + page_var_info(cap));
  uint16_t PCO_LDS* ns = (uint16_t PCO_LDS*)(vt + page_var_ns(cap));
  uint32_t padded = 1; while (padded < n_bins) padded <<= 1;
  for (uint32_t b = lane; b < padded; b += 64) {
    low[b] = b < n_bins ? (LV)plan.blower()[b] : (LV)~(LV)0;   // padded with L::MAX (compression_table.rs:22-26)
    ob[b] = b < n_bins ? plan.bob()[b] : 0;
    info[b] = b < n_bins ? plan.syminfo()[b] : 0;
  }
  const uint32_t T = 1u << asl;
  for (uint32_t i = lane; i < T; i += 64) ns[i] = plan.next_states()[i];
}

// Phase A1 for one variable: batches in reverse; binary search -> symbol; reverse tANS over 4 chains.
template <class LV>
__device__ void page_dissect_var(const uint8_t PCO_LDS* vt, const LV PCO_GLOBAL* lat, uint32_t PCO_GLOBAL* dis, uint32_t n_lat,
                                 uint32_t n_bins, uint32_t asl, uint32_t final_states[4]) {
  const uint32_t lane = lane_id(), cap = page_var_cap(n_bins);
  const LV PCO_LDS* low = (const LV PCO_LDS*)(vt + kPageVarLow);
  const uint32_t PCO_LDS* info = (const uint32_t PCO_LDS*)(vt + page_var_info(cap));
  const uint16_t PCO_LDS* ns = (const uint16_t PCO_LDS*)(vt + page_var_ns(cap));
  uint32_t PCO_LDS* words = (uint32_t PCO_LDS*)(enc_lds_base() + kPageLdsSym);
  uint32_t search_log = 0; while ((1u << search_log) < n_bins) search_log++;
  uint32_t state = 1u << asl;  // lanes 0..3: chain states (ans/encoding.rs:89-91)
  const uint32_t n_batches = (n_lat + kBatchN - 1) / kBatchN;
  for (uint32_t b = n_batches; b-- > 0;) {
    const uint32_t base = b * kBatchN;
    const uint32_t cnt = n_lat - base < kBatchN ? n_lat - base : kBatchN;
    // symbols: branch-free lower bound over padded search lowers (compression_table.rs:51-74)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t i = 4 * lane + k;
      uint32_t sym = 0;
      if (i < cnt) {
        const LV x = lat[base + i];
        for (uint32_t depth = 0; depth < search_log; depth++) {
          const uint32_t bis = 1u << (search_log - 1 - depth);
          if (x >= low[sym + bis]) sym += bis;
        }
        sym = sym < n_bins - 1 ? sym : n_bins - 1;
      }
      words[i] = sym << 16;
    }
    enc_wave_sync();
    if (asl != 0 && lane < 4) {  // encode_ans_in_reverse (chunk_latent_compressor.rs:96-132)
      const uint32_t steps = (cnt + 3) >> 2;
      for (uint32_t g = steps; g-- > 0;) {
        const uint32_t i = 4 * g + lane;
        if (i < cnt) {
          const uint32_t w = words[i];
          const uint32_t si = info[w >> 16];
          const uint32_t cutoff = si & 0x3fffu, minb = (si >> 14) & 15u, adj = si >> 18;
          const uint32_t bits = minb + (state >= cutoff ? 1u : 0u);
          const uint32_t val = state & ((1u << bits) - 1u);
          words[i] = w | (bits << 12) | val;
          state = ns[adj - 8192u + (state >> bits)];
        }
      }
    }
    enc_wave_sync();
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t i = 4 * lane + k; if (i < cnt) dis[base + i] = words[i]; }
    enc_wave_sync();
  }
  for (uint32_t j = 0; j < 4; j++) final_states[j] = __shfl(state, j, 64);
}

// Phase A2 for one variable's batch: pack 256 tANS fields then 256 offset fields.
template <class LV>
__device__ __forceinline__ void page_pack_batch(BitSink& sink, const uint8_t PCO_LDS* vt, const LV PCO_GLOBAL* lat, const uint32_t PCO_GLOBAL* dis,
                                                uint32_t base, uint32_t cnt, bool needs_ans, uint32_t max_ob, bool single_bin, uint32_t cap) {
  const uint32_t lane = lane_id();
  const LV PCO_LDS* low = (const LV PCO_LDS*)(vt + kPageVarLow);
  const uint8_t PCO_LDS* obs = vt + page_var_ob(cap);
  uint32_t w[4]; LV x[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t i = 4 * lane + k;
    w[k] = (i < cnt && !single_bin) ? dis[base + i] : 0u;
    x[k] = i < cnt ? lat[base + i] : (LV)0;
  }
  if (needs_ans) {
    uint32_t t = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) t += (4 * lane + k < cnt) ? ((w[k] >> 12) & 15u) : 0u;
    const uint32_t incl = wave_incl_scan(t);
    const uint32_t total = wave_last(incl);
    uint32_t rel = incl - t;
    uint64_t acc = 0; uint32_t accbits = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (4 * lane + k < cnt) { const uint32_t nb = (w[k] >> 12) & 15u; acc |= (uint64_t)(w[k] & 0xfffu) << accbits; accbits += nb; }
    }
    sink.put(rel, acc, accbits);  // <= 48 bits
    sink.advance(total);
  }
  if (max_ob != 0) {
    uint32_t ob[4]; uint32_t t = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t i = 4 * lane + k;
      const uint32_t sym = w[k] >> 16;
      ob[k] = i < cnt ? (uint32_t)obs[sym] : 0u;
      x[k] = (LV)(x[k] - low[sym]);
      t += ob[k];
    }
    const uint32_t incl = wave_incl_scan(t);
    const uint32_t total = wave_last(incl);
    uint32_t rel = incl - t;
#pragma unroll
    for (int k = 0; k < 4; k++) { sink.put(rel, (uint64_t)x[k], ob[k]); rel += ob[k]; }
    sink.advance(total);
  }
}

// ChunkMeta (metadata/chunk.rs:176-189, mode.rs:169-195, delta_encoding.rs:204-254, chunk_latent_var.rs:55-71,158-168)
template <class L>
__device__ void page_write_chunk_meta(BitSink& sink, const EncWorkspace& ws, uint32_t t) {
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  const uint32_t lane = lane_id();
  constexpr uint32_t LB = LBits<L>::v;
  if (uni(ch->fallback)) {
    // fallback_chunk_compressor (wrapped/chunk_compressor.rs:396-438): Classic, NoOp, one bin {w 1, lower 0, offset_bits L::BITS}
    sink.put_uniform(0, kBitsModeVariant); sink.put_uniform(0, kBitsDeltaVariant);
    sink.put_uniform(0, kBitsAnsSizeLog); sink.put_uniform(1, kBitsNBins);
    sink.put_uniform(0, LB); sink.put_uniform(LB, offset_bits_bits(LB));
    sink.finish_byte();
    return;
  }
  const uint32_t mode_kind = uni(ch->mode_kind), delta_kind = uni(ch->delta_kind), delta_order = uni(ch->delta_order);
  sink.put_uniform(mode_kind, kBitsModeVariant);
  if (mode_kind == kIntMult || mode_kind == kFloatMult) sink.put_uniform(uni((uint64_t)ch->mode_base), LB);
  else if (mode_kind == kFloatQuant) sink.put_uniform(uni(ch->mode_k), kBitsQuantK);
  sink.put_uniform(delta_kind, kBitsDeltaVariant);
  if (delta_kind == kDeltaConsecutive) { sink.put_uniform(delta_order, kBitsDeltaOrder); sink.put_uniform(0, 1); }
  else if (delta_kind == kDeltaLookback) { sink.put_uniform(uni(ch->window_n_log) - 1, kBitsLookbackWindowLog); sink.put_uniform(uni(ch->state_n_log), kBitsLookbackStateLog); sink.put_uniform(0, 1); }
#pragma unroll
  for (int v = 0; v < 3; v++) {
    if (!uni(ch->v[v].present)) continue;
    const PlanRef plan = plan_ref(ws, t, v);
    const uint32_t asl = uni(ch->v[v].ans_size_log), nbins = uni(ch->v[v].n_bins);
    const uint32_t lb = v == 0 ? 32u : LB, obb = offset_bits_bits(lb);
    sink.put_uniform(asl, kBitsAnsSizeLog); sink.put_uniform(nbins, kBitsNBins);
    const uint32_t bin_bits = asl + lb + obb;
    for (uint32_t b0 = 0; b0 < nbins; b0 += 64) {
      const uint32_t b = b0 + lane;
      const uint32_t nb = nbins - b0 < 64 ? nbins - b0 : 64;
      if (b < nbins) {
        const uint32_t rel = lane * bin_bits;
        sink.put(rel, plan.bweight()[b] - 1, asl);
        sink.put(rel + asl, plan.blower()[b], lb);
        sink.put(rel + asl + lb, plan.bob()[b], obb);
      }
      sink.advance(nb * bin_bits);
    }
  }
  sink.finish_byte();
}

// One page task: [standalone preamble + ChunkMeta] + page meta + page body (wrapped/chunk_compressor.rs:659-705).
template <class L>
__device__ void page_task(const EncWorkspace& ws, const PcoGfxEncodeTask& task, EncPage PCO_GLOBAL* pg, PcoGfxTaskResult PCO_GLOBAL* result) {
  const uint32_t t = uni(pg->chunk);
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  const uint32_t lane = lane_id();
  uint8_t PCO_LDS* smem = enc_lds_base();
  BitSink sink;
  sink.init((uint32_t PCO_LDS*)(smem + kPageLdsStg), (uint32_t PCO_GLOBAL*)pg->dst, uni((uint64_t)pg->dst_cap));
  const uint32_t pflags = uni(pg->flags);
  const uint32_t page_n = (uint32_t)uni((uint64_t)pg->n);
  const uint64_t pstart = uni((uint64_t)pg->start);
  const uint32_t dtype = uni(ch->dtype);
  constexpr uint32_t LB = LBits<L>::v;
  const uint32_t fallback = uni(ch->fallback);
  if (pflags & kPageFlagPreamble) {  // standalone/compressor.rs:191-203
    sink.put_uniform(dtype, 8);
    sink.put_uniform(page_n - 1, kBitsNEntries);
  }
  if (pflags & (kPageFlagPreamble | kPageFlagMetaOnly)) page_write_chunk_meta<L>(sink, ws, t);
  if (pflags & kPageFlagMetaOnly) {
    const uint64_t bytes = sink.close();
    if (lane == 0) store_result(result, bytes, sink.overflow ? PCO_GFX_INVALID_ARGUMENT : PCO_GFX_OK, fallback);
    return;
  }
  if (fallback) {
    // page meta: 4 x 0-bit states -> nothing; body: the page's raw ordered latents, L::BITS each
    const L PCO_GLOBAL* src = (const L PCO_GLOBAL*)task.src + pstart;
    const uint32_t num_kind = dtype_kind(dtype);
    for (uint32_t base = 0; base < page_n; base += kBatchN) {
      const uint32_t cnt = page_n - base < kBatchN ? page_n - base : kBatchN;
      uint32_t rel = 4 * lane * LB;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t i = 4 * lane + k;
        if (i < cnt) sink.put(rel, (uint64_t)to_latent_ordered<L>(src[base + i], num_kind), LB);
        rel += LB;
      }
      sink.advance(cnt * LB);
    }
    sink.finish_byte();
    const uint64_t bytes = sink.close();
    if (lane == 0) store_result(result, bytes, sink.overflow ? PCO_GFX_INVALID_ARGUMENT : PCO_GFX_OK, 1);
    return;
  }
  // ---- per variable parameters ----
  const uint32_t delta_kind = uni(ch->delta_kind), delta_order = uni(ch->delta_order);
  uint32_t present[3], n_bins[3], asl[3], max_ob[3], n_lat[3], skip[3], needs_ans[3], trivial[3], voff[3];
  uint32_t off = kPageLdsVar;
#pragma unroll
  for (int v = 0; v < 3; v++) {
    present[v] = uni(ch->v[v].present); n_bins[v] = uni(ch->v[v].n_bins); asl[v] = uni(ch->v[v].ans_size_log); max_ob[v] = uni(ch->v[v].max_ob);
    needs_ans[v] = uni(ch->v[v].needs_ans); trivial[v] = uni(ch->v[v].is_trivial);
    skip[v] = v == 2 ? 0u : uni(ch->v[v].lat_start);                 // the page's junk prefix (delta state lives in the page meta)
    if (skip[v] > page_n) skip[v] = page_n;
    n_lat[v] = page_n - skip[v];
    voff[v] = off; if (present[v]) off += page_var_bytes(asl[v], page_var_cap(n_bins[v]));
  }
  // ---- load tables, phase A1 (reverse dissect) ----
  uint32_t fs[3][4];
#pragma unroll
  for (int v = 0; v < 3; v++) {
    for (int j = 0; j < 4; j++) fs[v][j] = 1u << asl[v];
    if (!present[v] || trivial[v]) continue;
    const PlanRef plan = plan_ref(ws, t, v);
    if (v == 0) page_load_var_tables<uint32_t>(smem + voff[v], plan, n_bins[v], asl[v]); else page_load_var_tables<L>(smem + voff[v], plan, n_bins[v], asl[v]);
  }
  enc_wave_sync();
#pragma unroll
  for (int v = 0; v < 3; v++) {
    if (!present[v] || trivial[v] || n_bins[v] <= 1) continue;
    if (v == 0) page_dissect_var<uint32_t>(smem + voff[v], lat_ptr<uint32_t>(ws, t, 0) + pstart + skip[v], dissect_ptr(ws, t, 0) + pstart + skip[v], n_lat[v], n_bins[v], asl[v], fs[v]);
    else page_dissect_var<L>(smem + voff[v], lat_ptr<L>(ws, t, v) + pstart + skip[v], dissect_ptr(ws, t, v) + pstart + skip[v], n_lat[v], n_bins[v], asl[v], fs[v]);
  }
  __threadfence_block();
  enc_wave_sync();
  // ---- page meta (metadata/page.rs:22-34, page_latent_var.rs:19-26) ----
#pragma unroll
  for (int v = 0; v < 3; v++) {
    if (!present[v]) continue;
    if (v == 1) {
      const uint32_t nlps = delta_kind == kDeltaConsecutive ? delta_order : (delta_kind == kDeltaLookback ? (1u << uni(ch->state_n_log)) : 0u);
      for (uint32_t i = 0; i < nlps; i++) sink.put_uniform(uni((uint64_t)pg->moments[i]), LB);
    }
    for (int j = 0; j < 4; j++) sink.put_uniform(fs[v][j] - (1u << asl[v]), asl[v]);
  }
  sink.finish_byte();
  // ---- phase A2: pack batches forward (wrapped/chunk_compressor.rs:624-651) ----
  for (uint32_t base = 0; base < page_n; base += kBatchN) {
#pragma unroll
    for (int v = 0; v < 3; v++) {
      if (!present[v] || trivial[v] || base >= n_lat[v]) continue;
      const uint32_t cnt = n_lat[v] - base < kBatchN ? n_lat[v] - base : kBatchN;
      if (v == 0) page_pack_batch<uint32_t>(sink, smem + voff[v], lat_ptr<uint32_t>(ws, t, 0) + pstart + skip[v], dissect_ptr(ws, t, 0) + pstart + skip[v], base, cnt, needs_ans[v] != 0, max_ob[v], n_bins[v] <= 1, page_var_cap(n_bins[v]));
      else page_pack_batch<L>(sink, smem + voff[v], lat_ptr<L>(ws, t, v) + pstart + skip[v], dissect_ptr(ws, t, v) + pstart + skip[v], base, cnt, needs_ans[v] != 0, max_ob[v], n_bins[v] <= 1, page_var_cap(n_bins[v]));
    }
  }
  sink.finish_byte();
  const uint64_t bytes = sink.close();
  if (lane == 0) store_result(result, bytes, sink.overflow ? PCO_GFX_INVALID_ARGUMENT : PCO_GFX_OK, 0);
}

__device__ __forceinline__ bool page_is_fast(const EncChunk PCO_GLOBAL* ch, const EncPage PCO_GLOBAL* pg) {
  return uni(ch->status) == PCO_GFX_OK && uni(ch->fast_ok) != 0 && !(uni(pg->flags) & kPageFlagMetaOnly);
}

// grid = number of page tasks; results are per page task
#ifndef PCO_PAGE_MIN_WAVES
#define PCO_PAGE_MIN_WAVES 4
#endif
__global__ __launch_bounds__(64, PCO_PAGE_MIN_WAVES) void enc_page_kernel(EncWorkspace ws, const PcoGfxEncodeTask* tasks, PcoGfxTaskResult* results, uint32_t n_pages, uint32_t skip_fast) {
  const uint32_t p = blockIdx.x;
  if (p >= n_pages) return;
  EncPage PCO_GLOBAL* pg = (EncPage PCO_GLOBAL*)ws.pages + p;
  const uint32_t t = uni(pg->chunk);
  const PcoGfxEncodeTask task = tasks[t];
  const uint32_t status = uni(ws.chunks[t].status);
  if (skip_fast && page_is_fast((const EncChunk PCO_GLOBAL*)ws.chunks + t, pg)) return;   // encode_fast.hip wrote this page
  PcoGfxTaskResult PCO_GLOBAL* res = (PcoGfxTaskResult PCO_GLOBAL*)results + p;
  if (status != PCO_GFX_OK) {
    if (lane_id() == 0) store_result(res, 0, status, 0);
    return;
  }
  const int bits = dtype_bits(uni(task.dtype));
  if (bits == 64) page_task<uint64_t>(ws, task, pg, res);
  else if (bits == 32) page_task<uint32_t>(ws, task, pg, res);
  else if (bits == 16) page_task<uint16_t>(ws, task, pg, res);
  else page_task<uint8_t>(ws, task, pg, res);
}

}  // namespace pcogfx
