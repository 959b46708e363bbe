// stream_kernels.hip -- assembling / taking apart a stream of standalone chunks on the device.
//
// standalone::simple_compress writes its chunks back to back (standalone/simple.rs:62-91); the batched encoder leaves
// every chunk in its own worst-case-sized slot.  pco_gfx_compact_chunks turns the slots into that contiguous byte
// stream without a host round trip: an exclusive scan of the chunk sizes (one block), then a copy in which every
// destination dword-quad is written once, aligned, from an unaligned source read (chunks land at arbitrary byte
// offsets).  The result is what a file writer appends after the header, and what a rank hands to the RCCL gather of a
// chunk-sharded file (pcodec_amd/sharding.py).  HBM-bound: bytes read + bytes written = 2 x compressed size.
#pragma once
#include "pco_dev.h"

namespace pcogfx {

// offsets[i] = base + sum of the sizes of the chunks before i; offsets[n] = end.  A failed chunk contributes nothing.
__global__ __launch_bounds__(1024) void compact_scan_kernel(const PcoGfxTaskResult* res, uint32_t n, uint64_t base, uint64_t dst_cap, uint64_t* offsets, uint32_t* overflow) {
  __shared__ uint64_t part[16];
  __shared__ uint64_t carry_s;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = base;
  __syncthreads();
  for (uint32_t i0 = 0; i0 < n; i0 += 1024) {
    const uint32_t i = i0 + tid;
    const uint64_t sz = i < n && res[i].status == PCO_GFX_OK ? res[i].n_out : 0ull;
    const uint64_t incl = wave_incl_scan(sz);
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    uint64_t before = carry_s;
    for (uint32_t w = 0; w < wave; w++) before += part[w];
    if (i < n) offsets[i] = before + incl - sz;
    __syncthreads();
    if (tid == 1023) carry_s = before + incl;
    __syncthreads();
  }
  // (an asynchronous caller never sees `overflow`: the end offset it is handed reads ~0 when the destination is too small)
  if (tid == 0) { const bool over = carry_s > dst_cap; offsets[n] = over ? ~0ull : carry_s; *overflow = over ? 1u : 0u; }
}

// linear grid of n * slices_per_chunk blocks (gridDim.y stops at 65535): block b copies slice b % slices of chunk b / slices,
// i.e. bytes [s * slice, (s + 1) * slice) of the chunk
__global__ __launch_bounds__(256) void compact_copy_kernel(const PcoGfxEncodeTask* tasks, const PcoGfxTaskResult* res, const uint64_t* offsets, uint8_t* dst,
                                                           const uint32_t* overflow, uint32_t n, uint64_t slice_bytes, uint32_t slices) {
  const uint32_t i = blockIdx.x / slices, sl = blockIdx.x - i * slices;
  if (i >= n || *overflow) return;
  if (res[i].status != PCO_GFX_OK) return;
  const uint64_t len = res[i].n_out;
  const uint64_t s0 = (uint64_t)sl * slice_bytes;
  if (s0 >= len) return;
  const uint64_t s1 = s0 + slice_bytes < len ? s0 + slice_bytes : len;
  gcptr_u8 src = (gcptr_u8)tasks[i].dst;
  gptr_u8 out = (gptr_u8)dst + offsets[i];
  // destination-aligned 16-byte quads inside [s0, s1); the bytes before the first / after the last quad go one by one
  const uint64_t a0 = (uint64_t)(uintptr_t)(out + s0);
  uint64_t head = ((16 - (a0 & 15)) & 15);
  if (head > s1 - s0) head = s1 - s0;
  const uint64_t q0 = s0 + head, nq = (s1 - q0) >> 4, tail0 = q0 + (nq << 4);
  const uint32_t tid = threadIdx.x;
  if (tid < head) out[s0 + tid] = src[s0 + tid];
  if (tid < s1 - tail0) out[tail0 + tid] = src[tail0 + tid];
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;
  uint64_t k = tid;
  for (; k + 3 * 256 < nq; k += 4 * 256) {   // four loads in flight per thread
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = *(const u32x4_unaligned PCO_GLOBAL*)(src + q0 + ((k + u * 256) << 4));
#pragma unroll
    for (int u = 0; u < 4; u++) *(u32x4 PCO_GLOBAL*)(out + q0 + ((k + u * 256) << 4)) = v[u];
  }
  for (; k < nq; k += 256) *(u32x4 PCO_GLOBAL*)(out + q0 + (k << 4)) = *(const u32x4_unaligned PCO_GLOBAL*)(src + q0 + (k << 4));
}

}  // namespace pcogfx
