// Micro-benchmark of the decode walker's step (dec_walk_kernel, decode_fast.hip): one wave, synthetic tANS tables, cycles per step
// for the full step and for stripped-down variants.  hipcc --offload-arch=gfx950 -O3 walk_micro.hip -o walk_micro && ./walk_micro
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define LDSAS __attribute__((address_space(3)))
struct __attribute__((packed, aligned(1))) Q4 { uint32_t a, b, c, d; };
struct __attribute__((aligned(16))) A16 { uint32_t a, b, c, d; };
struct Regs { uint32_t fa; uint32_t saddr, e, bitaddr, d0, d1, d2, d3, x0, x1, obsum, obp, symacc; };
struct Masks { uint32_t m1, m2, m3, c63; };
constexpr uint32_t kSlice = 4912, kWin = 0, kTbl = 640, kOb = 640 + 4096;

template <int V>
__device__ __forceinline__ void step(Regs& r, const Masks& m, uint32_t obs_addr, uint32_t win_lo, uint32_t win_span) {
  const uint32_t e = r.e;
  uint32_t p;
  if (V & 1) p = e & 15u;   // no DPP prefix
  else p = ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)e, 0x111, 0xf, 0xf, true) & m.m1) + ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)e, 0x112, 0xf, 0xf, true) & m.m2) +
           ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)e, 0x113, 0xf, 0xf, true) & m.m3);
  const uint64_t xs = (((uint64_t)r.x1 << 32) | r.x0) >> (p & 63u);
  const uint32_t v = __builtin_amdgcn_ubfe((uint32_t)xs, 0u, e);
  r.saddr = (e >> 16) + (v << 2);
  r.e = *(const uint32_t LDSAS*)(uintptr_t)r.saddr;
  __builtin_amdgcn_sched_barrier(0);
  if (!(V & 2)) {   // shadow: window
    const uint32_t t = p + e;
    const uint32_t tot = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0xFF, 0xf, 0xf, true) & m.c63;
    constexpr int F = (V >> 4) & 7;   // window fetch flavour
    if (F == 2) {        // dword-aligned: 2 x ds_read2_b32, three-way cut
      const uint32_t o = (r.bitaddr & 31u) + tot;
      const uint32_t ab0 = __builtin_amdgcn_alignbit(r.d1, r.d0, o), ab1 = __builtin_amdgcn_alignbit(r.d2, r.d1, o), ab2 = __builtin_amdgcn_alignbit(r.d3, r.d2, o), ab3 = r.d3 >> (o & 31u);
      const bool lt = o < 32, lt2 = o < 64;
      r.x0 = lt ? ab0 : (lt2 ? ab1 : ab2); r.x1 = lt ? ab1 : (lt2 ? ab2 : ab3);
    } else {
      const uint32_t o = (r.bitaddr & 7u) + tot;
      const uint32_t ab0 = __builtin_amdgcn_alignbit(r.d1, r.d0, o), ab1 = __builtin_amdgcn_alignbit(r.d2, r.d1, o), ab2 = __builtin_amdgcn_alignbit(r.d3, r.d2, o);
      const bool lt = o < 32;
      r.x0 = lt ? ab0 : ab1; r.x1 = lt ? ab1 : ab2;
    }
    r.bitaddr += tot;
    if (r.bitaddr > win_lo + win_span) r.bitaddr -= win_span;   // stay inside the synthetic window (not in the real kernel)
    if (!(V & 8)) {
      if (F == 0) { const Q4 LDSAS* w = (const Q4 LDSAS*)(uintptr_t)(r.bitaddr >> 3); r.d0 = w->a; r.d1 = w->b; r.d2 = w->c; r.d3 = w->d; }
      else if (F == 1) { const A16 LDSAS* w = (const A16 LDSAS*)(uintptr_t)((r.bitaddr >> 3) & ~15u); r.d0 = w->a; r.d1 = w->b; r.d2 = w->c; r.d3 = w->d; }
      else if (F == 2) { const uint32_t LDSAS* w = (const uint32_t LDSAS*)(uintptr_t)((r.bitaddr >> 3) & ~3u); r.d0 = w[0]; r.d1 = w[1]; r.d2 = w[2]; r.d3 = w[3]; }
      else if (F == 3) { const uint64_t LDSAS* w = (const uint64_t LDSAS*)(uintptr_t)((r.bitaddr >> 3) & ~7u); const uint64_t a = w[0], b = w[1]; r.d0 = (uint32_t)a; r.d1 = (uint32_t)(a >> 32); r.d2 = (uint32_t)b; r.d3 = (uint32_t)(b >> 32); }
      else if (F == 4) { typedef uint64_t __attribute__((aligned(1))) u64u; const u64u LDSAS* w = (const u64u LDSAS*)(uintptr_t)(r.bitaddr >> 3); const uint64_t a = w[0], b = w[1]; r.d0 = (uint32_t)a; r.d1 = (uint32_t)(a >> 32); r.d2 = (uint32_t)b; r.d3 = (uint32_t)(b >> 32); }
      else if (F == 5) { typedef uint32_t __attribute__((aligned(1))) u32u; const u32u LDSAS* w = (const u32u LDSAS*)(uintptr_t)(r.bitaddr >> 3); r.d0 = w[0]; r.d1 = w[1]; r.d2 = w[2]; r.d3 = w[3]; }
      else if (F == 7) { r.fa += 24; if (r.fa > (win_lo + win_span) >> 3) r.fa -= win_span >> 3; const uint64_t LDSAS* w = (const uint64_t LDSAS*)(uintptr_t)(r.fa & ~7u); const uint64_t a = w[0]; r.d0 = (uint32_t)a; r.d1 = (uint32_t)(a >> 32); }
      else if (F == 6) { const uint64_t LDSAS* w = (const uint64_t LDSAS*)(uintptr_t)((r.bitaddr >> 3) & ~7u); const uint64_t a = w[0]; if (V & 256) { asm volatile("" :: "v"(a)); } else { r.d0 = (uint32_t)a; r.d1 = (uint32_t)(a >> 32); } }
    }
  }
  if (!(V & 4)) {   // shadow: symbol + offset bits
    r.symacc = __builtin_amdgcn_perm(e, r.symacc, 0x03020105u);
    const uint32_t ob = *(const uint8_t LDSAS*)(uintptr_t)(obs_addr + ((e >> 8) & 0xffu));
    r.obsum += r.obp; r.obp = ob;
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int V>
__global__ __launch_bounds__(64) void k(uint64_t* out, const uint32_t* init, int iters, int active_lanes) {
  extern __shared__ uint8_t lds[];
  uint32_t LDSAS* l32 = (uint32_t LDSAS*)lds;
  for (int i = threadIdx.x; i < 8 * (int)kSlice / 4; i += 64) l32[i] = init[i];
  __syncthreads();
  const uint32_t lane = threadIdx.x, slot = (lane >> 2) & 7, j = lane & 3;
  const uint32_t base = (uint32_t)(uintptr_t)(uint8_t LDSAS*)lds + slot * kSlice;
  Masks m = {j >= 1 ? ~0u : 0u, j >= 2 ? ~0u : 0u, j >= 3 ? ~0u : 0u, 63u};
  asm volatile("" : "+v"(m.m1), "+v"(m.m2), "+v"(m.m3), "+v"(m.c63));
  Regs r = {};
  r.saddr = base + kTbl + 4 * (lane * 37 % 1024); r.e = *(const uint32_t LDSAS*)(uintptr_t)r.saddr;
  r.bitaddr = 8 * (base + kWin) + lane % 13; r.fa = base + kWin;
  const uint32_t obs = base + kOb;
  uint64_t t0 = 0, t1 = 0;
  if ((int)lane < active_lanes) {
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
      step<V>(r, m, obs, 8 * (base + kWin), 8 * 400); step<V>(r, m, obs, 8 * (base + kWin), 8 * 400);
      step<V>(r, m, obs, 8 * (base + kWin), 8 * 400); step<V>(r, m, obs, 8 * (base + kWin), 8 * 400);
    }
    t1 = __builtin_readcyclecounter();
  }
  if (lane == 0) out[blockIdx.x * 2] = t1 - t0;
  if (r.saddr + r.obsum + r.symacc + r.x0 == 0x12345) out[blockIdx.x * 2 + 1] = r.e;
}

template <int V> void run(const char* name, uint64_t* dout, uint32_t* dinit, int blocks, int active) {
  const int iters = 4096;
  hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 40352);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 40352, 0, dout, dinit, iters, active);
  hipDeviceSynchronize();
  uint64_t h[2]; hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost);
  printf("%-44s blocks %5d lanes %2d: %.1f cycles/step\n", name, blocks, active, (double)h[0] / (4.0 * iters));
}

int main() {
  std::vector<uint32_t> init(8 * kSlice / 4);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); };
  for (int q = 0; q < 8; q++) {
    uint32_t* sl = init.data() + q * kSlice / 4;
    for (int i = 0; i < 112; i++) sl[i] = rnd();                                   // window
    for (int i = 0; i < 1024; i++) {                                                 // entries: btr 0..10, next base keeps base + 2^btr <= 1024
      const uint32_t btr = rnd() % 11, nb = rnd() % (1024 - (1u << btr) + 1), sym = rnd() % 176;
      sl[kTbl / 4 + i] = btr | (sym << 8) | ((q * kSlice + kTbl + 4 * nb) << 16);   // absolute LDS address (dynamic LDS starts at 0)
    }
    for (int i = 0; i < 44; i++) sl[kOb / 4 + i] = rnd() & 0x0f0f0f0f;
  }
  uint32_t* dinit; uint64_t* dout;
  hipMalloc(&dinit, init.size() * 4); hipMalloc(&dout, 1 << 16);
  hipMemcpy(dinit, init.data(), init.size() * 4, hipMemcpyHostToDevice);
  for (int blocks : {1}) {
    run<0>("full step, unaligned b128", dout, dinit, blocks, 32);
    run<16>("fetch: aligned b128", dout, dinit, blocks, 32);
    run<32>("fetch: dword-aligned 2 x read2_b32, 3-way cut", dout, dinit, blocks, 32);
    run<48>("fetch: 2 x aligned b64", dout, dinit, blocks, 32);
    run<64>("fetch: 2 x unaligned b64", dout, dinit, blocks, 32);
    run<80>("fetch: 4 x unaligned b32", dout, dinit, blocks, 32);
    run<96>("fetch: 1 x aligned b64", dout, dinit, blocks, 32);
    run<96 | 256>("fetch: 1 x aligned b64, result unused", dout, dinit, blocks, 32);
    run<112>("fetch: 1 x aligned b64, independent address", dout, dinit, blocks, 32);
    run<8>("no window fetch", dout, dinit, blocks, 32);
    run<2 | 4>("chain only (dpp prefix)", dout, dinit, blocks, 32);
  }
  return 0;
}
