// How much does an extra, independent LDS read per iteration cost a single wave that is chasing pointers through the LDS?
// hipcc --offload-arch=gfx950 -O3 lds_micro.hip -o lds_micro
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define LDSAS __attribute__((address_space(3)))
// V: bit0 extra b64 after the chase load, bit1 extra b64 before it, bit2 extra u8 after, bit3 16 VALU fillers, bit4 second b64 after,
//    bit5: the extra read's address depends on the chased value
template <int V>
__global__ __launch_bounds__(64) void k(uint64_t* out, const uint32_t* init, int iters) {
  extern __shared__ uint8_t lds[];
  uint32_t LDSAS* l32 = (uint32_t LDSAS*)lds;
  for (int i = threadIdx.x; i < 8192; i += 64) l32[i] = init[i];
  __syncthreads();
  const uint32_t lane = threadIdx.x;
  uint32_t a = (lane * 148u) & 32764u, fa = (lane >> 2) * 512u, acc = 0, f0 = lane, f1 = lane * 3;
  uint64_t pend = 0, pend2 = 0; uint32_t pend8 = 0;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      // "chain": a few dependent VALU ops on the chased value, as in the walker
      uint32_t x = a ^ (a >> 3); x = (x + acc) & 32764u; x = x ^ 4u; x = (x * 1u + 8u) & 32764u;
      if (V & 2) { acc ^= (uint32_t)pend; pend = *(const uint64_t LDSAS*)(uintptr_t)((fa + 8u * u) & 32760u); }
      a = *(const uint32_t LDSAS*)(uintptr_t)x;
      __builtin_amdgcn_sched_barrier(0);
      if (V & 1) { acc ^= (uint32_t)pend; const uint32_t ad = (V & 32) ? (x ^ 256u) & 32760u : (fa + 8u * u) & 32760u; pend = *(const uint64_t LDSAS*)(uintptr_t)ad; }
      if (V & 16) { acc ^= (uint32_t)pend2; pend2 = *(const uint64_t LDSAS*)(uintptr_t)((fa + 64u + 8u * u) & 32760u); }
      if (V & 4) { acc += pend8; pend8 = *(const uint8_t LDSAS*)(uintptr_t)((fa + u) & 32767u); }
      if (V & 8) {
#pragma unroll
        for (int q = 0; q < 8; q++) { f0 = f0 * 3u + f1; f1 = f1 ^ (f0 >> 5); }
      }
      fa += 40;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) out[0] = t1 - t0;
  if (a + acc + f0 + f1 + (uint32_t)pend + (uint32_t)pend2 + pend8 == 0x12345) out[1] = 1;
}
template <int V> void run(const char* name, uint64_t* dout, uint32_t* dinit) {
  const int iters = 4096;
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 32768, 0, dout, dinit, iters);
  (void)hipDeviceSynchronize();
  uint64_t h[2]; (void)hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost);
  printf("%-64s %.1f cycles/iteration\n", name, (double)h[0] / (4.0 * iters));
}
int main() {
  std::vector<uint32_t> init(8192);
  uint64_t s = 88172645463325252ull;
  for (auto& v : init) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 11) & 32764u; }
  uint32_t* dinit; uint64_t* dout;
  (void)hipMalloc(&dinit, init.size() * 4); (void)hipMalloc(&dout, 64);
  (void)hipMemcpy(dinit, init.data(), init.size() * 4, hipMemcpyHostToDevice);
  run<0>("pointer chase + 5 VALU", dout, dinit);
  run<8>("+ 16 independent VALU", dout, dinit);
  run<4>("+ independent ds_read_u8 after the chase load", dout, dinit);
  run<1>("+ independent ds_read_b64 after the chase load", dout, dinit);
  run<2>("+ independent ds_read_b64 before the chase load", dout, dinit);
  run<1 | 16>("+ 2 independent ds_read_b64 after", dout, dinit);
  run<1 | 4>("+ ds_read_b64 + ds_read_u8 after", dout, dinit);
  run<1 | 32>("+ ds_read_b64 after, address from the chased value", dout, dinit);
  run<1 | 8>("+ ds_read_b64 after + 16 VALU", dout, dinit);
  run<1 | 4 | 8>("+ ds_read_b64 + ds_read_u8 after + 16 VALU", dout, dinit);
  return 0;
}
