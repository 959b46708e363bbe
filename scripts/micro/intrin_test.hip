// micro test of intrinsic semantics on gfx950 (debug aid, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned lane = threadIdx.x;
  unsigned e = 0xA1B2C3D4u, acc = 0x11223344u;
  out[lane * 8 + 0] = __builtin_amdgcn_perm(e, 0u, 0x0c0c0c07u);
  out[lane * 8 + 1] = __builtin_amdgcn_perm(e, acc, 0x03020106u);
  out[lane * 8 + 2] = __builtin_amdgcn_sad_u8(0x01020304u, 0u, 100u);
  out[lane * 8 + 3] = __builtin_amdgcn_alignbit(0xAAAAAAAAu, 0x12345678u, 36u);
  unsigned bb = (lane & 3) == 0 ? 1u : ((lane & 3) == 1 ? 0x200u : ((lane & 3) == 2 ? 0x30000u : 0x4000000u));
  unsigned P = bb + (unsigned)__builtin_amdgcn_update_dpp(0, (int)bb, 0xB1, 0xf, 0xf, false);
  P = P + (unsigned)__builtin_amdgcn_update_dpp(0, (int)P, 0x4E, 0xf, 0xf, false);
  out[lane * 8 + 4] = P;
  out[lane * 8 + 5] = __builtin_amdgcn_ubfe(0xffffffffu, 0u, lane & 15);
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 8 * 4); k<<<1, 64>>>(d); unsigned h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 6; l++) printf("lane %d: perm1=%08x perm2=%08x sad=%u align=%08x P=%08x bfe=%08x\n", l, h[l*8], h[l*8+1], h[l*8+2], h[l*8+3], h[l*8+4], h[l*8+5]);
  return 0;
}
